"""SR-artifact stages: mirrors of `BlurCortex`, `StructNoise`, `SimulateMotion`, `SimulatedBoundaries`
(`fetalsyngen.generator.augmentation.artifacts`, reference augmentation/artifacts.py:24-604) on MI355X kernels.

Same constructors, `__call__(output, seg, device, genparams, **kwargs) -> (tensor, metadata)` contract, metadata
keys and numpy/torch draw order.  Volumes are only touched by HIP kernels (fsg_artifacts.hip, fsg_blur.hip,
fsg_zoom.hip, fsg_slice_acq.hip); the host draws the random plan and does the small-array algebra.

Reproducibility note (reference behaviour, kept): `generate_fractal_noise_3d` re-seeds numpy's global generator
from the wall clock (generator/artifacts/utils.py:365-367), so with the default "perlin" merge type every numpy
draw after the first Perlin field of a sample is not seed-reproducible in the reference either.
"""
from __future__ import annotations

from dataclasses import asdict, fields

import numpy as np
import torch

from ... import kernels as K
from ... import rng as _rng
from ... import tables as T
from ..artifacts.simulate_reco import PSFReconstructor, Scanner
from ..artifacts.utils import (ReconParams, ScannerParams, StructNoiseMergeParams, fractal_noise_plan,  # noqa: F401
                               gaussian_blur_3d, mog_params)
from .synthseg import RandTransform


def _need_gpu(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("fetalsyngen_amd artifact stages run on an MI355X only (device='cuda:N'); there is no CPU fallback")


def _pick_voxels(vol, op, value, ranks_fn):
    """ranks_fn(count) -> int64 ranks among the voxels with `vol op value` (raster order); returns (k,3) coordinates."""
    count, select = K.nonzero_ranks(vol.contiguous(), op, value)
    return select(ranks_fn(count))


class BlurCortex(RandTransform):
    """Locally blurred cortex: Gaussian-blurred copy blended in through a mixture of Gaussian blobs centred on
    cortex voxels, preferentially frontal (ref :24-133)."""

    def __init__(self, prob: float, cortex_label: int, nblur_min: int, nblur_max: int, sigma_gamma_loc: int = 3,
                 sigma_gamma_scale: int = 1, std_blur_shape: int = 2, std_blur_scale: int = 1):
        self.prob = prob
        self.cortex_label = cortex_label
        self.nblur_min = nblur_min
        self.nblur_max = nblur_max
        self.sigma_gamma_loc = sigma_gamma_loc
        self.sigma_gamma_scale = sigma_gamma_scale
        self.std_blur_shape = std_blur_shape
        self.std_blur_scale = std_blur_scale

    def blur_proba(self, shape, seg, device):
        """Sampling probability of each cortex voxel (raster order), host float32 (ref :64-81): two wide blobs on the
        frontal side, read at the cortex voxels and normalised.  `seg`: the float label map (cortex = cortex_label)."""
        p = self._cortex_weights(shape, seg, device)
        return p / p.sum()

    def _cortex_weights(self, shape, seg, device):
        """The blob field at the cortex voxels, not normalised (host float32, raster order)."""
        x, y, z = shape
        prob = K.mog3d(shape, *mog_params([(0, y, z // 2), (x, y, z // 2)], [x // 5, y // 5]), device)
        return K.compact_values(prob, seg, "==", float(self.cortex_label)).cpu()

    def __call__(self, output, seg, device, genparams: dict = {}, **kwargs):
        if np.random.rand() < self.prob or len(genparams.keys()) > 0:
            _need_gpu(output)
            nblur = np.random.randint(self.nblur_min, self.nblur_max) if "nblur" not in genparams.keys() else genparams["nblur"]
            std_blurs = np.random.gamma(self.std_blur_shape, self.std_blur_scale, 3)
            seg = seg.to(output.device).float().contiguous()
            # reference mode: the normalised probabilities into torch.multinomial, as the reference (:110); otherwise the
            # inverse-CDF draw scales by the total itself, and the million-element divide stays off the host
            cortex_prob = (self.blur_proba(output.shape, seg, output.device) if _rng.get_mode() == "reference"
                           else self._cortex_weights(output.shape, seg, output.device))
            idx = _rng.multinomial_distinct(cortex_prob, nblur)  # CPU generator
            count, select = K.nonzero_ranks(seg, "==", float(self.cortex_label))
            centers = select(idx)
            sigmas = np.random.gamma(self.sigma_gamma_loc, self.sigma_gamma_scale, (nblur, 3))
            gaussian = K.mog3d(output.shape, centers.numpy().astype(np.float32), sigmas.astype(np.float32), output.device)
            out = output.float().contiguous()
            output_blur = gaussian_blur_3d(out, stds=std_blurs, device=output.device)
            output = K.blend(out, output_blur.contiguous(), gaussian)  # out*(1-g) + blur*g
            return output, {"nblur": nblur}
        return output, {"nblur": None}


def _interp_tables(n_in, n_out):
    """Per-axis table of F.interpolate(mode='trilinear', align_corners=False) (ATen area_pixel_compute_source_index:
    src = scale*(dst+0.5)-0.5 clamped at 0, scale = n_in/n_out in fp32)."""
    scale = np.float32(n_in) / np.float32(n_out)
    d = np.arange(n_out, dtype=np.float32)
    src = np.maximum(scale * (d + np.float32(0.5)) - np.float32(0.5), np.float32(0.0)).astype(np.float32)
    lo = src.astype(np.int64)
    hi = lo + (lo < n_in - 1)
    w_hi = (src - lo.astype(np.float32)).astype(np.float32)
    tab = np.zeros(n_out, dtype=T.TAP_DTYPE)
    tab["lo"], tab["hi"], tab["w_lo"], tab["w_hi"] = lo, hi, np.float32(1.0) - w_hi, w_hi
    return tab


class StructNoise(RandTransform):
    """Multi-scale structured noise merged into the image inside the brain through a Perlin / Gaussian weight
    field (ref :136-342)."""

    def __init__(self, prob: float, wm_label: int, std_min: float, std_max: float, merge_params: StructNoiseMergeParams,
                 nstages_min: int = 1, nstages_max: int = 5):
        self.prob = prob
        self.wm_label = wm_label
        self.nstages_min = nstages_min
        self.nstages_max = nstages_max
        self.std_min = std_min
        self.std_max = std_max
        self.merge_params = merge_params

    def sample_seeds(self, genparams: dict = {}):
        mp = self.merge_params
        self.nstages = np.random.randint(self.nstages_min, self.nstages_max) if "nstages" not in genparams else genparams["nstages"]
        self.noise_std = self.std_min + (self.std_max - self.std_min) * np.random.rand()
        if mp.merge_type == "gaussian":
            self.gauss_nloc = (np.random.randint(mp.gauss_nloc_min, mp.gauss_nloc_max) if "nloc" not in genparams
                               else genparams["nloc"])
        elif mp.merge_type == "perlin":
            self._res = genparams["res"] if "res" in genparams else np.random.choice(mp.perlin_res_list)
            self._octave = genparams["octave"] if "octave" in genparams else np.random.choice(mp.perlin_octaves_list)

    def get_seeds(self):
        seeds = {"nstages": self.nstages, "noise_std": self.noise_std}
        if self.merge_params.merge_type == "gaussian":
            seeds["nloc"] = self.gauss_nloc
        elif self.merge_params.merge_type == "perlin":
            seeds["res"] = self._res
            seeds["octave"] = self._octave
        return seeds

    def _weights(self, shape, seg, device):
        """(weight volume or raw Perlin noise, its min/max keys or None)."""
        mp = self.merge_params
        if mp.merge_type == "gaussian":
            count, select = K.nonzero_ranks(seg, "==", float(self.wm_label))
            centers = select(_rng.distinct_ranks(count, self.gauss_nloc))
            sig = torch.clamp(mp.gauss_sigma_mu + mp.gauss_sigma_std * torch.randn(len(centers)), 1, 40).numpy()
            c, s = mog_params([tuple(v) for v in centers.tolist()], sig)
            return K.mog3d(shape, c, s, device), None
        if mp.merge_type == "perlin":
            plan = fractal_noise_plan(shape, (self._res,) * 3, octaves=self._octave, persistence=mp.perlin_persistence,
                                      lacunarity=mp.perlin_lacunarity, device=device)
            return K.perlin_fractal(plan)
        raise RuntimeError

    def get_merging_weights(self, shape, mask=None, device=None):
        """The weight volume itself (ref :185-236); `__call__` blends without materialising it.  `mask` is the white
        matter mask (bool / uint8 / float)."""
        if self.merge_params.merge_type == "gaussian":
            count, select = K.nonzero_ranks(mask.reshape(mask.shape[-3:]).contiguous(), ">", 0.0)
            centers = select(_rng.distinct_ranks(count, self.gauss_nloc))
            mp = self.merge_params
            sig = torch.clamp(mp.gauss_sigma_mu + mp.gauss_sigma_std * torch.randn(len(centers)), 1, 40).numpy()
            return K.mog3d(shape, *mog_params([tuple(v) for v in centers.tolist()], sig), device)
        raw, mm = self._weights(tuple(shape), None, device)
        return K.blend(None, None, raw, w_mm=mm, increase=self.merge_params.perlin_increase_size, want_weight=True,
                       want_out=False)[1]

    def _multiscale_noise(self, shape, device):
        """`nstages` rounds of (add white noise, double the grid by trilinear interpolation) (ref :308-320)."""
        lr = None
        for k in range(self.nstages):
            s_k = [i // 2 ** (self.nstages - k) for i in shape]
            s_next = [i // 2 ** (self.nstages - 1 - k) for i in shape]
            z = _rng.normal_field(s_k, stream_id=4 + k).device_tensor(device)
            lr = z if lr is None else K.axpy(lr, z)
            tabs = K.DeviceTables([_interp_tables(a, b) for a, b in zip(s_k, s_next)], device)
            lr = K.zoom3d(lr.contiguous(), tabs)
        return lr

    def __call__(self, output, seg, device, genparams: dict = {}, **kwargs):
        if np.random.rand() < self.prob or "nloc" in genparams.keys():
            _need_gpu(output)
            self.sample_seeds()
            out = output.float().contiguous()
            shape = tuple(out.shape)
            seg = seg.to(out.device).float().contiguous()
            lr = self._multiscale_noise(shape, out.device)
            if tuple(lr.shape) != shape:
                raise ValueError(f"StructNoise needs volume extents divisible by 2^nstages ({shape}, nstages={self.nstages})")
            w, w_mm = self._weights(shape, seg, out.device)
            # noisy = clamp(out + std * lr / max|lr|, 0, 2 max(out)); result = (1 - m w) out + m w noisy, m = seg > 0
            output = K.blend(out, lr, w, w_mm=w_mm, increase=self.merge_params.perlin_increase_size or 0.0, seg=seg,
                             noise_std=self.noise_std, b_mm=K.reduce_minmax(lr), a_mm=K.reduce_minmax(out))
            return output, self.get_seeds()
        return output, {}


class SimulateMotion(RandTransform):
    """Acquire motion-corrupted low-resolution slice stacks of the image and reconstruct it from them (ref :345-425)."""

    def __init__(self, prob: float, scanner_params: ScannerParams, recon_params: ReconParams):
        self.scanner_args = scanner_params
        self.recon_args = recon_params
        self.prob = prob

    def __call__(self, output, seg, device, genparams: dict = {}, **kwargs):
        if np.random.rand() < self.prob:
            _need_gpu(output)
            device = output.device
            dshape = (1, 1, *output.shape[-3:])
            res = kwargs["resolution"]
            res_ = np.float64(res[0])
            segf = seg.to(device).float().contiguous()
            d = {
                "resolution": res_,
                "volume": output.float().contiguous().view(dshape),
                "mask": K.threshold(segf, 0.0).view(dshape),  # (seg > 0).float()
                "seg": segf.view(dshape),
                "affine": torch.diag(torch.tensor(list(res) + [1])).to(device),
                "threshold": 0.1,
            }
            self.scanner_args.resolution_recon = res_
            scanner = Scanner(**asdict(self.scanner_args))
            d_scan = scanner.scan(d)
            recon = PSFReconstructor(**{f.name: getattr(self.recon_args, f.name) for f in fields(self.recon_args)})
            output, _ = recon.recon_psf(d_scan, want_weight=False)
            metadata = {
                "resolution_recon": d_scan["resolution_recon"],
                "resolution_slice": d_scan["resolution_slice"],
                "slice_thickness": d_scan["slice_thickness"],
                "gap": d_scan["gap"],
                "nstacks": len(torch.unique(d_scan["positions_host"][:, 1])),
            }
            metadata.update(recon.get_seeds())
            return output.squeeze(), metadata
        return output, {}


class SimulatedBoundaries(RandTransform):
    """Skull-stripping boundary simulation: no masking, a halo around the brain mask, and / or fuzzy, locally
    varying boundaries (ref :428-604).  Masks are float32 0/1 volumes here (int32 in the reference)."""

    def __init__(self, prob_no_mask: float, prob_if_mask_halo: float, prob_if_mask_fuzzy: float):
        self.prob_no_mask = prob_no_mask
        self.prob_halo = prob_if_mask_halo
        self.prob_fuzzy = prob_if_mask_fuzzy
        self.reset_seeds()

    def reset_seeds(self):
        self.no_mask_on = None
        self.halo_on = None
        self.halo_radius = None
        self.fuzzy_on = None
        self.n_generate_fuzzy = None
        self.n_centers = None
        self.base_sigma = None

    def sample_seeds(self):
        """numpy draws: rand [, rand [, randint(5,15)], rand [, randint(2,5), poisson(100), poisson(8)]] (ref :468-482)."""
        self.reset_seeds()
        self.no_mask_on = np.random.rand() < self.prob_no_mask
        if not self.no_mask_on:
            self.halo_on = np.random.rand() < self.prob_halo
            if self.halo_on:
                self.halo_radius = np.random.randint(5, 15)
            self.fuzzy_on = np.random.rand() < self.prob_fuzzy
            if self.fuzzy_on:
                self.n_generate_fuzzy = np.random.randint(2, 5)
                self.n_centers = np.random.poisson(100)
                self.base_sigma = np.random.poisson(8)

    def build_halo(self, mask, radius) -> torch.Tensor:
        """Dilation by a ball of `radius` (ref :484-499: conv3d with skimage's ball(radius), (2r+1)^3 taps): three
        passes of a capped squared-distance transform and a threshold."""
        m = mask.reshape(mask.shape[-3:]).float().contiguous()
        r = int(radius)
        return K.less_equal(K.distance_to_mask(m, r, "euclid2"), float(r * r))

    def generate_fuzzy_boundaries(self, mask, kernel_size=7, threshold_filter=3) -> torch.Tensor:
        """Grow the mask by random blobs seeded in its 7-voxel shell, then close (ref :501-522).  The reference keeps a
        random 10 % of the shell voxels through a randperm; in device-RNG mode each shell voxel is kept with p = 0.1."""
        shape = mask.shape
        m = mask.reshape(shape[-3:]).float().contiguous()
        shell = K.sub_gt(K.threshold(K.box_sum3d(m, kernel_size), 0.0), m, 0.0)
        if _rng.get_mode() == "reference":
            count, select = K.nonzero_ranks(shell, "!=", 0.0)
            kept = torch.randperm(count)[int(count * 0.9):]
            diff = K.scatter_ones(m.shape, select(kept, flat_device=True), m.device)
        else:
            key = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
            diff = K.bernoulli_keep(shell, 0.1, key, stream_id=9)
        dsamp = K.threshold(K.box_sum3d(diff, 3), float(threshold_filter))
        grown = K.threshold(K.box_sum3d(K.maximum(m, dsamp), 5), 0.0)       # dilate(.., 5)
        closing = K.equals(K.box_sum3d(grown, 5), float(5**3))              # erode(.., 5)
        return closing.view(shape)

    def __call__(self, output, seg, device, genparams: dict = {}, **kwargs):
        self.sample_seeds()
        metadata = {"no_mask_on": self.no_mask_on, "halo_on": self.halo_on, "fuzzy_on": self.fuzzy_on}
        if self.no_mask_on:
            return output, metadata
        _need_gpu(output)
        out = output.float().contiguous()
        mask = K.threshold(seg.to(out.device).float().contiguous(), 0.0)
        if self.halo_on:
            mask = self.build_halo(mask, self.halo_radius)
        if self.fuzzy_on:
            mask_modif = mask
            for _ in range(self.n_generate_fuzzy):
                mask_modif = self.generate_fuzzy_boundaries(mask_modif)
            # centres of the probability blobs: random voxels among those the fuzzy growth added (ref :565-574)
            count, select = K.nonzero_ranks(K.sub_gt(mask_modif, mask, 0.0), ">", 0.0)
            centers = select(_rng.distinct_ranks(count, self.n_centers))
            sigmas = [self.base_sigma + 10 * np.random.beta(2, 5) for _ in range(len(centers))]
            if len(centers):
                mog = K.mog3d(mask.shape, *mog_params([tuple(v) for v in centers.tolist()], sigmas), out.device)
            else:
                mog = torch.zeros_like(mask)
            n_dilate = 6 * (self.n_generate_fuzzy - 1)
            dist = K.distance_to_mask(mask, n_dilate, "l1")
            return K.boundary_mask(out, mask, mask_modif, mog, dist, n_dilate), metadata
        return K.mul(out, mask), metadata
