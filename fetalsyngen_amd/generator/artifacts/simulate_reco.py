"""Mirror of `fetalsyngen.generator.artifacts.simulate_reco` (reference simulate_reco.py:38-774): simulated
slice-stack acquisition (`Scanner`) and PSF-weighted re-reconstruction (`PSFReconstructor`) on MI355X.

Same classes, constructor keywords, method names, data-dict keys and the same consumption order of the numpy /
torch global generators.  What runs where:

  host   -- every random draw; the rigid-transform algebra of <= a few hundred slices (svort/rigid.py); the PSFs
            (a few hundred taps); per-stack bookkeeping.  The reference does all of this with device tensors and
            one tiny kernel launch per operation.
  device -- everything that touches slices or volumes: `fsg_slice_acq_forward_f32` (PSF acquisition and the
            1-tap mask acquisition), `fsg_slice_sums_f32`, the slice corruptions (`fsg_gamma_f32`, `fsg_scale_f32`,
            `fsg_slice_noise_f32`, `fsg_slice_void_f32`), `fsg_slice_acq_adjoint_f32` + `fsg_equalize_f32`,
            the 3x3x3 smoothing (`fsg_blur_axis_*` with box taps), and the spatially weighted merge
            (`fsg_perlin_fractal_f32` / `fsg_mog3d_f32` + `fsg_blend_f32`).

RNG: the small torch draws of the reference (`torch.rand(n)` for the signal voids, `torch.randperm`, the Perlin
lattices) come from the CPU global generator in the reference's order; the one large field (two normals per slice
pixel for the Rician noise) follows `fetalsyngen_amd.rng`: "reference" = `torch.randn` on the CPU generator for the
pixels above the threshold, scattered to a dense field and uploaded (what a CPU run of the reference draws);
"device" = in-kernel Philox keyed by one `torch.randint`.
"""
from __future__ import annotations

from functools import partial

import numpy as np
import torch

from ... import kernels as K
from ... import rng as _rng
from ... import tables as T
from .svort import (RigidTransform, get_PSF, interleave_index, mat_update_resolution, random_angle,
                    random_init_stack_transforms, reset_transform, sample_motion, slice_acquisition,
                    slice_acquisition_adjoint)
from .utils import ReconMergeParams, fractal_noise_plan, mog_3d_tensor


def PSFreconstruction(transforms, slices, slices_mask, vol_mask, params, slice_ids=None):
    """Adjoint of the acquisition with the reconstruction PSF, equalised (ref :38-54)."""
    return slice_acquisition_adjoint(transforms, params["psf"], slices, slices_mask, vol_mask, params["volume_shape"],
                                     params["res_s"] / params["res_r"], params["interp_psf"], True, slice_ids=slice_ids)


def _axis_resample_tables(n_src, res, res_r, nearest):
    """Per-axis table of F.grid_sample(align_corners=True) on linspace(-gmax, gmax, size_new) (ref :321-328)."""
    size_new = int(n_src * res / res_r)
    grid_max = (size_new - 1) * res_r / (n_src - 1) / res
    g = torch.linspace(-grid_max, grid_max, size_new)
    x = ((g + 1) / 2) * (n_src - 1)
    tab = np.zeros(size_new, dtype=T.TAP_DTYPE)
    if nearest:
        r = torch.round(x).to(torch.int64).numpy()
        ok = (r >= 0) & (r < n_src)
        tab["lo"], tab["hi"] = np.where(ok, r, -1), np.where(ok, r, 0)
        tab["w_lo"], tab["w_hi"] = 1.0, 0.0
    else:
        f = torch.floor(x)
        lo = f.to(torch.int64).numpy()
        wh = (x - f).numpy()
        hi = np.minimum(lo + 1, n_src - 1)
        tab["lo"], tab["hi"], tab["w_lo"], tab["w_hi"] = lo, hi, 1 - wh, np.where(lo + 1 > n_src - 1, 0.0, wh)
    return tab


class Scanner:
    """Simulated acquisition of several motion-corrupted low-resolution slice stacks (ref :57-466)."""

    def __init__(self, resolution_slice_fac_min, resolution_slice_fac_max, resolution_slice_max, slice_thickness_min,
                 slice_thickness_max, gap_min, gap_max, min_num_stack, max_num_stack, max_num_slices, noise_sigma_min,
                 noise_sigma_max, TR_min, TR_max, prob_gamma, gamma_std, prob_void, slice_size, restrict_transform: bool,
                 txy: float, resolution_recon: float = None, slice_noise_threshold: float = 0.1):
        self.resolution_slice_fac_min = resolution_slice_fac_min
        self.resolution_slice_fac_max = resolution_slice_fac_max
        self.resolution_slice_max = resolution_slice_max
        self.slice_thickness_min = slice_thickness_min
        self.slice_thickness_max = slice_thickness_max
        self.gap_min = gap_min
        self.gap_max = gap_max
        self.min_num_stack = min_num_stack
        self.max_num_stack = max_num_stack
        self.max_num_slices = max_num_slices
        self.noise_sigma_min = noise_sigma_min
        self.noise_sigma_max = noise_sigma_max
        self.TR_min = TR_min
        self.TR_max = TR_max
        self.prob_gamma = prob_gamma
        self.gamma_std = gamma_std
        self.prob_void = prob_void
        self.slice_size = slice_size
        self.resolution_recon = resolution_recon
        self.restrict_transform = restrict_transform
        self.txy = txy
        self.slice_noise_threshold = slice_noise_threshold

    # ---- host draws ---------------------------------------------------------------------------------
    def get_resolution(self, data, genparams: dict = {}):
        """numpy draws: uniform (slice resolution) [, uniform (recon resolution)], uniform (thickness), uniform (gap)."""
        resolution = data["resolution"]
        if "resolution_slice_fac" not in genparams:
            resolution_slice = np.random.uniform(
                self.resolution_slice_fac_min * resolution,
                min(self.resolution_slice_fac_max * resolution, self.resolution_slice_max))
        else:
            resolution_slice = genparams["resolution_slice_fac"]
        if self.resolution_recon is not None:
            data["resolution_recon"] = self.resolution_recon
        else:
            data["resolution_recon"] = np.random.uniform(resolution, resolution_slice)
        data["resolution_slice"] = resolution_slice
        data["slice_thickness"] = (np.random.uniform(self.slice_thickness_min, self.slice_thickness_max)
                                   if "slice_thickness" not in genparams else genparams["slice_thickness"])
        data["gap"] = np.random.uniform(self.gap_min, self.gap_max) if "gap" not in genparams else genparams["gap"]
        return data

    def sample_time(self, n_slice, genparams: dict = {}):
        TR = np.random.uniform(self.TR_min, self.TR_max) if "TR" not in genparams else genparams["TR"]
        return np.arange(n_slice) * TR

    # ---- slice corruptions (device) -------------------------------------------------------------------
    def random_gamma(self, slices, genparams: dict = {}):
        """gate rand(); gamma = exp(std * randn); s = 300 (s/300)^gamma, then s / max(s) (ref :210-234)."""
        if np.random.rand() < self.prob_gamma:
            gamma = np.exp(self.gamma_std * np.random.randn(1)[0]) if "gamma" not in genparams else genparams["gamma"]
            flat = slices.reshape(-1).contiguous()
            flat = K.gamma(flat, float(gamma))
            return K.scale(flat, K.reduce_minmax(flat), mode=0).view(slices.shape)
        return slices

    def add_noise(self, slices, genparams: dict = {}):
        """Rician noise on the pixels above `slice_noise_threshold`, in place (ref :236-256)."""
        sigma = (np.random.uniform(self.noise_sigma_min, self.noise_sigma_max)
                 if "noise_sigma" not in genparams else genparams["noise_sigma"])
        if not slices.is_contiguous():
            raise ValueError("add_noise works in place on a contiguous slice stack")
        if _rng.get_mode() == "reference":
            mask = (slices > self.slice_noise_threshold).reshape(-1).cpu()  # the compaction order of slices[mask]
            m = int(mask.sum())
            dense = torch.zeros((2, mask.numel()), dtype=torch.float32)
            dense[0, mask] = torch.randn(m)
            dense[1, mask] = torch.randn(m)
            d = K._upload(dense, slices.device)
            K.slice_noise_(slices, self.slice_noise_threshold, sigma, d[0], d[1])
        else:
            key = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
            K.slice_noise_(slices, self.slice_noise_threshold, sigma, seed=key, stream_id=3)
        return slices

    def signal_void(self, slices):
        """Gaussian-shaped signal drops on a random subset of slices, in place (ref :258-298).
        torch draws (CPU generator): rand(n) gate, rand(nv) x2 centre, rand(nv,1,1) angle, then a, A, sx."""
        n = slices.shape[0]
        idx = torch.rand(n) < self.prob_void
        nv = int(idx.sum())
        if nv > 0:
            h, w = slices.shape[-2:]
            y = torch.linspace(-(h - 1) / 2, (h - 1) / 2, h)
            x = torch.linspace(-(w - 1) / 2, (w - 1) / 2, w)
            yc = (torch.rand(nv) - 0.5) * (h - 1)
            xc = (torch.rand(nv) - 0.5) * (w - 1)
            theta = 2 * np.pi * torch.rand((nv, 1, 1))
            c, s = torch.cos(theta), torch.sin(theta)
            a = 30 + torch.rand_like(theta) * 90
            A = torch.rand_like(theta) * 0.5 + 0.5
            sx = torch.rand_like(theta) * 30 + 39
            sy = a**2 / sx
            sx = -0.5 / sx**2
            sy = -0.5 / sy**2
            par = torch.stack([yc, xc, c.view(-1), s.view(-1), A.view(-1), sx.view(-1), sy.view(-1)], 1).float()
            ids = torch.nonzero(idx).view(-1).to(torch.int32)
            dev = slices.device
            K.slice_void_(slices.view(n, h, w), K._upload(ids, dev), K._upload(par.contiguous(), dev),
                          K._upload(y, dev), K._upload(x, dev))
        return slices

    # ---- the scan ----------------------------------------------------------------------------------
    def scan(self, data, genparams: dict = {}):
        data = self.get_resolution(data, genparams={})
        res, res_r, res_s = data["resolution"], data["resolution_recon"], data["resolution_slice"]
        s_thick, gap = data["slice_thickness"], data["gap"]
        volume = data["volume"]
        device = volume.device
        if not volume.is_cuda:
            raise RuntimeError("fetalsyngen_amd.Scanner runs on an MI355X only (device='cuda:N'); there is no CPU fallback")
        vs = volume.shape

        # ground truth on the reconstruction grid (ref :319-333)
        if res_r != res:
            tl = K.DeviceTables([_axis_resample_tables(vs[i + 2], res, res_r, False) for i in range(3)], device)
            tn = K.DeviceTables([_axis_resample_tables(vs[i + 2], res, res_r, True) for i in range(3)], device)
            volume_gt = K.zoom3d(volume.reshape(vs[-3:]).float().contiguous(), tl)[None, None]
            seg_gt = K.zoom3d(data["seg"].reshape(vs[-3:]).float().contiguous(), tn)[None, None]
        else:
            volume_gt, seg_gt = volume.clone(), data["seg"].clone()
        data["volume_gt"], data["seg_gt"] = volume_gt, seg_gt

        psf_acq = get_PSF(res_ratio=(res_s / res, res_s / res, s_thick / res), device=device)
        psf_rec = get_PSF(res_ratio=(res_s / res_r, res_s / res_r, s_thick / res_r), device=device)
        psf_delta = get_PSF(0, device=device)
        data["psf_rec"], data["psf_acq"] = psf_rec, psf_acq

        if self.slice_size is None:
            ss = int(np.sqrt((vs[-1] ** 2 + vs[-2] ** 2 + vs[-3] ** 2) / 2.0) * res / res_s)
            ss = int(np.ceil(ss / 32.0) * 32)
        else:
            ss = self.slice_size
        ns = int(max(vs) * res / gap) + 2

        stacks, stacks_no_psf, transforms, transforms_gt, positions = [], [], [], [], []
        num_stacks = np.random.randint(self.min_num_stack, self.max_num_stack + 1)
        rand_motion = True
        while True:
            transform_init = random_init_stack_transforms(ns, gap, self.restrict_transform, self.txy, device)
            ts = self.sample_time(ns)
            transform_motion = sample_motion(ts, device, rand_motion)
            interleave_idx = interleave_index(ns, (np.random.randint(2, int(np.sqrt(ns)) + 1) if rand_motion else 2))
            transform_motion = transform_motion[interleave_idx]
            transform_target = transform_motion.compose(transform_init)

            mat = K._upload(mat_update_resolution(transform_target.matrix(), res_r, res).contiguous(), device)
            # The reference acquires all ns slices through the PSF, then the 1-tap mask acquisition, and only then keeps
            # the contiguous run of slices that see enough of the brain (ref :386-420).  The acquisitions draw nothing,
            # so the cheap mask acquisition goes first here and the PSF acquisition is run for the kept run only.
            slices_no_psf = slice_acquisition(mat, data["mask"], None, None, psf_delta, (ss, ss), res_s / res, False, False)
            nnz = K.slice_sums(slices_no_psf.view(ns, ss, ss)).cpu()  # one host sync per stack
            idx = nnz > (nnz.max() * np.random.uniform(0.1, 0.3))
            if idx.sum() == 0:
                continue
            nz = torch.nonzero(idx)
            first, last = int(nz[0, 0]), int(nz[-1, 0])
            idx[first:last] = True
            slices = slice_acquisition(mat[first : last + 1], volume, None, None, psf_acq, (ss, ss), res_s / res, False, False)
            slices_no_psf = slices_no_psf[first : last + 1]
            transform_init = reset_transform(transform_init[idx])
            transform_target = transform_target[idx]
            slices = self.random_gamma(slices)
            slices = self.add_noise(slices)
            slices = self.signal_void(slices)
            if (self.max_num_slices is not None
                    and sum(st.shape[0] for st in stacks) + slices.shape[0] >= self.max_num_slices):
                break
            stacks.append(slices)
            stacks_no_psf.append(slices_no_psf)
            transforms.append(transform_init)
            transforms_gt.append(transform_target)
            positions.append(torch.arange(slices.shape[0], dtype=slices.dtype) - slices.shape[0] // 2)
            if len(stacks) >= num_stacks:
                break
        stacks_ids = np.random.choice(20, len(stacks), replace=False)
        positions = torch.cat(
            [torch.stack((positions[i], torch.full_like(positions[i], s_i)), -1) for i, s_i in enumerate(stacks_ids)], 0)
        transforms = RigidTransform.cat(transforms)
        transforms_gt = RigidTransform.cat(transforms_gt)

        data["slice_shape"] = (ss, ss)
        data["volume_shape"] = volume_gt.shape[-3:]
        data["stacks"] = torch.cat(stacks, 0)
        data["stacks_no_psf"] = torch.cat(stacks_no_psf, 0)
        data["positions"] = positions.to(device)
        data["positions_host"] = positions
        data["transforms"] = transforms.matrix().to(device)
        data["transforms_angle"] = transforms
        data["transforms_gt"] = transforms_gt.matrix().to(device)
        data["transforms_gt_angle"] = transforms_gt
        data.pop("volume")
        return data


class PSFReconstructor:
    """PSF-weighted scattered-data reconstruction of the stacks with simulated registration errors, slice
    drop-out, smoothing and a spatially varying merge with the ground truth (ref :469-774)."""

    def __init__(self, prob_misreg_slice: float, slices_misreg_ratio: float, prob_misreg_stack: float, txy: float,
                 prob_merge: float, merge_params: ReconMergeParams, prob_smooth: float, prob_rm_slices: float,
                 rm_slices_min: float, rm_slices_max: float):
        self.prob_misreg_slice = prob_misreg_slice
        self.slices_misreg_ratio = slices_misreg_ratio
        self.prob_misreg_stack = prob_misreg_stack
        self.txy_stack = txy
        self.prob_merge = prob_merge
        self.merge_params = merge_params
        assert merge_params.merge_type in ["gaussian", "perlin"], (
            f"Merge type {merge_params.merge_type} not supported, only gaussian and perlin are supported.")
        self.prob_smooth = prob_smooth
        self.prob_rm_slices = prob_rm_slices
        self.rm_slices_min = rm_slices_min
        self.rm_slices_max = rm_slices_max

    def sample_seeds(self, genparams: dict = {}):
        """numpy draws: rand x3 (smooth, rm, misreg-slice gates) [, uniform (rm ratio)], rand (merge gate), then
        randint (n gaussians) or choice x2 (perlin res, octaves) (ref :523-560)."""
        self._smooth_volume_on = np.random.rand() < self.prob_smooth
        self._rm_slices_on = np.random.rand() < self.prob_rm_slices
        self._misreg_slice_on = np.random.rand() < self.prob_misreg_slice
        if "rm_slices_ratio" in genparams:
            self._rm_slices_ratio = genparams["rm_slices_ratio"]
        else:
            self._rm_slices_ratio = (np.random.uniform(self.rm_slices_min, self.rm_slices_max)
                                     if self._rm_slices_on else None)
        self._misreg_stack_on = []
        self._merge_volume_on = np.random.rand() < self.prob_merge
        mp = self.merge_params
        if mp.merge_type == "gaussian":
            self._ngaussians_merge = (genparams["ngaussians_merge"] if "ngaussians_merge" in genparams
                                      else np.random.randint(mp.gauss_ngaussians_min, mp.gauss_ngaussians_max))
        elif mp.merge_type == "perlin":
            self._res = genparams["res"] if "res" in genparams else np.random.choice(mp.perlin_res_list)
            self._octave = genparams["octave"] if "octave" in genparams else np.random.choice(mp.perlin_octaves_list)

    def get_seeds(self):
        seeds = {
            "smooth_volume_on": self._smooth_volume_on,
            "rm_slices_on": self._rm_slices_on,
            "rm_slices_ratio": self._rm_slices_ratio,
            "misreg_stack_on": self._misreg_stack_on,
            "misreg_slice_on": self._misreg_slice_on,
            "merge_volume_on": self._merge_volume_on,
        }
        if self.merge_params.merge_type == "gaussian":
            seeds["merge_type"] = "gaussian"
            seeds["ngaussians_merge"] = self._ngaussians_merge
        elif self.merge_params.merge_type == "perlin":
            seeds["merge_type"] = "perlin"
            seeds["res"] = self._res
            seeds["octave"] = self._octave
        return seeds

    def smooth_volume(self, volume):
        """3x3x3 mean filter, zero padded (ref :584-595): three box passes on the blur kernels."""
        if not self._smooth_volume_on:
            return volume
        v = volume.reshape(volume.shape[-3:]).contiguous()
        ones = np.ones(3, np.float32)
        v = K.blur_axis(K.blur_axis(v, 0, ones), 1, ones)
        return K.blur_axis(v, 2, np.full(3, 1.0 / 27.0, np.float32)).view(volume.shape)

    def misregistration_trf(self, positions, base_axisangle):
        """Per-stack in-plane shift + rotation error composed onto the slice transforms (ref :597-627).
        numpy draws per stack: rand gate, uniform x2, random_angle(n_in_stack)."""
        positions = positions.cpu()
        nslices = len(positions)
        rand_angle = torch.zeros((nslices, 6))
        for pos in torch.unique(positions[:, 1]):
            self._misreg_stack_on.append(np.random.rand() < self.prob_misreg_stack)
            if not self._misreg_stack_on[-1]:
                continue
            idx = torch.where(positions[:, 1] == pos)[0]
            tx = torch.ones(len(idx)) * np.random.uniform(-self.txy_stack, self.txy_stack)
            ty = torch.ones(len(idx)) * np.random.uniform(-self.txy_stack, self.txy_stack)
            rand_angle[idx, 3:] = random_angle(len(idx), restricted=True, device=None)
            rand_angle[idx, :3] = torch.stack((tx, ty, torch.zeros_like(tx)), -1)
        return RigidTransform(rand_angle, trans_first=True).compose(base_axisangle)

    def misregister_slices(self, trf, trf_gt):
        """Swap the ground-truth transform of (at most one) random slice for its initial guess (ref :629-647)."""
        trf1 = trf.axisangle()
        trf2 = trf_gt.axisangle().clone()
        if self._misreg_slice_on:
            idx_misreg = torch.randperm(trf2.shape[0])[: int(self.slices_misreg_ratio * trf2.shape[0])]
            idx_misreg = idx_misreg[:1]
            trf2[idx_misreg] = trf1[idx_misreg]
        return RigidTransform(trf2, trans_first=True)

    # ---- merge with the ground truth ------------------------------------------------------------------
    def _gaussian_centers(self, vol_mask):
        """`_ngaussians_merge` distinct voxels of the mask, as the reference's randperm over them (ref :661-664)."""
        m = vol_mask.reshape(vol_mask.shape[-3:]).contiguous()  # bool mask, or the float label map itself (> 0)
        count, select = K.nonzero_ranks(m, ">", 0.0)
        idx = _rng.distinct_ranks(count, self._ngaussians_merge)
        return select(idx)  # (k,3) int64 host, first-axis index first

    def get_merging_weights(self, shape, vol_mask=None):
        """The merge weight volume itself (ref :649-690); `merge_volumes` blends without materialising it."""
        mp = self.merge_params
        if vol_mask is not None and mp.merge_type == "gaussian":
            centers = self._gaussian_centers(vol_mask)
            sigmas = [torch.clamp(20 + 10 * torch.randn(1), 5, 40) for _ in range(len(centers))]
            return mog_3d_tensor(shape, centers=[tuple(c) for c in centers.tolist()], sigmas=sigmas, device=self.device)
        if mp.merge_type == "perlin":
            raw, mm = self._perlin_raw(shape)
            return K.blend(None, None, raw, w_mm=mm, increase=mp.perlin_increase_size, want_weight=True, want_out=False)[1]
        raise RuntimeError

    def _perlin_raw(self, shape):
        mp = self.merge_params
        plan = fractal_noise_plan(tuple(int(v) for v in shape), (self._res,) * 3, octaves=self._octave,
                                  persistence=mp.perlin_persistence, lacunarity=mp.perlin_lacunarity, device=self.device)
        return K.perlin_fractal(plan)

    def merge_volumes(self, vol_mask, volume, volume_gt, want_weight=True):
        """merged = w * volume + (1 - w) * volume_gt (ref :692-709).  Perlin weights are normalised inside the blend."""
        if not self._merge_volume_on:
            return volume, (torch.zeros_like(volume) if want_weight else None)
        shape = volume.shape[-3:]
        v = volume.reshape(shape).contiguous()
        gt = volume_gt.reshape(shape).float().contiguous()
        if self.merge_params.merge_type == "perlin":
            raw, mm = self._perlin_raw(shape)
            out = K.blend(gt, v, raw, w_mm=mm, increase=self.merge_params.perlin_increase_size, want_weight=want_weight)
        else:
            w = self.get_merging_weights(shape, vol_mask)
            out = (K.blend(gt, v, w), w) if want_weight else K.blend(gt, v, w)
        if want_weight:
            return out[0].view(volume.shape), out[1]
        return out.view(volume.shape), None

    def kept_slices_idx(self, nslices):
        if self._rm_slices_on:
            n = int(nslices * self._rm_slices_ratio)
            return torch.randperm(nslices)[n:]
        return torch.arange(nslices)

    def recon_psf(self, data, want_weight=True):
        params = {
            "psf": data["psf_rec"],
            "slice_shape": data["slice_shape"],
            "interp_psf": True,
            "res_s": data["resolution_slice"],
            "res_r": data["resolution_recon"],
            "s_thick": data["slice_thickness"],
            "volume_shape": data["volume_shape"],
        }
        rec = partial(PSFreconstruction, slices_mask=None, vol_mask=None, params=params)
        return self.__recon_volume(data, rec, want_weight)

    def __recon_volume(self, data, rec, want_weight=True):
        self.sample_seeds()
        self.device = data["stacks"].device
        trf = self.misregister_slices(data["transforms_angle"], data["transforms_gt_angle"])
        trf = self.misregistration_trf(data.get("positions_host", data["positions"]), trf)
        kept_idx = self.kept_slices_idx(data["stacks"].shape[0])
        mats = K._upload(trf.matrix()[kept_idx].contiguous(), self.device)
        volume = rec(mats, data["stacks"], slice_ids=kept_idx)  # the kept slices are addressed in place, not gathered
        volume = self.smooth_volume(volume)
        # the reference forms `mask = seg_gt > 0` here (:772); only the Gaussian merge reads it, and the
        # voxel-selection kernels apply the `> 0` test to the label map directly
        return self.merge_volumes(data["seg_gt"], volume, data["volume_gt"], want_weight)
