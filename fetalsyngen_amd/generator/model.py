"""Generator orchestration: mirror of `fetalsyngen.generator.model.FetalSynthGen`
(reference model.py:27-276).

Same constructor keywords, attributes (`shape, resolution, intensity_generator, spatial_deform,
resampled, biasfield, gamma, noise, artifacts, device`), methods and return tuples, same
`synth_params` schema, same stage order and the same consumption order of the numpy / torch global
generators.  `sample()` first collects every stage's random plan on the host, uploads all small
arrays in one copy, then runs the fused kernel sequence

    gmm -> coords min/max -> warp(+gamma+bias, labels) -> blur x,y,z -> resample+noise
        -> zoom min/max -> zoom+normalise

`generate()` / `augment()` remain individually callable (stage by stage, un-fused).
The optional SR-artifact stages (`blur_cortex`, `struct_noise`, `simulate_motion`, `boundaries`) are
outside this package's scope; any callable with the reference's artifact signature is
still applied in the reference's place.
"""
from __future__ import annotations

from typing import Iterable

import numpy as np
import torch

from .. import kernels as K
from .. import rng as _rng
from .. import tables as T
from .augmentation.synthseg import RandBiasField, RandGamma, RandNoise, RandResample
from .deformation.affine_nonrigid import SpatialDeformation
from .intensity.rand_gmm import ImageFromSeeds


class FetalSynthGen:
    def __init__(
        self,
        shape: Iterable[int],
        resolution: Iterable[float],
        device: str,
        intensity_generator: ImageFromSeeds,
        spatial_deform: SpatialDeformation,
        resampler: RandResample,
        bias_field: RandBiasField,
        noise: RandNoise,
        gamma: RandGamma,
        blur_cortex=None,
        struct_noise=None,
        simulate_motion=None,
        boundaries=None,
        rng: str | None = None,
    ):
        if not str(device).startswith("cuda"):
            raise RuntimeError(
                f"fetalsyngen_amd.FetalSynthGen runs on an MI355X only (device='cuda:N'), got device={device!r}; "
                "there is no CPU fallback"
            )
        self.shape = shape
        self.resolution = resolution
        self.intensity_generator = intensity_generator
        self.spatial_deform = spatial_deform
        self.resampled = resampler
        self.biasfield = bias_field
        self.gamma = gamma
        self.noise = noise
        self.artifacts = {
            "blur_cortex": blur_cortex,
            "struct_noise": struct_noise,
            "simulate_motion": simulate_motion,
            "boundaries": boundaries,
        }
        self.device = device
        self.rng = rng  # None: module default (fetalsyngen_amd.rng.get_mode())

    def prewarm(self, shape=None) -> int:
        """Build and upload every per-axis table this configuration can ask for (the low-res size of
        RandResample takes at most shape*(1 - min/max resolution) distinct values per axis; the coarse
        deformation / bias grids a handful).  Optional: tables are otherwise cached on first use.
        Returns the number of tables now resident."""
        shape = tuple(int(v) for v in (shape or self.shape))
        res = np.array(self.resolution, dtype=np.float64)
        rs, sd, bf = self.resampled, self.spatial_deform, self.biasfield
        n = 0
        for a in range(3):
            size = shape[a]
            lo = int(size * res[a] / max(rs.max_resolution, res[a]))
            for m in range(max(lo - 1, 1), size + 1):
                K._device_table(T._resample_axis_table(m, size), self.device)
                f = np.float64(m) / np.float64(size)
                K._device_table(T.zoom_table(m, float(1 / f), int(np.round(m * (1 / f)))), self.device)
                n += 2
            for lo_s, hi_s in ((sd.nonlin_scale_min, sd.nonlin_scale_max), (bf.scale_min, bf.scale_max)):
                for s_ in range(max(int(np.floor(lo_s * size)) - 1, 1), int(np.ceil(hi_s * size)) + 2):
                    K._device_table(T.zoom_table(s_, float(np.float64(size) / np.float64(s_)), size), self.device)
                    n += 1
        return n

    def _validated_genparams(self, d):
        if not isinstance(d, dict):
            return d
        return {k: self._validated_genparams(v) for k, v in d.items() if v is not None}

    # ---- stage-by-stage API ---------------------------------------------------------------------
    def _intensity_prior(self, image):
        img = image.to(self.device).float().contiguous()
        return K.scale(img, K.reduce_minmax(img), mode=2)  # (x-min)/(max-min)*255, ref :138

    def generate(self, image, segmentation, seeds, genparams: dict = {}):
        with _rng.use(self.rng):
            if seeds is not None:
                seeds, selected_seeds = self.intensity_generator.load_seeds(
                    seeds=seeds, genparams=genparams.get("selected_seeds", {}))
                output, seed_intensities = self.intensity_generator.sample_intensities(
                    seeds=seeds, device=self.device, genparams=genparams.get("seed_intensities", {}))
            else:
                if image is None:
                    raise ValueError(
                        "If no seeds are passed, an image must be loaded to be used as intensity prior!")
                output = self._intensity_prior(image)
                selected_seeds, seed_intensities = {}, {}
            segmentation = segmentation.to(self.device)
            image = image.to(self.device) if image is not None else None
            image, segmentation, output, deform_params = self.spatial_deform.deform(
                image=image, segmentation=segmentation, output=output,
                genparams=genparams.get("deform_params", {}))
        return output, segmentation, image, {
            "selected_seeds": selected_seeds,
            "seed_intensities": seed_intensities,
            "deform_params": deform_params,
        }

    def _apply_artifacts(self, output, segmentation, genparams):
        artifacts = {}
        for name, artifact in self.artifacts.items():
            if artifact is not None:
                output, metadata = artifact(output, segmentation, self.device, genparams.get("artifact_params", {}),
                                            resolution=self.resolution)
                artifacts[name] = metadata
        return output, artifacts

    def augment(self, image, segmentation, genparams: dict = {}):
        with _rng.use(self.rng):
            output, gamma_params = self.gamma(image, self.device, genparams=genparams.get("gamma_params", {}))
            output, bf_params = self.biasfield(output, self.device, genparams=genparams.get("bf_params", {}))
            output, factors, resample_params = self.resampled(
                output, np.array(self.resolution), self.device, genparams=genparams.get("resample_params", {}))
            output, noise_params = self.noise(output, self.device, genparams=genparams.get("noise_params", {}))
            output = self.resampled.resize_back(output, factors)
            output, artifacts = self._apply_artifacts(output, segmentation, genparams)
        return output, {
            "gamma_params": gamma_params,
            "bf_params": bf_params,
            "resample_params": resample_params,
            "noise_params": noise_params,
            "artifacts": artifacts,
        }

    # ---- fused path -----------------------------------------------------------------------------
    def sample(self, image, segmentation, seeds, genparams: dict = {}):
        out, seg, img, params = self._pipeline(image, segmentation, seeds, genparams, scale01=False)
        return out, seg, img, params

    def _pipeline(self, image, segmentation, seeds, genparams, scale01: bool, segmentation_u8=None):
        if genparams:
            genparams = self._validated_genparams(genparams)
        dev = self.device
        ig, sd = self.intensity_generator, self.spatial_deform
        with _rng.use(self.rng):
            # ---------------- host: every random draw, in the reference's order ----------------
            labels, label_parts, gmm_plan, selected_seeds = None, None, None, {}
            if seeds is not None:
                gs = genparams.get("selected_seeds", {})
                if hasattr(seeds, "parts") and ig.meta_labels <= 4:  # device-resident SeedBank
                    m2s = ig.draw_subclusters(gs)
                    label_parts, selected_seeds = seeds.parts(m2s), {"mlabel2subclusters": m2s}
                    shape = tuple(label_parts[0].shape)
                else:
                    labels, selected_seeds = ig.load_seeds(seeds=seeds, genparams=gs)
                    shape = tuple(labels.shape)
                gmm_plan = ig.plan_intensities(shape, genparams.get("seed_intensities", {}))
            else:
                if image is None:
                    raise ValueError(
                        "If no seeds are passed, an image must be loaded to be used as intensity prior!")
                shape = tuple(image.shape)
            dplan = sd.plan(shape, random_shift=True, genparams=genparams.get("deform_params", {}))
            g = self.gamma.plan(genparams.get("gamma_params", {}))
            bplan = self.biasfield.plan(shape, genparams.get("bf_params", {}))
            rplan = self.resampled.plan(shape, np.array(self.resolution), genparams.get("resample_params", {}))
            low_shape = rplan.new_size if rplan.active else shape
            nplan = self.noise.plan(low_shape, genparams.get("noise_params", {}))

            # ---------------- one upload of all small arrays ------------------------------------
            arena = T.Arena()
            sb = sd.make_spec(dplan, shape, flip_in_kernel=True, arena=arena) if dplan.active else None
            bias_tabs, bias_off = None, None
            if bplan.active:
                bias_tabs = K.DeviceTables(self.biasfield.tables(bplan, shape), dev, arena)
                bias_off = arena.add(bplan.grid.numpy())
            rs_tabs = back_tabs = None
            if rplan.active:
                rs_tabs = K.DeviceTables(rplan.tabs, dev, arena)
                bt, new = T.zoom_tables(rplan.new_size, 1 / np.asarray(rplan.factors))
                back_tabs = K.DeviceTables(bt, dev, arena)
            gm_off = None
            if gmm_plan is not None:
                gm_off = (arena.add(gmm_plan.mus.numpy()), arena.add(gmm_plan.sigmas.numpy()), gmm_plan.mus.numel())
            arena.upload(dev)

            def f32_view(off, shp):
                n = int(np.prod(shp))
                return arena.dev[off : off + 4 * n].view(torch.float32).view(tuple(shp))

            # ---------------- device ----------------------------------------------------------
            seed_intensities = {}
            if gmm_plan is not None:
                mus, sigmas = f32_view(gm_off[0], (gm_off[2],)), f32_view(gm_off[1], (gm_off[2],))
                f = gmm_plan.field
                z = f.device_tensor(dev) if f.host is not None else None
                if label_parts is not None:
                    output = K.gmm_sample_parts(label_parts, mus, sigmas, noise=z, seed=f.seed or 0,
                                                stream_id=f.stream_id)
                else:
                    if labels.dtype not in (torch.uint8, torch.int64):
                        labels = labels.long()
                    labels = labels.to(dev).contiguous()
                    output = K.gmm_sample(labels, mus, sigmas, noise=z, seed=f.seed or 0, stream_id=f.stream_id)
                seed_intensities = {"mus": mus, "sigmas": sigmas}
            else:
                output = self._intensity_prior(image)

            bias_dev = f32_view(bias_off, tuple(bplan.grid.shape)) if bplan.active else None
            gam = float(g) if g is not None else None
            image = image.to(dev) if image is not None else None
            # one init launch for every min/max key of the sample: [min x,y,z | zoom min] [zoom max | unused x3]
            mm8 = K.new_minmax(dev, 4, 4)
            if dplan.active:
                spec = sb.build()
                image, segmentation, output = sd.run(dplan, image, segmentation, output, spec=spec,
                                                     mm6=K.coords_floormin(spec, mm8), gamma=gam, bias=bias_dev,
                                                     bias_tabs=bias_tabs, segmentation_u8=segmentation_u8)
            else:
                segmentation = segmentation.to(dev)
                if gam is not None:
                    output = K.gamma(output, gam)
                if bplan.active:
                    output = K.bias_mul(output, bias_dev, bias_tabs)

            has_art = any(a is not None for a in self.artifacts.values())
            fuse_scale = scale01 and not has_art
            f = nplan.field if nplan.active else None
            z = f.device_tensor(dev) if (f is not None and f.host is not None) else None
            if rplan.active:
                blurred = self.resampled.blur(output, rplan.stds)
                low = K.resample_noise(blurred, rs_tabs, noise_std=nplan.std32 if nplan.active else 0.0, noise=z,
                                       seed=(f.seed if (f is not None and f.host is None) else None),
                                       stream_id=f.stream_id if f is not None else 0)
                mm2 = K.zoom_minmax(low, back_tabs, mm=mm8[3:5])
                output = K.zoom_normalise(low, back_tabs, mm2, mode=1 if fuse_scale else 0)
            else:
                if nplan.active:
                    output = K.add_noise(output, nplan.std32, noise=z, seed=f.seed or 0, stream_id=f.stream_id)
                if fuse_scale:
                    output = K.scale(output, K.reduce_minmax(output), mode=1)
            output, artifacts = self._apply_artifacts(output, segmentation, genparams)
            if scale01 and has_art:
                output = K.scale(output.contiguous(), K.reduce_minmax(output.contiguous()), mode=1)

        synth_params = {
            "selected_seeds": selected_seeds,
            "seed_intensities": seed_intensities,
            "deform_params": dplan.params,
            "gamma_params": {"gamma": g},
            "bf_params": bplan.params,
            "resample_params": {"spacing": rplan.spacing.tolist() if rplan.active else None},
            "noise_params": {"noise_std": nplan.std32 if nplan.active else None},
            "artifacts": artifacts,
        }
        return output, segmentation, image, synth_params
