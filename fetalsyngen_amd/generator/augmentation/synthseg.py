"""Intensity augmentations: mirrors of `RandResample`, `RandBiasField`, `RandNoise`, `RandGamma`
(`fetalsyngen.generator.augmentation.synthseg`, reference synthseg.py:14-275).

Each class keeps the reference's constructor and `__call__` contract.  A call is split into
`plan()` -- the numpy/torch draws, in the reference's order, no GPU work -- and a kernel launch, so that
`FetalSynthGen` can collect all plans of a sample first and then run the fused kernels.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import kernels as K
from ... import rng
from ... import tables as T


class RandTransform:
    """Nominal base class (the reference derives from monai.transforms.Transform, ref :14-22)."""

    def __call__(self, *args, **kwargs):
        raise NotImplementedError

    def random_uniform(self, min_val, max_val):
        return np.random.uniform(min_val, max_val)


# ---------------------------------------------------------------------------------------------
class ResamplePlan:
    __slots__ = ("active", "spacing", "stds", "new_size", "factors", "tabs")

    def __init__(self):
        self.active, self.spacing, self.stds, self.new_size, self.factors, self.tabs = False, None, None, None, None, None


class RandResample(RandTransform):
    def __init__(self, prob: float, min_resolution: float, max_resolution: float):
        self.prob = prob
        self.min_resolution = min_resolution
        self.max_resolution = max_resolution

    def plan(self, in_shape, input_resolution, genparams: dict = {}) -> ResamplePlan:
        """rand() gate, uniform() spacing, rand() blur jitter (ref :63-80)."""
        p = ResamplePlan()
        gate = np.random.rand() < self.prob
        if not (gate or "spacing" in genparams):
            return p
        p.active = True
        if "spacing" in genparams:
            spacing = np.array(genparams["spacing"])
        else:
            spacing = np.array([1.0, 1.0, 1.0]) * self.random_uniform(self.min_resolution, self.max_resolution)
        p.spacing = spacing
        p.stds, p.new_size, p.factors, p.tabs = T.resample_plan(tuple(in_shape), input_resolution, spacing,
                                                                np.random.rand())
        return p

    @staticmethod
    def blur(output, stds):
        """x pass, then y and z passes (one fused launch when both are active and the shape allows; same result)."""
        if stds[0] > 0:
            output = K.blur_axis(output.contiguous(), 0, T.gaussian_taps(float(stds[0])))
        if stds[1] > 0 and stds[2] > 0:
            fused = K.blur_yz(output.contiguous(), T.gaussian_taps(float(stds[1])), T.gaussian_taps(float(stds[2])))
            if fused is not None:
                return fused
        for axis in (1, 2):
            if stds[axis] > 0:
                output = K.blur_axis(output.contiguous(), axis, T.gaussian_taps(float(stds[axis])))
        return output

    def __call__(self, output, input_resolution, device, genparams: dict = {}):
        p = self.plan(output.shape, input_resolution, genparams)
        if not p.active:
            return output, None, {"spacing": None}
        small = self.blur_resample(output.contiguous(), p.stds, K.DeviceTables(p.tabs, output.device))
        return small, p.factors, {"spacing": p.spacing.tolist()}

    @classmethod
    def blur_resample(cls, output, stds, tabs, **noise):
        """Blur + down-sampling (+ the noise epilogue): the fused pair of launches (csrc/fsg_blur_rs.hip) where the
        configuration allows -- the kernels `FetalSynthGen.sample` runs -- else the three blur passes and K7."""
        if all(s_ > 0 for s_ in stds):
            small = K.blur_resample(output, tabs, [T.gaussian_taps(float(s_)) for s_ in stds], **noise)
            if small is not None:
                return small
        return K.resample_noise(cls.blur(output, stds).contiguous(), tabs, **noise)

    def resize_back(self, output_resized, factors):
        """Zoom by 1/factors and divide by the global max (ref :109-114): two launches, the zoomed
        volume is written once, already normalised."""
        if factors is None:
            return output_resized
        tabs, _ = T.zoom_tables(output_resized.shape, 1 / np.asarray(factors))
        dt = K.DeviceTables(tabs, output_resized.device)
        src = output_resized.contiguous()
        return K.zoom_normalise(src, dt, K.zoom_minmax(src, dt), mode=0)


# ---------------------------------------------------------------------------------------------
class BiasPlan:
    __slots__ = ("active", "grid", "params")

    def __init__(self):
        self.active, self.grid = False, None
        self.params = {"bf_scale": None, "bf_std": None, "bf_size": None}


class RandBiasField(RandTransform):
    def __init__(self, prob: float, scale_min: float, scale_max: float, std_min: float, std_max: float):
        self.prob = prob
        self.scale_min = scale_min
        self.scale_max = scale_max
        self.std_min = std_min
        self.std_max = std_max

    def plan(self, image_size, genparams: dict = {}) -> BiasPlan:
        """rand() gate, rand(1) scale, rand(1) std, torch.randn(coarse grid) (ref :157-176)."""
        p = BiasPlan()
        gate = np.random.rand() < self.prob
        if not (gate or len(genparams.keys()) > 0):
            return p
        p.active = True
        scale = genparams["bf_scale"] if "bf_scale" in genparams else (
            self.scale_min + np.random.rand(1) * (self.scale_max - self.scale_min))
        size = np.maximum(np.round(scale * np.array(tuple(image_size))).astype(int), 1).tolist()
        std = genparams["bf_std"] if "bf_std" in genparams else (
            self.std_min + (self.std_max - self.std_min) * np.random.rand(1))
        std32 = np.asarray(std, dtype=np.float32)  # same rounding as torch.tensor(std, dtype=float32)
        p.grid = torch.from_numpy(std32 * torch.randn(size, dtype=torch.float32).numpy())
        p.params = {"bf_scale": scale, "bf_std": std, "bf_size": size}
        return p

    @staticmethod
    def tables(plan: BiasPlan, image_size):
        g = plan.grid
        tabs, new = T.zoom_tables_between(tuple(g.shape), tuple(int(v) for v in image_size))  # factor = size / grid
        if new != tuple(int(v) for v in image_size):
            raise ValueError("bias grid does not zoom to the image size")
        return tabs

    def __call__(self, output, device, genparams: dict = {}):
        p = self.plan(output.shape, genparams)
        if not p.active:
            return output, p.params
        dt = K.DeviceTables(self.tables(p, output.shape), output.device)
        return K.bias_mul(output.contiguous(), p.grid.to(output.device), dt), p.params


# ---------------------------------------------------------------------------------------------
class NoisePlan:
    __slots__ = ("active", "std32", "field")

    def __init__(self):
        self.active, self.std32, self.field = False, None, None


class RandNoise(RandTransform):
    def __init__(self, prob: float, std_min: float, std_max: float):
        self.prob = prob
        self.std_min = std_min
        self.std_max = std_max

    def plan(self, shape, genparams: dict = {}) -> NoisePlan:
        """rand() gate, rand(1) std, Gaussian field of `shape` (ref :218-232)."""
        p = NoisePlan()
        gate = np.random.rand() < self.prob
        if not (gate or "noise_std" in genparams):
            return p
        p.active = True
        std = genparams["noise_std"] if "noise_std" in genparams else (
            self.std_min + (self.std_max - self.std_min) * np.random.rand(1))
        p.std32 = float(np.asarray(std, dtype=np.float32).reshape(-1)[0])  # == torch.tensor(std, f32).item()
        p.field = rng.normal_field(tuple(shape), stream_id=2)
        return p

    def __call__(self, output, device, genparams: dict = {}):
        p = self.plan(output.shape, genparams)
        if not p.active:
            return output, {"noise_std": None}
        f = p.field
        noise = f.device_tensor(output.device) if f.host is not None else None
        out = K.add_noise(output.contiguous(), p.std32, noise=noise, seed=f.seed or 0, stream_id=f.stream_id)
        return out, {"noise_std": p.std32}


# ---------------------------------------------------------------------------------------------
class RandGamma(RandTransform):
    def __init__(self, prob: float, gamma_std: float):
        self.prob = prob
        self.gamma_std = gamma_std

    def plan(self, genparams: dict = {}):
        """rand() gate, randn(1) exponent (ref :263-268); returns gamma or None."""
        gate = np.random.rand() < self.prob
        if not (gate or "gamma" in genparams):
            return None
        return genparams["gamma"] if "gamma" in genparams else np.exp(self.gamma_std * np.random.randn(1)[0])

    def __call__(self, output, device, genparams: dict = {}):
        g = self.plan(genparams)
        if g is None:
            return output, {"gamma": None}
        return K.gamma(output.contiguous(), float(g)), {"gamma": g}
