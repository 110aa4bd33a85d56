"""Shared helpers for the parity tests (not a test module)."""
import numpy as np
import torch

from oracle import fsg_oracle as O

DEFAULT_SEED_LABELS = O.DEFAULT_SEED_LABELS
DEFAULT_GEN_CLASSES = O.DEFAULT_GEN_CLASSES


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_generator(shape, device, prob=1.0, nonlin_scale=(0.03, 0.06), size=None, bf_scale=(0.004, 0.02),
                   resolution=(0.5, 0.5, 0.5), res_range=(0.5, 1.5), rng="reference", artifacts=None):
    """fetalsyngen_amd generator configured like tests/golden/make_golden.py:build_generator."""
    from fetalsyngen_amd.generator.model import FetalSynthGen
    from fetalsyngen_amd.generator.intensity.rand_gmm import ImageFromSeeds
    from fetalsyngen_amd.generator.deformation.affine_nonrigid import SpatialDeformation
    from fetalsyngen_amd.generator.augmentation.synthseg import RandResample, RandBiasField, RandNoise, RandGamma

    size = list(shape) if size is None else list(size)
    return FetalSynthGen(
        shape=list(shape), resolution=list(resolution), device=device,
        intensity_generator=ImageFromSeeds(1, 6, DEFAULT_SEED_LABELS, DEFAULT_GEN_CLASSES),
        spatial_deform=SpatialDeformation(20, 0.02, 0.1, size, prob, True, nonlin_scale[0], nonlin_scale[1], 4, 0.5,
                                          device),
        resampler=RandResample(prob, res_range[0], res_range[1]),
        bias_field=RandBiasField(prob, bf_scale[0], bf_scale[1], 0.01, 0.3),
        noise=RandNoise(prob, 5, 15),
        gamma=RandGamma(prob, 0.1),
        rng=rng,
        **(artifacts or {}),
    )


E2E = {
    "e2e_32_s0": dict(),
    "e2e_32_s1": dict(nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2)),
    "e2e_32_s2": dict(prob=0.5),
    "e2e_48_s0": dict(nonlin_scale=(0.08, 0.2)),
    "e2e_nc_s1": dict(nonlin_scale=(0.1, 0.2)),
    "e2e_sz_s4": dict(nonlin_scale=(0.1, 0.2), size=(32, 32, 32)),
}


def default_artifacts(prob=1.0, merge_type="perlin"):
    from fetalsyngen_amd.generator.defaults import default_artifacts as _d

    return _d(prob=prob, merge_type=merge_type)
