"""Alias package: make `fetalsyngen.<...>` import paths (and therefore the reference's Hydra
`_target_` strings, configs/dataset/generator/default.yaml:1-56) resolve to fetalsyngen_amd."""
from __future__ import annotations

import importlib
import sys
import types

_MODULES = [
    "generator",
    "generator.model",
    "generator.intensity",
    "generator.intensity.rand_gmm",
    "generator.deformation",
    "generator.deformation.affine_nonrigid",
    "generator.augmentation",
    "generator.augmentation.synthseg",
    "generator.augmentation.artifacts",
    "generator.artifacts",
    "generator.artifacts.utils",
    "generator.artifacts.simulate_reco",
    "generator.artifacts.svort",
    "data",
    "data.datasets",
    "utils",
    "utils.generation",
    "utils.image_reading",
]


def install(force: bool = False) -> None:
    if "fetalsyngen" in sys.modules and not force:
        existing = sys.modules["fetalsyngen"]
        if not getattr(existing, "__fsg_alias__", False):
            raise RuntimeError("a real `fetalsyngen` package is already imported; pass force=True to shadow it")
    root = types.ModuleType("fetalsyngen")
    root.__fsg_alias__ = True
    root.__path__ = []
    sys.modules["fetalsyngen"] = root
    for name in _MODULES:
        mod = importlib.import_module(f"fetalsyngen_amd.{name}")
        sys.modules[f"fetalsyngen.{name}"] = mod
        parent, _, leaf = name.rpartition(".")
        setattr(sys.modules["fetalsyngen" + ("." + parent if parent else "")], leaf, mod)
