"""Drop-in boundary through the reference's own config (SURVEY 8(b) "Hydra config keys").

The kwargs come from tests/golden/generator_default.json / synth_train.json, written by
tests/golden/make_config_fixture.py from configs/dataset/generator/default.yaml:1-142 and synth_train.yaml:1-9.
hydra is absent: `_instantiate` below is the documented behaviour of `hydra.utils.instantiate` for this file
(recursive: a mapping with `_target_` -> import the dotted path and call it with the instantiated kwargs).
With `compat.install()` every `fetalsyngen.*` target must resolve to the MI355X implementation and accept
exactly the reference's keyword names.
"""
import importlib
import inspect
import json
import sys
from pathlib import Path

import pytest

GOLD = Path(__file__).parent / "golden"


def _instantiate(node):
    if isinstance(node, dict):
        kw = {k: _instantiate(v) for k, v in node.items() if k != "_target_"}
        if "_target_" in node:
            mod, _, name = node["_target_"].rpartition(".")
            return getattr(importlib.import_module(mod), name)(**kw)
        return kw
    if isinstance(node, list):
        return [_instantiate(v) for v in node]
    return node


def _targets(node, out):
    if isinstance(node, dict):
        if "_target_" in node:
            out.append((node["_target_"], sorted(k for k in node if k != "_target_")))
        for v in node.values():
            _targets(v, out)
    return out


@pytest.fixture()
def alias():
    from fetalsyngen_amd import compat

    saved = {k: v for k, v in sys.modules.items() if k == "fetalsyngen" or k.startswith("fetalsyngen.")}
    compat.install()
    yield
    for k in [k for k in sys.modules if k == "fetalsyngen" or k.startswith("fetalsyngen.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_every_target_resolves_and_accepts_the_yaml_kwargs(alias):
    cfg = json.loads((GOLD / "generator_default.json").read_text())
    targets = _targets(cfg, [])
    assert len(targets) == 15  # generator + 6 stages + 4 artifact stages + 4 parameter dataclasses
    for target, keys in targets:
        mod, _, name = target.rpartition(".")
        cls = getattr(importlib.import_module(mod), name)
        assert cls.__module__.startswith("fetalsyngen_amd."), (target, cls.__module__)
        params = inspect.signature(cls.__init__).parameters
        missing = [k for k in keys if k not in params]
        assert not missing, f"{target}: YAML keys {missing} are not constructor keywords"
        required = [k for k, p in params.items()
                    if k != "self" and p.default is inspect.Parameter.empty
                    and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
        unset = [k for k in required if k not in keys]
        assert not unset, f"{target}: constructor needs {unset}, which the YAML does not give"


def test_default_yaml_instantiates_the_whole_generator(alias):
    cfg = json.loads((GOLD / "generator_default.json").read_text())
    gen = _instantiate(cfg)
    from fetalsyngen_amd.generator.model import FetalSynthGen

    assert type(gen) is FetalSynthGen
    assert list(gen.shape) == [256, 256, 256] and list(gen.resolution) == [0.5, 0.5, 0.5] and gen.device == "cuda:0"
    # `${..device}` / `${..shape}` reached the nested stage
    assert gen.spatial_deform.device == "cuda:0" and list(gen.spatial_deform.size) == [256, 256, 256]
    assert gen.intensity_generator.min_subclusters == 1 and gen.intensity_generator.max_subclusters == 6
    assert gen.resampled.max_resolution == 1.5 and gen.biasfield.std_max == 0.3
    assert gen.gamma.gamma_std == 0.1 and gen.noise.std_max == 15
    assert set(gen.artifacts) == {"blur_cortex", "struct_noise", "simulate_motion", "boundaries"}
    assert all(a is not None for a in gen.artifacts.values())
    sm = gen.artifacts["simulate_motion"]
    assert sm.scanner_args.max_num_slices == 250 and sm.recon_args.merge_params.perlin_increase_size == 0.25


def test_synth_train_yaml_dataset_target(alias, tmp_path):
    """synth_train.yaml: FetalSynthDataset with the generator mounted under `generator` (paths replaced by a tiny tree)."""
    from tests.util_bids import write_tree

    cfg = json.loads((GOLD / "synth_train.json").read_text())
    assert cfg["_target_"] == "fetalsyngen.data.datasets.FetalSynthDataset"
    assert set(cfg) == {"_target_", "bids_path", "seed_path", "sub_list", "load_image", "image_as_intensity", "generator"}
    bids, seeds = write_tree(tmp_path, (16, 16, 16), ["sub-a", "sub-b"])
    cfg["bids_path"], cfg["seed_path"] = str(bids), str(seeds)
    ds = _instantiate(cfg)
    assert len(ds) == 2 and ds.generator.device == "cuda:0"
    with pytest.raises(FileNotFoundError):
        _instantiate({**cfg, "seed_path": str(tmp_path / "nope")})


def test_install_refuses_to_shadow_a_real_package():
    from fetalsyngen_amd import compat

    import types

    saved = {k: v for k, v in sys.modules.items() if k == "fetalsyngen" or k.startswith("fetalsyngen.")}
    sys.modules["fetalsyngen"] = types.ModuleType("fetalsyngen")
    try:
        with pytest.raises(RuntimeError):
            compat.install()
        compat.install(force=True)
        assert sys.modules["fetalsyngen"].__fsg_alias__
    finally:
        for k in [k for k in sys.modules if k == "fetalsyngen" or k.startswith("fetalsyngen.")]:
            del sys.modules[k]
        sys.modules.update(saved)
