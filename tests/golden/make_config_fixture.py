#!/usr/bin/env python3
"""Turn the reference's Hydra generator config into a JSON fixture of constructor kwargs.

Runs only in the build container (reads /root/reference/configs).  hydra-core / omegaconf are absent, so the
two OmegaConf features the file uses are resolved here: the `defaults:` list of synth_train.yaml (one entry,
`generator/default`, mounted under the key `generator`) and relative interpolations `${..key}` (value of `key`
in the parent mapping).  Output: tests/golden/generator_default.json, tests/golden/synth_train.json --
what `hydra.utils.instantiate` would receive (reference configs/dataset/generator/default.yaml:1-142,
configs/dataset/synth_train.yaml:1-9).

Usage: python tests/golden/make_config_fixture.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import json
import re
from pathlib import Path

import yaml

HERE = Path(__file__).resolve().parent
_INTERP = re.compile(r"^\$\{(\.+)([A-Za-z_][A-Za-z0-9_]*)\}$")


def resolve(node, parents=()):
    """Resolve `${..key}` (dots = levels up from the node that holds the value, OmegaConf relative syntax)."""
    if isinstance(node, dict):
        return {k: resolve(v, parents + (node,)) for k, v in node.items()}
    if isinstance(node, list):
        return [resolve(v, parents) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node.strip())
        if m:
            up = len(m.group(1))  # "." = the mapping holding the value, ".." = its parent, ...
            holder = parents[len(parents) - up]
            return resolve(holder[m.group(2)], parents[: len(parents) - up])
    return node


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    cfg_dir = Path(args.ref) / "configs" / "dataset"
    gen = resolve(yaml.load((cfg_dir / "generator" / "default.yaml").read_text(), Loader=yaml.SafeLoader))
    (HERE / "generator_default.json").write_text(json.dumps(gen, indent=1) + "\n")
    ds = yaml.load((cfg_dir / "synth_train.yaml").read_text(), Loader=yaml.SafeLoader)
    defaults = ds.pop("defaults")
    assert defaults == ["generator/default"], defaults
    ds["generator"] = gen
    (HERE / "synth_train.json").write_text(json.dumps(resolve(ds), indent=1) + "\n")
    print("wrote generator_default.json, synth_train.json")


if __name__ == "__main__":
    main()
