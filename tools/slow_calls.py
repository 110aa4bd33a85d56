#!/usr/bin/env python3
"""Which call makes an SR-artifact stage slow in some repetitions?  Every function of fetalsyngen_amd.kernels is wrapped
with synchronise + timer; per repetition: stage time, the sum inside kernels.*, and each call above --min-ms with its
tensor-argument shapes.  Diagnostic (synchronising).

    python tools/slow_calls.py --stage blur_cortex --reps 12
"""
import argparse
import json
import sys
import time
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fetalsyngen_amd import kernels as K  # noqa: E402
from fetalsyngen_amd import rng  # noqa: E402
from fetalsyngen_amd.generator.defaults import default_artifacts  # noqa: E402
from fetalsyngen_amd.phantom import make_segmentation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="blur_cortex")
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--min-ms", type=float, default=3.0)
    a = ap.parse_args()
    dev = "cuda:0"
    rng.set_mode("device")
    shape = (a.size,) * 3
    seg = torch.from_numpy(make_segmentation(shape)[0].astype(np.float32)).to(dev)
    img = (0.1 * seg + 0.05 * torch.rand(shape, device=dev)) * (seg > 0)
    img = img / img.max()
    st = default_artifacts(prob=1.0)[a.stage]
    calls, depth = [], [0]

    def wrap(name, fn):
        def inner(*args, **kw):
            if depth[0]:
                return fn(*args, **kw)
            depth[0] += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                torch.cuda.synchronize()
                depth[0] -= 1
                shapes = [tuple(x.shape) for x in args if isinstance(x, (torch.Tensor, np.ndarray))]
                calls.append((name, (time.perf_counter() - t0) * 1e3, shapes))
        return inner

    for nm, fn in list(vars(K).items()):
        if isinstance(fn, types.FunctionType) and fn.__module__ == K.__name__ and not nm.startswith("__"):
            setattr(K, nm, wrap(nm, fn))
    for rep in range(a.reps + 1):
        np.random.seed(100 + rep)
        torch.manual_seed(100 + rep)
        calls.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _y, meta = st(img, seg, dev, {}, resolution=[0.5, 0.5, 0.5])
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) * 1e3
        inside = sum(c[1] for c in calls)
        print(json.dumps({"rep": rep, "ms": round(tot, 1), "in_kernels_py": round(inside, 1), "outside": round(tot - inside, 1),
                          "slow": [(c[0], round(c[1], 1), c[2]) for c in calls if c[1] >= a.min_ms]}), flush=True)


if __name__ == "__main__":
    main()
