#!/usr/bin/env python3
"""The reconstruction scatter (sa_adjoint_nn_lds_kernel) on the problems SimulateMotion really hands it at 384^3, under the
tuning knobs of fsg_slice_acq_set_tuning (LDS accumulator cells, PSF planes per chunk (0 = derived), 16x16-tile extent).
Captures the adjoint calls of a few stage repetitions, then times each under every setting (ms, min of 2).

    python tools/sa_adjoint_sweep.py [--reps 6]
"""
import argparse
import itertools
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fetalsyngen_amd import _lib  # noqa: E402
from fetalsyngen_amd import kernels as K  # noqa: E402
from fetalsyngen_amd import rng  # noqa: E402
from fetalsyngen_amd.generator.defaults import default_artifacts  # noqa: E402
from fetalsyngen_amd.phantom import make_segmentation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--default-only", action="store_true", help="time the default setting only")
    a = ap.parse_args()
    dev = "cuda:0"
    rng.set_mode("device")
    shape = (a.size,) * 3
    seg = torch.from_numpy(make_segmentation(shape)[0].astype(np.float32)).to(dev)
    img = (0.1 * seg + 0.05 * torch.rand(shape, device=dev)) * (seg > 0)
    img = img / img.max()
    st = default_artifacts(prob=1.0)["simulate_motion"]
    captured = []
    orig = K.slice_acq_adjoint

    def grab(*args, **kw):
        captured.append((args, kw))
        return orig(*args, **kw)

    K.slice_acq_adjoint = grab
    for rep in range(a.reps + 1):
        np.random.seed(100 + rep)
        torch.manual_seed(100 + rep)
        st(img, seg, dev, {}, resolution=[0.5, 0.5, 0.5])
    torch.cuda.synchronize()
    K.slice_acq_adjoint = orig
    lib = _lib.load()

    def timed(args, kw):
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            orig(*args, **kw)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best

    settings = [] if a.default_only else list(itertools.product((3072, 6144, 12288, 18432), (0, 1, 2, 4), (0, 20, 40)))
    for args, kw in captured:
        tr, psf, slices = args[0], args[1], args[2]
        res = float(args[6])
        lib.fsg_slice_acq_set_tuning(3072, 0, 20)
        orig(*args, **kw)
        base = timed(args, kw)
        rows = []
        for cap, zc, t16 in settings:
            lib.fsg_slice_acq_set_tuning(cap, zc, t16)
            rows.append((round(timed(args, kw), 2), cap, zc, t16))
        lib.fsg_slice_acq_set_tuning(3072, 0, 20)
        rows.sort()
        print(json.dumps({"slices": list(slices.shape), "n": int(tr.shape[0]), "psf": list(psf.shape), "taps": int((psf > 0).sum()),
                          "pitch": round(res, 3), "default_ms": round(base, 2), "best": rows[:5], "worst": rows[-1] if rows else None}), flush=True)


if __name__ == "__main__":
    main()
