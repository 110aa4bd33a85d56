#!/usr/bin/env python3
"""Per-repetition breakdown of SimulateMotion at 384^3 (tools/sr_stage_bench.py's set-up): wall time of the stage, and -- in a
second, synchronising pass over the same seeds -- the time inside the slice-acquisition forward / adjoint entry points with the
problem each call had (slices, PSF).  Diagnostic; the synchronising pass is slower than the stage.

    python tools/sr_rep_breakdown.py [--reps 8]
"""
import argparse
import json
import sys
import time
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
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--probe", action="store_true", help="time every PSF forward call through each forward kernel alone")
    a = ap.parse_args()
    dev = "cuda:0"
    rng.set_mode("device")
    shape = (a.size,) * 3
    seg = torch.from_numpy(make_segmentation(shape)[0].astype(np.float32)).to(dev)
    img = (0.1 * seg + 0.05 * torch.rand(shape, device=dev)) * (seg > 0)
    img = img / img.max()
    st = default_artifacts(prob=1.0)["simulate_motion"]
    calls = []
    orig_f, orig_a = K.slice_acq_forward, K.slice_acq_adjoint
    record = [False]

    def wrap(name, fn):
        def inner(tr, *args, **kw):
            if not record[0]:
                return fn(tr, *args, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(tr, *args, **kw)
            torch.cuda.synchronize()
            psf = args[3] if name == "fwd" else args[0]
            calls.append((name, (time.perf_counter() - t0) * 1e3, int(tr.shape[0]), tuple(psf.shape), int((psf > 0).sum())))
            if name == "fwd" and a.probe and psf.numel() > 1:
                # the same call through each forward kernel alone, with the orientation figures the per-slice choice uses
                from fetalsyngen_amd import _lib as _L
                row = {"slices": int(tr.shape[0]), "psf": list(psf.shape), "ss": list(args[4]), "res": round(float(args[5]), 3)}
                for tag, flag in (("direct", 131072), ("plate", 262144), ("default", 0)):
                    _L.load().fsg_set_tuning(flag)
                    fn(tr, *args, **kw)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    fn(tr, *args, **kw)
                    torch.cuda.synchronize()
                    row[tag] = round((time.perf_counter() - t1) * 1e3, 2)
                _L.load().fsg_set_tuning(0)
                T = tr.reshape(-1, 12).float().cpu()
                cost = 10 * T[:, 4].abs() + 35 * T[:, 8].abs()
                row.update(x_axis=[round(float(T[:, i].abs().mean()), 2) for i in (0, 4, 8)],
                           y_axis=[round(float(T[:, i].abs().mean()), 2) for i in (1, 5, 9)],
                           normal=[round(float(T[:, i].abs().mean()), 2) for i in (2, 6, 10)],
                           cost_mean=round(float(cost.mean()), 1), cost_min=round(float(cost.min()), 1), cost_max=round(float(cost.max()), 1))
                print("  probe", json.dumps(row), flush=True)
            return r
        return inner

    K.slice_acq_forward = wrap("fwd", orig_f)
    K.slice_acq_adjoint = wrap("adj", orig_a)
    # svort/slice_acq.py reaches the kernels through `K.<name>` at call time, so patching the module attribute covers the stage
    for rep in range(a.reps + 1):
        row = {}
        for mode in ("wall", "sync"):
            np.random.seed(100 + rep)
            torch.manual_seed(100 + rep)
            record[0] = mode == "sync"
            calls.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _y, meta = st(img, seg, dev, {}, resolution=[0.5, 0.5, 0.5])
            torch.cuda.synchronize()
            row[mode + "_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        fw = [c for c in calls if c[0] == "fwd"]
        ad = [c for c in calls if c[0] == "adj"]
        row.update(nstacks=int(meta["nstacks"]), res_slice=round(float(meta["resolution_slice"]), 3),
                   thick=round(float(meta["slice_thickness"]), 2), gap=round(float(meta["gap"]), 2),
                   fwd_ms=round(sum(c[1] for c in fw), 1), fwd_calls=[(c[2], c[3], c[4], round(c[1], 1)) for c in fw],
                   adj_ms=round(sum(c[1] for c in ad), 1), adj_calls=[(c[2], c[3], c[4], round(c[1], 1)) for c in ad])
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
