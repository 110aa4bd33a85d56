#!/usr/bin/env python3
"""Time the slice-acquisition kernels at BASELINE config 4 scale (384^3 volume at 0.5 mm, one stack of a
3 mm-thick, 0.8 mm in-plane acquisition) with HIP events on the launch stream.  Prints one JSON line.

    python tools/sr_bench.py [--size 384] [--slices 80] [--reps 5]
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fetalsyngen_amd import kernels as K  # noqa: E402
from fetalsyngen_amd.generator.artifacts.svort import get_PSF, random_stack  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--slices", type=int, default=80)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--res-slice", type=float, default=0.8)
    ap.add_argument("--thick", type=float, default=3.0)
    a = ap.parse_args()
    dev, res = "cuda:0", 0.5
    vs = (a.size,) * 3
    ss = int(np.sqrt(3 * a.size**2 / 2.0) * res / a.res_slice)
    ss = int(np.ceil(ss / 32.0) * 32)
    psf = get_PSF(res_ratio=(a.res_slice / res, a.res_slice / res, a.thick / res)).to(dev)
    delta = get_PSF(0).to(dev)
    np.random.seed(0)
    tr = random_stack(a.slices, gap=a.size * res / a.slices / res, max_angle=0.3).to(dev)
    vol = torch.rand(vs, device=dev)
    out = {"volume": vs, "slices": [a.slices, ss, ss], "psf": list(psf.shape), "psf_taps": int((psf > 0).sum())}
    rs = a.res_slice / res
    s = K.slice_acq_forward(tr, vol, None, None, psf, (ss, ss), rs)
    ntap = out["psf_taps"]
    npix = a.slices * ss * ss
    for name, fn in {
        "forward_linear_ms": lambda: K.slice_acq_forward(tr, vol, None, None, psf, (ss, ss), rs),
        "forward_delta_ms": lambda: K.slice_acq_forward(tr, vol, None, None, delta, (ss, ss), rs),
        "adjoint_nearest_psf_eq_ms": lambda: K.slice_acq_adjoint(tr, psf, s, None, None, vs, rs, interp_psf=True, equalize=True),
        "adjoint_linear_ms": lambda: K.slice_acq_adjoint(tr, psf, s, None, None, vs, rs, interp_psf=False),
        "forward_nearest_psf_ms": lambda: K.slice_acq_forward(tr, vol, None, None, psf, (ss, ss), rs, interp_psf=True),
        "forward_torch_ms": lambda: K.slice_acq_forward(tr, vol, None, None, psf, (ss, ss), rs, semantics="torch"),
    }.items():
        out[name] = round(timed(fn, a.reps), 3)
    # adjoint (interp_psf + equalize): direct atomics vs LDS pre-summation under a few shapes, and vs stack tilt
    from fetalsyngen_amd import _lib
    lib = _lib.load()
    adj = lambda t=tr: K.slice_acq_adjoint(t, psf, s, None, None, vs, rs, interp_psf=True, equalize=True)  # noqa: E731
    prev = lib.fsg_set_tuning(128)
    out["adjoint_direct_ms"] = round(timed(adj, a.reps), 3)
    lib.fsg_set_tuning(prev)
    sweep = {}
    cfgs = ((4096, 0, 20), (3072, 0, 20), (6144, 0, 20), (4096, 3, 20), (4096, 2, 20), (6144, 1, 20), (4096, 0, 0), (4096, 0, 40))
    for ang in (0.3, 0.8, 1.5):
        np.random.seed(1)
        t2 = random_stack(a.slices, gap=a.size * res / a.slices / res, max_angle=ang).to(dev)
        row = {}
        prev = lib.fsg_set_tuning(128)
        row["direct"] = round(timed(lambda: adj(t2), a.reps), 2)
        lib.fsg_set_tuning(prev)
        for cap, zc, t16 in cfgs:
            lib.fsg_slice_acq_set_tuning(cap, zc, t16)
            row[f"{cap}/{zc}/{t16}"] = round(timed(lambda: adj(t2), a.reps), 2)
        sweep[f"tilt{ang}"] = row
    lib.fsg_slice_acq_set_tuning(3072, 0, 20)
    out["adjoint_sweep_ms"] = sweep
    out["pixel_taps"] = npix * ntap
    out["forward_linear_Gtaps_per_s"] = round(npix * ntap / out["forward_linear_ms"] / 1e6, 2)
    out["adjoint_Gtaps_per_s"] = round(npix * ntap / out["adjoint_nearest_psf_eq_ms"] / 1e6, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
