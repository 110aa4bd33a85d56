#!/usr/bin/env python3
"""One launch each of the direct-atomics and the LDS pre-summing slice-acquisition adjoint (and one forward) on the
tools/sr_bench.py workload, for `rocprofv3 --pmc WRITE_SIZE` / `--pmc FETCH_SIZE` (separate passes, counters only):

    cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python3 $REPO/tools/sr_pmc.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fetalsyngen_amd import _lib, kernels as K  # noqa: E402
from fetalsyngen_amd.generator.artifacts.svort import get_PSF, random_stack  # noqa: E402

dev, res, size, n, res_s, thick = "cuda:0", 0.5, 384, 80, 0.8, 3.0
vs = (size,) * 3
ss = int(np.ceil(int(np.sqrt(3 * size**2 / 2.0) * res / res_s) / 32.0) * 32)
psf = get_PSF(res_ratio=(res_s / res, res_s / res, thick / res)).to(dev)
np.random.seed(1)
tr = random_stack(n, gap=size * res / n / res, max_angle=0.8).to(dev)
vol = torch.rand(vs, device=dev)
rs = res_s / res
s = K.slice_acq_forward(tr, vol, None, None, psf, (ss, ss), rs)
lib = _lib.load()
prev = lib.fsg_set_tuning(128)
K.slice_acq_adjoint(tr, psf, s, None, None, vs, rs, interp_psf=True, equalize=True)
lib.fsg_set_tuning(prev)
K.slice_acq_adjoint(tr, psf, s, None, None, vs, rs, interp_psf=True, equalize=True)
torch.cuda.synchronize()
print("pixel_taps", n * ss * ss * int((psf > 0).sum()))
