#!/usr/bin/env python3
"""One randomly oriented stack of bench.py's config4_sr problem through the reconstruction scatter, twice (for counter passes:
tools/sa_adjoint_pmc.sh)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fetalsyngen_amd import kernels as K  # noqa: E402
from fetalsyngen_amd.generator.artifacts.svort import get_PSF, random_init_stack_transforms  # noqa: E402

dev, size, res, res_slice, thick, nsl = "cuda:0", 384, 0.5, 0.8, 3.0, 80
shape = (size,) * 3
ss = int(np.ceil(int(np.sqrt(3 * size ** 2 / 2.0) * res / res_slice) / 32.0) * 32)
psf = get_PSF(res_ratio=(res_slice / res, res_slice / res, thick / res)).to(dev)
np.random.seed(0)
tr = random_init_stack_transforms(nsl, size * res / nsl / res, False, 0).matrix().to(dev)
vol = torch.rand(shape, device=dev)
rs_ = res_slice / res
sl = K.slice_acq_forward(tr, vol, None, None, psf, (ss, ss), rs_)
for _ in range(2):
    K.slice_acq_adjoint(tr, psf, sl, None, None, shape, rs_, interp_psf=True, equalize=True)
torch.cuda.synchronize()
print("pixel_taps", nsl * ss * ss * int((psf > 0).sum()), "wave_taps", nsl * ss * ss * int((psf > 0).sum()) // 64)
