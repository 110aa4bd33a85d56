#!/usr/bin/env python3
"""Host-side phases of BlurCortex.__call__ at 384^3 (diagnostic): where do the 90 ms of some repetitions go?"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fetalsyngen_amd import kernels as K  # noqa: E402
from fetalsyngen_amd import rng  # noqa: E402
from fetalsyngen_amd.generator.artifacts.utils import mog_params  # noqa: E402
from fetalsyngen_amd.phantom import make_segmentation  # noqa: E402

dev = "cuda:0"
rng.set_mode("device")
shape = (384,) * 3
seg = torch.from_numpy(make_segmentation(shape)[0].astype(np.float32)).to(dev)
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
label = 2.0
x, y, z = shape
for rep in range(16):
    if rep == 8:
        torch.set_num_threads(8)
        print("-- torch.set_num_threads(8)")
    t = [time.perf_counter()]
    prob = K.mog3d(shape, *mog_params([(0, y, z // 2), (x, y, z // 2)], [x // 5, y // 5]), dev)
    pd = K.compact_values(prob, seg, "==", label)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    p = pd.cpu(); t.append(time.perf_counter())
    s = p.sum(); t.append(time.perf_counter())
    p = p / s; t.append(time.perf_counter())
    cdf = torch.cumsum(p.double(), 0); t.append(time.perf_counter())
    u = torch.rand(40, dtype=torch.float64) * float(cdf[-1]); t.append(time.perf_counter())
    v = torch.searchsorted(cdf, u, right=True); t.append(time.perf_counter())
    names = ["gpu", "cpu()", "sum", "div", "cumsum", "rand", "searchsorted"]
    print(rep, p.numel(), {n: round((b - a) * 1e3, 2) for n, a, b in zip(names, t[:-1], t[1:])}, flush=True)
