"""K7 (resample + Philox noise) time by low-res size m, default dispatch.  python tools/k7_sweep.py [lo hi step]"""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from fetalsyngen_amd import kernels as K, tables as T
lo, hi, st = (int(v) for v in (sys.argv[1:4] + ["150", "256", "1"])[:3])
dev = "cuda:0"; shape = (256,) * 3
img = torch.rand(shape, device=dev) * 255
res = {}
for m in range(lo, hi + 1, st):
    stds, new, fac, rtabs = T.resample_plan(shape, [0.5] * 3, [0.5 * 256 / m] * 3, 0.5)
    rt = K.DeviceTables(rtabs, dev)
    for _ in range(3): K.resample_noise(img, rt, noise_std=9.0, seed=3, stream_id=2)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.resample_noise(img, rt, noise_std=9.0, seed=3, stream_id=2)
    e1.record(); e1.synchronize()
    res[new[0]] = round(e0.elapsed_time(e1) * 100, 1)
print(json.dumps(res))
