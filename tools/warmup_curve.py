"""Host enqueue time of the first steps of the bench loop (what a short --warmup/--steps run sees)."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from bench import build_generator
from fetalsyngen_amd import sharding
from fetalsyngen_amd.data.datasets import SeedBank
from fetalsyngen_amd.phantom import make_seed_volumes
dev = "cuda:0"; shape = (256,) * 3
segs, banks = [], []
for v in range(4):
    seg, seeds = make_seed_volumes(shape, v); segs.append(torch.from_numpy(seg).to(dev)); banks.append(SeedBank(seeds, dev))
gen = build_generator(shape, dev, "device"); gen.prewarm(); gen.reserve(shape, samples_in_flight=12)
torch.cuda.synchronize()
ts = []
t_all = time.perf_counter()
for i in range(60):
    t0 = time.perf_counter()
    sharding.seed_for_sample(1234, i)
    out = gen._pipeline(None, segs[i % 4], banks[i % 4], {}, scale01=True)
    ts.append((time.perf_counter() - t0) * 1e6)
    if i in (4, 24):
        torch.cuda.synchronize(); print("sync at", i, "elapsed ms", round((time.perf_counter() - t_all) * 1e3, 2))
torch.cuda.synchronize()
print("host us per step:", [round(t) for t in ts])
print("total ms", round((time.perf_counter() - t_all) * 1e3, 2))
