"""Host-only cost of one step: enqueue time per sample (no sync inside the loop) vs end-to-end time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from fetalsyngen_amd.data.datasets import SeedBank
from fetalsyngen_amd.phantom import make_seed_volumes
from fetalsyngen_amd import sharding

import os
shape = (int(os.environ.get("FSG_HT_SIZE", "256")),) * 3
seg, seeds = make_seed_volumes(shape)
bank = SeedBank(seeds, "cuda:0"); segd = torch.from_numpy(seg).to("cuda:0")
gen = bench.build_generator(shape, "cuda:0", "device"); gen.prewarm()
def run(n, sync_each):
    t0 = time.perf_counter()
    for i in range(n):
        sharding.seed_for_sample(1, i)
        gen._pipeline(None, segd, bank, {}, scale01=True)
        if sync_each: torch.cuda.synchronize()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
run(10, False)
print("enqueue us/step, total us/step (async):", run(100, False))
print("sync each step:", run(50, True))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable(); run(100, False); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
