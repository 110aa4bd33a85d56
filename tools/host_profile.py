"""Host-side time breakdown of one sample (cProfile) -- where Python spends the step."""
import cProfile, pstats, sys, io
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import bench
from fetalsyngen_amd.data.datasets import SeedBank
from fetalsyngen_amd.phantom import make_seed_volumes
from fetalsyngen_amd import sharding

shape = (256,) * 3
seg, seeds = make_seed_volumes(shape)
bank = SeedBank(seeds, "cuda:0"); segd = torch.from_numpy(seg).to("cuda:0")
gen = bench.build_generator(shape, "cuda:0", "device")
def run(n):
    for i in range(n):
        sharding.seed_for_sample(1, i)
        gen._pipeline(None, segd, bank, {}, scale01=True)
    torch.cuda.synchronize()
run(5)
pr = cProfile.Profile(); pr.enable(); run(50); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
