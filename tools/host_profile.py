"""Where the host's per-sample time goes: cProfile over N samples at a small volume (the GPU work is then far shorter
than the host's, so wall per sample == host enqueue cost).  Prints the wall per sample without the profiler first.

    python tools/host_profile.py [--size 64] [--n 400] [--top 45]
"""
from __future__ import annotations

import argparse
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from bench import build_generator  # noqa: E402
from fetalsyngen_amd import sharding  # noqa: E402
from fetalsyngen_amd.data.datasets import SeedBank  # noqa: E402
from fetalsyngen_amd.phantom import make_seed_volumes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--n", type=int, default=400)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--sort", default="tottime")
    a = ap.parse_args()
    device = "cuda:0"
    shape = (a.size,) * 3
    seg, seeds = make_seed_volumes(shape)
    bank, segd = SeedBank(seeds, device), torch.from_numpy(seg).to(device)
    gen = build_generator(shape, device, "device")
    gen.prewarm()

    def run(n, base):
        for i in range(n):
            sharding.seed_for_sample(1234, base + i)
            gen._pipeline(None, segd, bank, {}, scale01=True)

    run(50, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.n, 50)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host enqueue: {(t1 - t0) / a.n * 1e6:.1f} us/sample (wall incl. drain {(time.perf_counter() - t0) / a.n * 1e6:.1f})")
    pr = cProfile.Profile()
    pr.enable()
    run(a.n, 1000)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats(a.sort).print_stats(a.top)


if __name__ == "__main__":
    main()
