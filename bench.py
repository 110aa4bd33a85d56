#!/usr/bin/env python3
"""Benchmark of the per-volume synthesis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full pass of the path (GMM draw -> deformation min/max -> fused warp(+gamma+bias,
labels) -> 3-pass Gaussian blur -> resample+noise -> zoom-back+[0,1] normalise) over one synthetic
256^3 label volume that is already resident in HBM (uint8 seed labels + fp32 segmentation), every
stage gate on, device-Philox RNG, outputs left in HBM.  N > 1: one process per GPU (launched by
torch.distributed.run), every rank processes its own volumes (independent work, no collective in the
data path; gloo is used for the barrier and the max-over-ranks time only).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable copy


def build_generator(shape, device, rng_mode):
    from fetalsyngen_amd.generator.augmentation.synthseg import RandBiasField, RandGamma, RandNoise, RandResample
    from fetalsyngen_amd.generator.deformation.affine_nonrigid import SpatialDeformation
    from fetalsyngen_amd.generator.intensity.rand_gmm import ImageFromSeeds
    from fetalsyngen_amd.generator.model import FetalSynthGen

    labels = [0] + list(range(10, 50))
    classes = [0] + [10] * 10 + [20] * 10 + [30] * 10 + list(range(40, 50))
    p = 1.0  # every gate on, so every kernel of the path runs in every step
    return FetalSynthGen(
        shape=list(shape), resolution=[0.5, 0.5, 0.5], device=device,
        intensity_generator=ImageFromSeeds(1, 6, labels, classes),
        spatial_deform=SpatialDeformation(20, 0.02, 0.1, list(shape), p, True, 0.03, 0.06, 4, 0.5, device),
        resampler=RandResample(p, 0.5, 1.5), bias_field=RandBiasField(p, 0.004, 0.02, 0.01, 0.3),
        noise=RandNoise(p, 5, 15), gamma=RandGamma(p, 0.1), rng=rng_mode)


def _drop_events(events):
    from fetalsyngen_amd import _lib

    lib = _lib.load()
    for e0, e1, _ in events:
        lib.fsg_event_destroy(e0)
        lib.fsg_event_destroy(e1)
    events.clear()


def blur_traffic_per_sample(passes, size):
    """HBM-side bytes of one sample's blur (x pass + fused y,z pass) from the committed PMC summary (profiles/r*_blur_pmc.json: FETCH_SIZE
    and WRITE_SIZE collected in separate rocprofv3 --pmc passes of the same kernels, gfx950 x2 fetch
    correction applied), averaged over the radius mix this run actually launched."""
    files = sorted((REPO / "profiles").glob("r*_blur_pmc.json"))
    if not files or not passes:
        return None
    table = json.loads(files[-1].read_text())["sizes"].get(str(size))
    if not table:
        return None

    def lookup(prefix, R):
        pts = {}
        for k, v in table.items():
            if k.startswith(prefix):
                pts[int(k[len(prefix):].split(",")[0].rstrip(">"))] = v["traffic_bytes"]
        if not pts:
            return None
        rs = sorted(pts)
        if R <= rs[0]:
            return pts[rs[0]]
        if R >= rs[-1]:
            return pts[rs[-1]]
        for lo_, hi_ in zip(rs, rs[1:]):
            if lo_ <= R <= hi_:
                return pts[lo_] + (pts[hi_] - pts[lo_]) * (R - lo_) / (hi_ - lo_)

    # per blur section: x pass, then the fused y+z launch when both are active with the same radius (<= 8), else two passes
    tot, nsec = 0.0, 0
    i = 0
    while i < len(passes):
        sec = []
        while i < len(passes) and (not sec or passes[i][0] > sec[-1][0]):
            sec.append(passes[i])
            i += 1
        byaxis = dict(sec)
        if 0 in byaxis:
            t = lookup("blur_strided_v4<", byaxis[0])
            if t is None:
                return None
            tot += t
        if 1 in byaxis and 2 in byaxis and byaxis[1] == byaxis[2] and byaxis[1] <= 8:
            t = lookup("blur_yz_fused_kernel<", byaxis[1])
            if t is None:
                return None
            tot += t
        else:
            for axis in (1, 2):
                if axis in byaxis:
                    t = lookup("blur_contig_lds<" if axis == 2 else "blur_strided_v4<", byaxis[axis])
                    if t is None:
                        return None
                    tot += t
        nsec += 1
    return round(tot / max(nsec, 1))


def blur_microbench(shape, device, sigma=1.3, reps=20):
    """Back-to-back launches of each axis pass between HIP events; algorithmic bytes = 8 B/voxel/pass."""
    from fetalsyngen_amd import kernels as K
    from fetalsyngen_amd import tables as T

    taps = T.gaussian_taps(sigma)
    nvox = int(np.prod(shape))
    out = {}
    # L3-warm: same 64 MiB buffer re-read; L3-cold: cycle through > 256 MiB of distinct buffers
    bufs = [torch.rand(shape, device=device) * 255 for _ in range(6)]
    for label, pool in (("l3_warm", bufs[:1]), ("l3_cold", bufs)):
        res = {}
        for axis in range(3):
            for w in range(3):
                K.blur_axis(pool[w % len(pool)], axis, taps)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for r in range(reps):
                K.blur_axis(pool[r % len(pool)], axis, taps)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            res[f"axis{axis}"] = {"us": round(us, 2), "GBps": round(8.0 * nvox / us / 1e3, 1)}
        # the y and z passes as one launch: two passes' algorithmic bytes (16 B/voxel)
        if K.blur_yz(pool[0], taps, taps) is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for r in range(reps):
                K.blur_yz(pool[r % len(pool)], taps, taps)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            res["fused_yz"] = {"us": round(us, 2), "GBps": round(16.0 * nvox / us / 1e3, 1)}
        out[label] = res
    return out


def config4_sr(device, size=384, reps=3):
    """BASELINE configs[3] side line (not the metric): the SR-artifact slice-stack simulation (SimulateMotion = Scanner.scan +
    PSFReconstructor.recon_psf, default YAML ranges, device RNG) on one 384^3 / 0.5 mm volume, wall ms per volume."""
    from fetalsyngen_amd.generator.defaults import default_artifacts
    from fetalsyngen_amd.phantom import make_segmentation

    shape = (size,) * 3
    seg = torch.from_numpy(make_segmentation(shape)[0].astype(np.float32)).to(device)
    img = (0.1 * seg + 0.05 * torch.rand(shape, device=device)) * (seg > 0)
    img = img / img.max()
    stage = default_artifacts(prob=1.0)["simulate_motion"]
    ms, stacks = [], []
    for rep in range(reps + 1):
        np.random.seed(100 + rep)
        torch.manual_seed(100 + rep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _y, meta = stage(img, seg, device, {}, resolution=[0.5, 0.5, 0.5])
        torch.cuda.synchronize()
        ms.append(round((time.perf_counter() - t0) * 1e3, 1))
        stacks.append(int(meta["nstacks"]))
    return {"workload": f"SimulateMotion on one {size}^3 volume (2-6 stacks of slices, PSF acquisition + reconstruction)",
            "ms_per_volume": ms[1:], "nstacks": stacks[1:], "mean_ms": round(float(np.mean(ms[1:])), 1)}


def cpu_baseline(shape, threads):
    """The CPU restatement of the reference path (oracle/, validated against the real reference by the
    golden vectors) timed on this host: one warm-up + four timed full-size samples."""
    from fetalsyngen_amd.phantom import make_seed_volumes
    from oracle import fsg_oracle as O

    torch.set_num_threads(threads)
    seg, seeds = make_seed_volumes(shape)
    cfg = O.Config(shape, prob=1.0)
    seg_t = torch.from_numpy(seg)
    times = []
    for rep in range(5):  # 1 warm-up + 4 timed volumes (different draws): ~10-15 s of CPU work
        np.random.seed(rep)
        torch.manual_seed(rep)
        t0 = time.perf_counter()
        O.run_sample(cfg, seg_t, seeds)
        times.append(time.perf_counter() - t0)
    timed = times[1:]
    return {"value": round(len(timed) / sum(timed), 4), "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"1 warm-up + {len(timed)} timed {shape[0]}^3 volumes, all gates on, torch CPU ops "
                      f"({sum(timed):.2f} s, {min(timed):.2f}-{max(timed):.2f} s per volume)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--rng", default="device", choices=["device", "reference"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--no-sr", action="store_true", help="skip the 384^3 SimulateMotion side line")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the samples are spread over (round robin)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # one process per GPU: keep each rank's host-side plans (small numpy / torch CPU ops) on its share of the cores
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))

    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    _lib.load()  # fail loudly if the HIP library is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to benchmark")
    if os.environ.get("FSG_BENCH_SHARE_GPU0"):  # rehearsal of the N>1 path on a 1-GPU box: every rank on cuda:0
        local = 0
    device = f"cuda:{local}"
    torch.cuda.set_device(device)
    shape = (args.size,) * 3
    nvox = int(np.prod(shape))

    # inputs resident in HBM before the timed region: 4 distinct label volumes per rank
    banks, segs = [], []
    for v in range(4):
        seg, seeds = make_seed_volumes(shape, variant=rank * 4 + v)
        banks.append(SeedBank(seeds, device))
        segs.append(torch.from_numpy(seg).to(device))
    gen = build_generator(shape, device, args.rng)
    gen.prewarm()  # static per-axis tables of this configuration -> device (outside the timed region)

    blur_ms = []

    streams = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else [None]

    def step(i, timed):
        sharding.seed_for_sample(1234, rank + world * i)
        k = i % 4
        st = streams[i % len(streams)]
        if st is None:
            out, seg_d, _img, _p = gen._pipeline(None, segs[k], banks[k], {}, scale01=True)
        else:
            with torch.cuda.stream(st):
                out, seg_d, _img, _p = gen._pipeline(None, segs[k], banks[k], {}, scale01=True)
        return out, seg_d

    # per-step blur timing: HIP events recorded on the launch stream around the three axis passes of every
    # sample, inside fsg_sample_run (FetalSynthGen.blur_events)
    gen.blur_events = []

    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize()
    _drop_events(gen.blur_events)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lib = _lib.load()
    sections, blur_total_ms, passes = 0, 0.0, []
    ms = ctypes.c_float()
    for e0, e1, pl in gen.blur_events:
        _lib.check(lib.fsg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "fsg_event_elapsed_ms")
        blur_total_ms += ms.value
        sections += 1
        passes.extend(pl)
    _drop_events(gen.blur_events)
    gen.blur_events = None
    # one blur = three axis passes (8 B/voxel/pass, 24 B/voxel for the blur) issued as two launches: the x pass
    # and the fused y+z pass (the intermediate stays in LDS)
    npass = len(passes)
    blur_us = blur_total_ms * 1e3 / max(sections, 1)           # per sample
    alg_bytes = 8.0 * nvox * npass / max(sections, 1)          # per sample
    achieved = alg_bytes / blur_us / 1e3 if sections else 0.0  # GB/s

    traffic_sample = blur_traffic_per_sample(passes, args.size)  # HBM bytes of the two launches (committed PMC table)
    result = {
        "metric": "synthetic volumes/sec at 256^3 (full deform+GMM+blur+resample path)",
        "value": round(world * args.steps / dt, 3),
        "unit": "volumes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: single {args.size}^3 label volume per step, full path, all gates on",
                   "rng": args.rng, "inputs": "uint8 seed labels + fp32 segmentation resident in HBM",
                   "outputs": "fp32 [0,1] image + fp32 labels in HBM", "volumes_per_rank": args.steps,
                   "parallelism": f"{world} independent replicas (no collective)", "streams_per_gpu": args.streams},
        "roofline": {"bound": "hbm",
                     "kernel": "separable 3-pass blur = x pass (fsg_blur_axis_taps_host_f32) + fused y,z pass "
                               "(fsg_blur_yz_taps_host_f32), HIP events around both launches of every timed sample",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": (None if traffic_sample is None else int(traffic_sample / 2)),  # per launch, like achieved
                     "us_per_launch": round(blur_us / 2, 2), "launches_timed": 2 * sections,
                     "algorithmic_bytes_per_launch": int(alg_bytes / 2),
                     "per_sample": {"us": round(blur_us, 2), "axis_passes": round(npass / max(sections, 1), 2), "launches": 2,
                                    "algorithmic_bytes": int(alg_bytes), "traffic": traffic_sample}},
    }
    if rank == 0:
        if not args.no_microbench:
            result["blur_microbench"] = blur_microbench(shape, device)
        if world == 1 and not args.no_sr:
            result["config4_sr"] = config4_sr(device)
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            result["cpu_baseline"] = cpu_baseline(shape, threads)
            result["gpu_over_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
