#!/usr/bin/env python3
"""Benchmark of the per-volume synthesis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full pass of the path (GMM draw -> deformation min/max -> fused warp(+gamma+bias,
labels) -> 3-pass Gaussian blur -> resample+noise -> zoom-back+[0,1] normalise) over one synthetic
256^3 label volume that is already resident in HBM (uint8 seed labels + fp32 segmentation), every
stage gate on, device-Philox RNG, outputs left in HBM.

N > 1: one process per GPU.  Either the caller creates the ranks (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- when
WORLD_SIZE is NOT set -- this script starts N child ranks itself before anything touches a GPU, waits for
them and relays rank 0's line (the parent never initialises HIP).  Every rank processes its own volumes
(independent work, no collective in the data path; a gloo group carries the barrier, the max-over-ranks
time and the per-rank reports only).

Prints ONE JSON line (rank 0).  Beside the headline (`value`, BASELINE configs[1]) the line carries
`roofline` (blur kernel, live HIP events), `roofline_step`, `config3` (32 x 256^3 volumes dealt i % N,
strong scaling), `config5` (streaming epoch, CPU-tensor contract and device-resident), `ranks`, and at
N = 1 `blur_microbench`, `config4_sr`, `cpu_baseline`.

`--dry-plan`: no GPU work at all -- every rank only draws the host plans of its samples (what the Python
side does per sample); used by the CPU test of the N-rank launcher and as a host-cost probe.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable copy
# the real reference timed in the build container: the anchor the CPU port is read against
REFERENCE_ANCHOR = {"value": 0.156, "unit": "volumes/s", "threads": 8, "s_per_volume": 6.42,
                    "what": "real reference (imported, CPU, torch 2.10, all gates on, sub-sta21 256^3) in the build "
                            "container, BASELINE.md section 2.  On that machine (8 threads) the port takes 4.7-6.8 s per volume, "
                            "i.e. the reference's cost within ~25 %; the figure below is the port on THIS box's host cores"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--rng", default="keyed", choices=["keyed", "device", "reference"],
                    help="keyed (default): every draw of a sample from Philox under its (base_seed, index) key, in C / on the "
                         "device; device: the reference's host draw order, large fields from device Philox; reference: host tape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--no-sr", action="store_true", help="skip the 384^3 SimulateMotion side line")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--batch-volumes", type=int, default=32, help="config3: volumes in the batch dealt i %% N")
    ap.add_argument("--stream-volumes", type=int, default=2000, help="config5: volumes streamed per rank and mode")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the samples are spread over (round robin)")
    ap.add_argument("--tune", type=int, default=0, help="fsg_set_tuning flags (A/B runs, e.g. 16384 = unfused blur + K7)")
    ap.add_argument("--no-look-ahead", dest="look_ahead", action="store_false",
                    help="keyed mode: do not name the next sample (its draw job then runs as a launch of its own)")
    ap.add_argument("--dry-plan", action="store_true", help="host plans only, no GPU (launcher test / host-cost probe)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# N-rank launcher (parent process: never touches a GPU)
# ----------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Start `--gpus` child ranks of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set per child), wait, check
    that N distinct devices reported, relay rank 0's JSON line.  Exit code != 0 if any child failed."""
    import tempfile

    n = args.gpus
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile("w+") as out0_file:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), FSG_BENCH_CHILD="1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                          stdout=out0_file if r == 0 else subprocess.DEVNULL, stderr=None))
        # a rank that dies would leave the others waiting in a barrier: stop them (exact PIDs) as soon as one fails
        codes = [None] * n
        while any(c is None for c in codes):
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    codes[r] = pr.poll()
            if any(c not in (None, 0) for c in codes):
                for r, pr in enumerate(procs):
                    if codes[r] is None:
                        pr.terminate()
                for r, pr in enumerate(procs):
                    if codes[r] is None:
                        try:
                            codes[r] = pr.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            pr.kill()
                            codes[r] = pr.wait()
                break
            time.sleep(0.1)
        out0_file.seek(0)
        out0 = out0_file.read()
    if any(codes):
        sys.stderr.write(f"bench.py launcher: rank exit codes {codes}\n")
        sys.stdout.write(out0 or "")
        return 1
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None:
        sys.stderr.write("bench.py launcher: rank 0 printed no JSON line\n")
        return 1
    res = json.loads(line)
    ranks = res.get("ranks", [])
    shared = bool(os.environ.get("FSG_BENCH_SHARE_GPU0")) or args.dry_plan
    devices = {(r.get("device"), r.get("uuid")) for r in ranks}
    if res.get("n_gpus") != n or sorted(r.get("rank") for r in ranks) != list(range(n)) or (
            not shared and len(devices) != n):
        sys.stderr.write(f"bench.py launcher: expected {n} ranks on {n} distinct devices, got n_gpus={res.get('n_gpus')} "
                         f"ranks={ranks}\n")
        return 1
    print(line, flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------
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
    for e0, e1, *_ in events:
        lib.fsg_event_destroy(e0)
        lib.fsg_event_destroy(e1)
    events.clear()


def _blur_launch_plan(sec):
    """[(kernel prefix of the PMC table, radius)] for one sample's blur section `[(axis, radius), ...]`: x pass, then
    the fused y+z launch when both are active with the same radius (<= 8), else one launch per axis."""
    byaxis = dict(sec)
    out = []
    if 0 in byaxis:
        out.append(("blur_strided_v4<", byaxis[0]))
    if 1 in byaxis and 2 in byaxis and byaxis[1] == byaxis[2] and byaxis[1] <= 8:
        out.append(("blur_yz_fused_kernel<", byaxis[1]))
    else:
        for axis in (1, 2):
            if axis in byaxis:
                out.append(("blur_contig_lds<" if axis == 2 else "blur_strided_v4<", byaxis[axis]))
    return out


def blur_traffic(sections, size):
    """HBM-side bytes of the timed blur launches from the committed PMC summary (profiles/r*_blur_pmc.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc passes of the same kernels, gfx950 x2 fetch correction applied),
    averaged over the radius mix this run actually launched.  -> (bytes per launch, launches) or (None, launches)."""
    launches = [l for sec in sections for l in _blur_launch_plan(sec)]
    files = sorted((REPO / "profiles").glob("r*_blur_pmc.json"))
    if not files or not launches:
        return None, len(launches)
    table = json.loads(files[-1].read_text())["sizes"].get(str(size))
    if not table:
        return None, len(launches)

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

    tot = 0.0
    for prefix, R in launches:
        t = lookup(prefix, R)
        if t is None:
            return None, len(launches)
        tot += t
    return tot / len(launches), len(launches)


def stage_rooflines(traces, nvox):
    """Per-launch figures from the stage traces (HIP events behind every launch of a sample, fsg_sample_plan::trace_events):
    mean microseconds, the algorithmic bytes of that launch (SURVEY 8(d) per-unit figures x the voxels it processes: N =
    full grid, M = low-res grid of THAT sample) and the fraction of the HBM peak.  Each interval holds one barrier packet
    besides the kernel, so `us` is an upper bound of the kernel's own duration (rocprof's per-kernel averages are in
    profiles/)."""
    import numpy as np

    def alg_bytes(stage, meta):
        N = float(np.prod(meta["shape"]))
        M = float(np.prod(meta["low_shape"])) if meta.get("low_shape") else N
        m0 = meta["low_shape"][0] if meta.get("low_shape") else meta["shape"][0]
        return {
            "upload": 65536.0, "draw": 65536.0,
            # seed labels read (the subject's uint16 code volume, or 4 x uint8 volumes) + float32 image written (K1; rows / face minima ~0)
            "head": (4.0 + meta.get("label_bytes", 4)) * N,
            "gmm": 8.0 * N,
            "floormin": 0.0,               # reads the coarse grid only
            "rows": 0.0,
            "warp": 13.0 * N,              # image 4 + 4, uint8 label twin 1, float32 labels 4 (K4a + K4b, K5 fused: 0)
            "blur_x": 8.0 * N, "blur_y": 8.0 * N, "blur_z": 8.0 * N,
            "blur_yz": 8.0 * N,            # two axis passes for one read + one write of the volume
            "k7": 4.0 * N + 4.0 * M,
            "k9a": 4.0 * M,                # min / max of the zoom-back: reads the low-res volume, writes nothing
            "k9b": 4.0 * M + 4.0 * N,
            "blur_rs_x": 4.0 * N + 4.0 * N * m0 / meta["shape"][0],
            "blur_rs_yz": 4.0 * N * m0 / meta["shape"][0] + 4.0 * M,
            "pointwise": 8.0 * N,
        }.get(stage, 0.0)

    acc = {}
    for tr in traces:
        for stage, us in tr.elapsed_us():
            a = acc.setdefault(stage, {"us": [], "bytes": []})
            a["us"].append(us)
            a["bytes"].append(alg_bytes(stage, tr.meta or {"shape": (1, 1, 1)}))
    out, total = {}, 0.0
    for stage, a in acc.items():
        us, by = float(np.mean(a["us"])), float(np.mean(a["bytes"]))
        total += us * len(a["us"]) / max(len(traces), 1)
        out[stage] = {"us": round(us, 2), "us_min": round(float(np.min(a["us"])), 2), "us_max": round(float(np.max(a["us"])), 2),
                      "launches_per_sample": round(len(a["us"]) / max(len(traces), 1), 2),
                      "algorithmic_bytes": int(by), "GBps": round(by / us / 1e3, 1) if us > 0 else None,
                      "frac": round(by / us / 1e3 / HBM_PEAK_GBS, 4) if us > 0 else None}
    out["_sum_us_per_sample"] = round(total, 2)
    out["_samples"] = len(traces)
    out["_note"] = ("untimed pass after the timed region, one event behind every launch: each interval = the launch + one "
                    "barrier packet, so the sum exceeds ms_per_step of the un-instrumented run")
    return out


def blur_microbench(shape, device, sigma=1.3, reps=20):
    """Back-to-back launches of each blur kernel between HIP events.  GBps = bytes the launch must move (read the volume
    once + write it once = 8 B/voxel, also for the fused y+z launch, whose intermediate never leaves LDS) / time; the
    fused launch additionally reports the two-pass equivalent (16 B/voxel of per-pass algorithmic bytes)."""
    import numpy as np
    import torch

    from fetalsyngen_amd import kernels as K
    from fetalsyngen_amd import tables as T

    taps = T.gaussian_taps(sigma)
    nvox = int(np.prod(shape))
    out = {}
    # L3-warm: same 64 MiB buffer re-read; L3-cold: cycle through > 256 MiB of distinct buffers
    bufs = [torch.rand(shape, device=device) * 255 for _ in range(6)]
    for label, pool in (("l3_warm", bufs[:1]), ("l3_cold", bufs)):
        res = {}

        def timed(fn):
            for w in range(3):
                fn(pool[w % len(pool)])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for r in range(reps):
                fn(pool[r % len(pool)])
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps

        for axis in range(3):
            us = timed(lambda x, a=axis: K.blur_axis(x, a, taps))
            res[f"axis{axis}"] = {"us": round(us, 2), "GBps": round(8.0 * nvox / us / 1e3, 1)}
        if K.blur_yz(pool[0], taps, taps) is not None:
            us = timed(lambda x: K.blur_yz(x, taps, taps))
            res["fused_yz"] = {"us": round(us, 2), "GBps": round(8.0 * nvox / us / 1e3, 1),
                               "two_pass_equivalent_GBps": round(16.0 * nvox / us / 1e3, 1)}
        out[label] = res
    return out


def config4_sr(device, size=384, reps=3):
    """BASELINE configs[3] side line (not the metric): the SR-artifact slice-stack simulation (SimulateMotion = Scanner.scan +
    PSFReconstructor.recon_psf, default YAML ranges, device RNG) on one 384^3 / 0.5 mm volume, wall ms per volume."""
    import numpy as np
    import torch

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
    out = {"workload": f"SimulateMotion on one {size}^3 volume (2-6 stacks of slices, PSF acquisition + reconstruction)",
           "ms_per_volume": ms[1:], "nstacks": stacks[1:], "mean_ms": round(float(np.mean(ms[1:])), 1)}
    # the two kernels that carry the stage (forward 37 %, adjoint 49 % of its GPU time), one stack of 80 slices at 0.8 mm in
    # plane / 3 mm thick: HIP-event time, pixel-taps/s (their unit of work: one PSF tap of one slice pixel = a 2x2x2 gather +
    # blend, or one scattered contribution) and the bytes they must move (the volume once + the slices once)
    from fetalsyngen_amd import kernels as K
    from fetalsyngen_amd.generator.artifacts.svort import get_PSF, random_init_stack_transforms

    res, res_slice, thick, nsl = 0.5, 0.8, 3.0, 80
    ss = int(np.ceil(int(np.sqrt(3 * size ** 2 / 2.0) * res / res_slice) / 32.0) * 32)
    psf = get_PSF(res_ratio=(res_slice / res, res_slice / res, thick / res)).to(device)
    np.random.seed(0)
    # stack orientations as Scanner.scan draws them (uniformly random rotations, svort transform.py:178-188, :359-369): the
    # figures below are means over four stacks -- the forward kernel's time depends on the orientation (4-39 ms by direct
    # gathers, 11-18 ms from the LDS plate; chosen per slice, profiles/r03_i_slice_acq_forward.txt)
    stacks4 = [random_init_stack_transforms(nsl, size * res / nsl / res, False, 0).matrix().to(device) for _ in range(4)]
    tr = stacks4[0]
    vol = torch.rand(shape, device=device)
    rs_ = res_slice / res
    sl = K.slice_acq_forward(tr, vol, None, None, psf, (ss, ss), rs_)
    ntap, npix = int((psf > 0).sum()), nsl * ss * ss

    def timed(fn, reps_=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps_):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps_

    fwd = float(np.mean([timed(lambda t_=t_: K.slice_acq_forward(t_, vol, None, None, psf, (ss, ss), rs_)) for t_ in stacks4]))
    adj = float(np.mean([timed(lambda t_=t_: K.slice_acq_adjoint(t_, psf, sl, None, None, shape, rs_, interp_psf=True, equalize=True))
                         for t_ in stacks4]))
    byt = 4.0 * size ** 3 + 4.0 * npix
    out["kernels"] = {
        "problem": {"slices": [nsl, ss, ss], "psf": list(psf.shape), "psf_taps": ntap, "pixel_taps": npix * ntap},
        "forward": {"kernel": "sa_forward_linear_fast_kernel / sa_forward_plate_kernel (per slice, by orientation)", "ms": round(fwd, 3), "G_pixel_taps_per_s": round(npix * ntap / fwd / 1e6, 1),
                    "algorithmic_bytes": int(byt), "GBps": round(byt / fwd / 1e6, 1), "bound": "direct gathers: L1/TA lines per wave-gather (grows with the z component of the slice's x axis); plate: instruction issue + LDS"},
        "adjoint": {"kernel": "sa_adjoint_nn_lds_kernel (+ equalize)", "ms": round(adj, 3), "G_pixel_taps_per_s": round(npix * ntap / adj / 1e6, 1),
                    "algorithmic_bytes": int(byt + 4.0 * size ** 3), "GBps": round((byt + 4.0 * size ** 3) / adj / 1e6, 1),
                    "bound": "two passes over the taps in LDS (ds_add pre-summation) + one global atomic per touched cell"}}
    return out


def cpu_baseline(shape, threads):
    """The CPU restatement of the reference path (oracle/, validated against the real reference by the
    golden vectors) timed on this host: one warm-up + four timed full-size samples."""
    import numpy as np
    import torch

    from fetalsyngen_amd.phantom import make_seed_volumes
    from oracle import fsg_oracle as O

    torch.set_num_threads(threads)
    O.REFERENCE_LOOPS = True  # the zooms as the reference's per-slice Python loops (utils/generation.py:374-386), same values
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
            "sample": f"1 warm-up + {len(timed)} timed {shape[0]}^3 volumes, all gates on, torch CPU ops, zooms in the reference's "
                      f"per-slice loop form ({sum(timed):.2f} s, {min(timed):.2f}-{max(timed):.2f} s per volume)",
            "reference_anchor": REFERENCE_ANCHOR}


# ----------------------------------------------------------------------------------------------------
class Ranks:
    """The little cross-rank traffic a benchmark of independent replicas needs (gloo): barrier, max of a time, gather of
    small Python objects."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        if world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            self.dist = dist

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, x: float) -> float:
        if self.world == 1:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def run_dry_plan(args, R: Ranks):
    """Host plans only: what the Python side does per sample before the one native call (no GPU, no HIP library)."""
    from fetalsyngen_amd import sharding

    shape = (args.size,) * 3
    gen = build_generator(shape, "cuda:0", args.rng if args.rng != "reference" else "device")
    digest = 0.0
    if args.rng == "keyed":  # the host side of a keyed sample: the key and the C draws (fsg_keyed_draw, no GPU)
        kc = gen.keyed_context(shape)
        for i in range(args.warmup):
            kc.draws(sharding.sample_key(1234, R.rank + R.world * i))
        R.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            d = kc.draws(sharding.sample_key(1234, R.rank + R.world * (args.warmup + i)))
            digest += d.spacing + d.gamma
    else:
        for i in range(args.warmup):
            sharding.seed_for_sample(1234, R.rank + R.world * i)
            gen.plan_only(shape)
        R.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            sharding.seed_for_sample(1234, R.rank + R.world * (args.warmup + i))
            plans = gen.plan_only(shape)
            digest += float(plans[1].mus.sum())
    R.barrier()
    dt_rank = time.perf_counter() - t0
    dt = R.max(dt_rank)
    reports = R.gather({"rank": R.rank, "device": "none (dry plan)", "uuid": None, "pid": os.getpid(),
                        "plans_per_s": round(args.steps / dt_rank, 1), "digest": round(digest, 3)})
    if R.rank == 0:
        print(json.dumps({
            "metric": "host plans/sec (dry plan: all per-sample host draws, no GPU work)", "value": round(R.world * args.steps / dt, 1),
            "unit": "plans/s", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64/f32 host", "data": "synthetic", "dry_plan": True, "rng": args.rng,
            "config": {"workload": f"host plans of BASELINE configs[1] ({args.size}^3), no device work"},
            "ranks_seen": sorted(r["rank"] for r in reports), "ranks": reports}), flush=True)


def run(args, rank, world, local):
    import numpy as np
    import torch

    R = Ranks(rank, world)
    # one process per GPU: keep each rank's host-side plans (small numpy / torch CPU ops) on its share of the cores
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(world, 1))))
    if args.dry_plan:
        run_dry_plan(args, R)
        R.close()
        return 0

    import ctypes

    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import MemorySynthDataset, SeedBank
    from fetalsyngen_amd.data.staging import PrefetchingStream
    from fetalsyngen_amd.phantom import make_seed_volumes

    _lib.load()  # fail loudly if the HIP library is missing
    if args.tune:
        _lib.load().fsg_set_tuning(args.tune)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to benchmark")
    if os.environ.get("FSG_BENCH_SHARE_GPU0"):  # rehearsal of the N>1 path on a 1-GPU box: every rank on cuda:0
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    device = f"cuda:{local}"
    torch.cuda.set_device(device)
    props = torch.cuda.get_device_properties(local)
    shape = (args.size,) * 3
    nvox = int(np.prod(shape))

    # inputs resident in HBM before any timed region: 4 distinct label volumes (the same four on every rank, so that
    # the batch of config3 does not depend on N); further "subjects" are cheap on-device transforms of these
    base_banks, base_segs = [], []
    for v in range(4):
        seg, seeds = make_seed_volumes(shape, variant=v)
        base_banks.append(SeedBank(seeds, device))
        base_segs.append(torch.from_numpy(seg).to(device))

    def subject(k):
        """Label volumes of synthetic subject k: base variant k % 4, rolled 8*(k//4) voxels along y, mirrored in z for odd k//4."""
        b, r = k % 4, k // 4
        if r == 0:
            return base_segs[b], base_banks[b]
        fn = (lambda x: torch.roll(x, 8 * r, 1).flip(2).contiguous()) if r % 2 else (lambda x: torch.roll(x, 8 * r, 1).contiguous())
        return fn(base_segs[b]), base_banks[b].transformed(fn)

    gen = build_generator(shape, device, args.rng)
    gen.prewarm()  # static per-axis tables of this configuration -> device (outside the timed region)
    if not os.environ.get("FSG_BENCH_NO_RESERVE"):
        gen.reserve(shape, samples_in_flight=12)  # memory pool sized for the samples the host keeps in flight

    result_extra = {}
    if rank == 0 and not args.no_microbench:
        result_extra["blur_microbench"] = blur_microbench(shape, device)

    streams = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else [None]
    mus_seen = []

    keyed = args.rng == "keyed"

    def step(i):
        if keyed:
            key = sharding.sample_key(1234, rank + world * i)
        else:
            key = None
            sharding.seed_for_sample(1234, rank + world * i)
        k = i % 4
        st = streams[i % len(streams)]
        if st is None:
            # a stream of samples knows what comes next: sample i + 1's draw job rides in sample i's floor(min) launch
            # (fsg_keyed_sample_run's look-ahead; --no-look-ahead switches it off)
            nxt = sharding.sample_key(1234, rank + world * (i + 1)) if keyed and args.look_ahead else None
            out, seg_d, _img, p = gen._pipeline(None, base_segs[k], base_banks[k], {}, scale01=True, key=key, next_key=nxt)
        else:
            with torch.cuda.stream(st):
                out, seg_d, _img, p = gen._pipeline(None, base_segs[k], base_banks[k], {}, scale01=True, key=key)
        return out, seg_d, p

    # ---- headline: BASELINE configs[1], W untimed + K timed steps, barrier + synchronize on both sides ----------
    # blur timing inside the timed region: HIP events recorded on the launch stream around the blur launches of every 4th
    # sample, inside fsg_sample_run (FetalSynthGen.blur_events / blur_events_every)
    # Setup, untimed and outside the W warm-up steps the caller asked for: the first sample loads every code object (~240 ms),
    # the next dozen still run with cold interpreter / allocator / clock state (host 300-400 us per sample against 260 later:
    # gpurun_out r4m, tools/warmup_curve.py).  Like prewarm() and reserve() above this is one-off process start-up, so a short
    # --warmup measures the same steady state as a long one; reported as "setup_samples".
    setup_samples = 0 if os.environ.get("FSG_BENCH_NO_SETUP_SAMPLES") else int(os.environ.get("FSG_BENCH_SETUP_SAMPLES", "24"))
    for i in range(setup_samples):
        step(10_000_000 + i)
    torch.cuda.synchronize()
    # everything alive now (modules, tables, caches) is long-lived: keep the cyclic collector from re-scanning it inside the
    # timed steps (a full collection is ~1 ms, a sixth of the driver's 20-step region)
    import gc
    gc.collect()
    gc.freeze()
    gen.blur_events = []
    gen.blur_events_every = 4  # every 4th sample: each event record is a barrier packet in the launch queue (~5.5 us of bubble)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    _drop_events(gen.blur_events)
    R.barrier()
    torch.cuda.synchronize()
    edge = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if os.environ.get("FSG_BENCH_EDGES") else None
    t0 = time.perf_counter()
    if edge:
        edge[0].record()
    for i in range(args.steps):
        _o, _s, p = step(args.warmup + i)
        mus_seen.append(p["resample_params"]["spacing"])
    if edge:
        edge[1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_sync = time.perf_counter() - t0
    R.barrier()
    dt_rank = time.perf_counter() - t0
    dt = R.max(dt_rank)
    if edge and rank == 0:  # diagnostic: where the timed region's fixed part goes (stderr, not part of the line)
        print(json.dumps({"edges": {"wall_ms": round(dt_rank * 1e3, 3), "enqueue_done_ms": round(t_enq * 1e3, 3), "sync_done_ms": round(t_sync * 1e3, 3),
                                    "gpu_span_ms": round(edge[0].elapsed_time(edge[1]), 3)}}), file=sys.stderr, flush=True)

    # ---- per-launch times of the same samples' kind, UNTIMED: HIP events behind every launch of 16 further samples ------------
    gen.blur_events_save, gen.blur_events = gen.blur_events, None
    gen.stage_traces = []
    for i in range(16):
        step(20_000_000 + i)
    torch.cuda.synchronize()
    traces, gen.stage_traces = gen.stage_traces, None
    gen.blur_events = gen.blur_events_save
    roofline_kernels = stage_rooflines(traces, nvox)
    for tr in traces:
        tr.close()
    # what one event record costs the launch stream by itself (a barrier packet): 33 back-to-back records, median gap.
    # `us_net` = `us` minus that bubble; the net figures add up to the un-instrumented step (cross-check below).
    from fetalsyngen_amd import kernels as _K
    evs = [_lib.load().fsg_event_create() for _ in range(33)]
    st_raw = _K._stream(torch.device(device))
    torch.cuda.synchronize()
    for e in evs:
        _lib.load().fsg_event_record(e, st_raw)
    torch.cuda.synchronize()
    gaps, ms_ = [], ctypes.c_float()
    for e0, e1 in zip(evs, evs[1:]):
        _lib.check(_lib.load().fsg_event_elapsed_ms(e0, e1, ctypes.byref(ms_)), "fsg_event_elapsed_ms")
        gaps.append(ms_.value * 1e3)
    for e in evs:
        _lib.load().fsg_event_destroy(e)
    bubble = float(np.median(gaps))
    net = 0.0
    for k_, v_ in roofline_kernels.items():
        if isinstance(v_, dict):
            v_["us_net"] = round(max(v_["us"] - bubble, 0.0), 2)
            v_["frac_net"] = (round(v_["algorithmic_bytes"] / v_["us_net"] / 1e3 / HBM_PEAK_GBS, 4) if v_["us_net"] > 0 else None)
            net += v_["us_net"] * v_["launches_per_sample"]
    roofline_kernels["_event_bubble_us"] = round(bubble, 2)
    roofline_kernels["_sum_us_net_per_sample"] = round(net, 2)

    lib = _lib.load()
    blur_total_ms, sections, lows = 0.0, [], []
    ms = ctypes.c_float()
    for e0, e1, pl, low in gen.blur_events:
        _lib.check(lib.fsg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "fsg_event_elapsed_ms")
        blur_total_ms += ms.value
        sections.append(pl)
        lows.append(low)
    _drop_events(gen.blur_events)
    gen.blur_events = None
    # What the events bracket.  r03 default: the FUSED pair (csrc/fsg_blur_rs.hip) -- blur and down-sampling of an axis as one
    # operator: x launch reads N, writes N m0/n0; y,z launch reads N m0/n0, writes M (+ the noise draw).  Unfused
    # (--tune 16384, or a configuration outside the fused domain): x pass + fused y,z pass, 8 B/voxel each.
    size_v = float(args.size)
    fused = [bool(lib.fsg_blur_resample_supported(args.size, args.size, args.size, lo[0], lo[1], lo[2],
                                                  *[2 * dict(sec).get(a_, 0) + 1 for a_ in range(3)])) for sec, lo in zip(sections, lows)]
    alg_bytes, traffic_bytes, nlaunch, npass = 0.0, 0.0, 0, 0
    rs_table = None
    rs_files = sorted((REPO / "profiles").glob("r*_blur_rs_pmc.json"))
    if rs_files:
        rs_table = json.loads(rs_files[-1].read_text())["sizes"].get(str(args.size))

    def rs_ratio(prefix, radius):
        if not rs_table:
            return None
        pts = {int(k[len(prefix):].split(",")[0].rstrip(">")): v["traffic_over_algorithmic"] for k, v in rs_table.items()
               if k.startswith(prefix)}
        return pts[min(pts, key=lambda r_: abs(r_ - radius))] if pts else None

    traffic_known = True
    for sec, lo, fz in zip(sections, lows, fused):
        npass += len(sec)
        if fz:
            bx = 4.0 * nvox + 4.0 * nvox * lo[0] / size_v
            byz = 4.0 * nvox * lo[0] / size_v + 4.0 * lo[0] * lo[1] * lo[2]
            alg_bytes += bx + byz
            nlaunch += 2
            radii = dict(sec)
            rx = rs_ratio("blur_rs_x_kernel<", radii.get(0, 1))
            ryz = rs_ratio("blur_rs_yz_kernel<", max(radii.get(1, 1), radii.get(2, 1)))
            if rx is None or ryz is None:
                traffic_known = False
            else:
                traffic_bytes += rx * bx + ryz * byz
        else:
            launches = _blur_launch_plan(sec)
            alg_bytes += 8.0 * nvox * len(launches)
            nlaunch += len(launches)
            t_l, _n = blur_traffic([sec], args.size)
            if t_l is None:
                traffic_known = False
            else:
                traffic_bytes += t_l * len(launches)
    us_launch = blur_total_ms * 1e3 / max(nlaunch, 1)
    bytes_launch = alg_bytes / max(nlaunch, 1)
    traffic_launch = traffic_bytes / max(nlaunch, 1) if (traffic_known and nlaunch) else None
    achieved = bytes_launch / us_launch / 1e3 if nlaunch else 0.0
    n_fused = sum(fused)
    # whole step: SURVEY 8(d) per-kernel algorithmic bytes with the fusions as built (gamma/bias in the warp epilogue: 0; blur
    # and down-sampling per axis; K9+K10 one evaluation): K1 5 + warp 8 (image) + 5 (uint8 label read, float32 label write)
    # + x launch (4 + 4 mu_x) + y,z launch (4 mu_x + 4 mu) + K9 (4 mu + 4) B/voxel, mu_x = m/n, mu = M/N
    mus_x = [int(size_v * 0.5 / s_[0]) / size_v for s_ in mus_seen if s_]
    mu_x = float(np.mean(mus_x)) if mus_x else 1.0
    mu = float(np.mean([v ** 3 for v in mus_x])) if mus_x else 1.0
    step_bytes = (26.0 + 8.0 * mu_x + 8.0 * mu) * nvox

    result = {
        "metric": "synthetic volumes/sec at 256^3 (full deform+GMM+blur+resample path)",
        "value": round(world * args.steps / dt, 3),
        "unit": "volumes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "setup_samples": setup_samples,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: single {args.size}^3 label volume per step, full path, all gates on",
                   "rng": args.rng, "inputs": "uint8 seed labels (and the uint16 code volume built from them, which the GMM draw reads) + fp32 segmentation (and its cached uint8 copy, which the label gather reads) resident in HBM",
                   "outputs": "fp32 [0,1] image + fp32 labels in HBM", "volumes_per_rank": args.steps,
                   "parallelism": f"{world} independent replicas (no collective)", "streams_per_gpu": args.streams,
                   **({"tuning_flags": args.tune} if args.tune else {})},
        "roofline": {"bound": "hbm",
                     "kernel": ("separable blur, fused per axis with the down-sampling (csrc/fsg_blur_rs.hip): x launch "
                                "(blur_rs_x_kernel) + y,z launch with the noise epilogue (blur_rs_yz_kernel)"
                                if n_fused == len(sections) and sections else
                                "separable 3-pass blur = x pass (blur_strided_v4) + fused y,z pass (blur_yz_fused_kernel)")
                               + "; HIP events on the launch stream around those launches of every 4th timed sample (rank 0)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": (None if traffic_launch is None else int(traffic_launch)),
                     "us_per_launch": round(us_launch, 2), "launches_timed": nlaunch,
                     "algorithmic_bytes_per_launch": int(bytes_launch),
                     "accounting": "bytes one launch must move, averaged over the timed launches: fused x launch 4N + 4N m/n "
                                   "(volume read once, x-resampled volume written once), fused y,z launch 4N m/n + 4M (M = "
                                   "low-res voxels of that sample); unfused launches 8 B/voxel.  traffic = those bytes x the "
                                   "traffic/algorithmic ratio of the launch's kernel in the committed PMC table "
                                   f"(profiles/{rs_files[-1].name if rs_files else 'none'}: 1.00-1.03)",
                     "samples_fused": n_fused, "samples_timed": len(sections),
                     "axis_passes_per_sample": round(npass / max(len(sections), 1), 2),
                     "launches_per_sample": round(nlaunch / max(len(sections), 1), 2)},
        "roofline_step": {"bound": "hbm", "bytes_per_voxel": "26 + 8*mu_x + 8*mu (K1 5, warp 8 image + 1 uint8 label read + 4 float32 label write, "
                                                             "blur+resample x 4+4mu_x, y,z 4mu_x+4mu, K9/K10 4+4mu)",
                          "mu_mean": round(mu, 4), "mu_x_mean": round(mu_x, 4), "algorithmic_bytes_per_step": int(step_bytes),
                          "achieved": round(step_bytes / (dt / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    roofline_kernels["_check_sum_net_le_step"] = bool(roofline_kernels["_sum_us_net_per_sample"] <= dt / args.steps * 1e6 * 1.03)
    result["roofline_kernels"] = roofline_kernels
    result.update(result_extra)

    # ---- config3: a batch of B volumes dealt i % N (strong scaling), per-sample keys independent of N -------------
    c3 = None
    if not args.no_config3:
        B = args.batch_volumes
        mine = list(sharding.shard(B, rank, world))
        subj = {i: subject(i) for i in mine}
        walls, outs = [], {}
        for rep in range(4):  # first repetition is a warm-up of the allocator for this many live outputs
            outs.clear()
            torch.cuda.synchronize()
            R.barrier()
            t0 = time.perf_counter()
            for i in mine:
                if keyed:
                    ckey = sharding.sample_key(4321, i)
                else:
                    ckey = None
                    sharding.seed_for_sample(4321, i)
                o, s_, _im, _p = gen._pipeline(None, subj[i][0], subj[i][1], {}, scale01=True, key=ckey)
                outs[i] = (o, s_)
            torch.cuda.synchronize()
            R.barrier()
            walls.append(R.max(time.perf_counter() - t0))
        sums = {i: [float(o.double().sum()), float(s_.double().sum())] for i, (o, s_) in outs.items()}
        allsums = {}
        for d in R.gather(sums):
            allsums.update(d)
        outs.clear()
        subj.clear()
        wall = float(np.median(walls[1:]))
        c3 = {"workload": f"BASELINE configs[2]: batch of {B} distinct {args.size}^3 label volumes, sample i on rank i % {world}, "
                          "keys (base_seed, i) independent of N, outputs kept in HBM",
              "scaling": "strong", "volumes": B, "wall_ms": round(wall * 1e3, 3), "wall_ms_runs": [round(w * 1e3, 3) for w in walls[1:]],
              "volumes_per_s": round(B / wall, 1),
              "checksum": round(sum(v[0] for v in allsums.values()), 6), "label_checksum": round(sum(v[1] for v in allsums.values()), 1),
              "checksum_note": "sum over the batch of each output's float64 voxel sum: equal for every N when the batch is independent of the sharding"}

    # ---- config5: streaming epoch into a consumer, (a) reference contract on the CPU, (b) device-resident ----------
    c5 = None
    if not args.no_config5:
        ds = MemorySynthDataset(gen, base_segs, base_banks)
        n_stream = args.stream_volumes
        idx = [rank + world * j for j in range(n_stream)]
        c5 = {"workload": f"BASELINE configs[4]: {n_stream} volumes per rank streamed through PrefetchingStream into a consumer "
                          f"that touches every image ({args.size}^3, 4 cached subjects per rank)", "scaling": "weak",
              "volumes_per_rank": n_stream}
        import torch as _t

        modes = (("cpu_contract", dict(to_host=True, depth=3)),
                 ("cpu_contract_batched", dict(to_host=True, depth=2, batch_size=4, batch_streams=2)),
                 ("cpu_f16_u8_batched", dict(to_host=True, depth=2, batch_size=4, batch_streams=2, image_dtype=_t.float16,
                                             label_dtype=_t.uint8)),
                 ("device_resident", dict(to_host=False)),
                 ("device_resident_batched", dict(to_host=False, batch_size=4, batch_streams=2)))
        for key, kw in modes:
            acc = torch.zeros((), dtype=torch.float64, device=device)
            host_acc = 0.0
            for item in PrefetchingStream(ds, idx[:16], base_seed=99, **kw):  # warm-up (ring allocation, pinning)
                pass
            torch.cuda.synchronize()
            R.barrier()
            t0 = time.perf_counter()
            for item in PrefetchingStream(ds, idx, base_seed=99, **kw):
                img = item["image"]
                if img.is_cuda:
                    acc += img.view(-1)[::4097].sum()
                else:
                    host_acc += float(img.view(-1)[::4097].sum())
            torch.cuda.synchronize()
            R.barrier()
            dts = R.max(time.perf_counter() - t0)
            outputs = {"cpu_contract": "float32 image (1,H,W,D) + int64 labels on the CPU (pinned ring), reference data/datasets.py:315-323",
                       "cpu_contract_batched": "the same contract, 4 samples per native call on 2 HIP streams, collated (4,1,H,W,D), one D2H copy per tensor and batch",
                       "cpu_f16_u8_batched": "opt-in: float16 image + uint8 labels on the CPU (48 MiB instead of 192 MiB per volume), batches of 4",
                       "device_resident": "float32 image + uint8 labels in HBM (the fused warp writes the uint8 labels itself)",
                       "device_resident_batched": "float32 image + uint8 labels in HBM, 4 samples per native call on 2 HIP streams"}[key]
            c5[key] = {"volumes_per_s": round(world * n_stream / dts, 1), "s": round(dts, 3), "outputs": outputs}
        del ds

    reports = R.gather({"rank": rank, "device": device, "name": props.name, "uuid": str(getattr(props, "uuid", "")) or None,
                        "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid(),
                        "volumes_per_s": round(args.steps / dt_rank, 1)})
    if rank == 0:
        result["ranks_seen"] = sorted(r["rank"] for r in reports)
        result["ranks"] = reports
        if c3 is not None:
            result["config3"] = c3
        if c5 is not None:
            result["config5"] = c5
            result["value_cpu_contract"] = c5["cpu_contract"]["volumes_per_s"]
        if world == 1 and not args.no_sr:
            result["config4_sr"] = config4_sr(device)
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            result["cpu_baseline"] = cpu_baseline(shape, threads)
            result["gpu_over_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
            result["gpu_over_reference_anchor"] = round(result["value"] / REFERENCE_ANCHOR["value"], 1)
        print(json.dumps(result), flush=True)
    R.close()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)  # parent of N ranks: no GPU call in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or unset "
                         "WORLD_SIZE to let bench.py start its own ranks")
    return run(args, rank, world, local)


if __name__ == "__main__":
    sys.exit(main())
