#!/usr/bin/env python3
"""Per-kernel microbenchmark at 256^3 (or --size): HIP-event timing of each kernel of the path in
isolation, with the algorithmic bytes of DESIGN.md.  Also the target of `rocprofv3 --pmc` runs.

    python tools/kernel_bench.py [--size 256] [--reps 20] [--only warp,blur] [--cold]
"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from fetalsyngen_amd import kernels as K, tables as T
from fetalsyngen_amd.phantom import make_seed_volumes, combined_seed_labels

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--m", type=int, default=171, help="low-res size for the resample kernels")
ap.add_argument("--rot", type=float, default=12.0, help="rotation (degrees) about each axis")
ap.add_argument("--tune", type=int, default=0, help="fsg_set_tuning flags")
ap.add_argument("--variant", type=int, default=0, help="fsg_warp_set_variant")
ap.add_argument("--zoom-ty", type=int, default=0, help="fsg_zoom_set_tuning: output rows per workgroup (0 = default)")
args = ap.parse_args()
dev = "cuda:0"
from fetalsyngen_amd import _lib
_lib.load().fsg_set_tuning(args.tune)
if args.zoom_ty:
    _lib.check(_lib.load().fsg_zoom_set_tuning(args.zoom_ty, 12000), "fsg_zoom_set_tuning")
_lib.load().fsg_warp_set_variant(args.variant)
n = args.size
shape = (n, n, n)
N = n ** 3
only = set(args.only.split(",")) if args.only else None

def timeit(name, fn, bytes_alg):
    if only and not any(name.startswith(o) for o in only):
        return
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(args.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    print(json.dumps({"kernel": name, "us": round(us, 2), "alg_MB": round(bytes_alg / 1e6, 1),
                      "GBps": round(bytes_alg / us / 1e3, 1)}), flush=True)

rs = np.random.RandomState(0)
seg, seeds = make_seed_volumes(shape)
lab8 = torch.from_numpy(combined_seed_labels(seeds, {1: 3, 2: 4, 3: 2, 4: 5})).to(dev)
segf = torch.from_numpy(seg).to(dev)
seg8 = segf.to(torch.uint8)
mus = torch.from_numpy((25 + 200 * rs.rand(50)).astype(np.float32)).to(dev)
sig = torch.from_numpy((5 + 20 * rs.rand(50)).astype(np.float32)).to(dev)
img = K.gmm_sample(lab8, mus, sig, seed=1, stream_id=1)

timeit("gmm_u8_philox", lambda: K.gmm_sample(lab8, mus, sig, seed=1, stream_id=1), 5 * N)

# deformation as the default config draws it
from fetalsyngen_amd.utils.generation import make_affine_matrix
r = np.deg2rad(args.rot)
A = make_affine_matrix([r, -r, r], [0.01, -0.02, 0.015], [1.05, 0.95, 1.02]).astype(np.float32)
fsz = max(1, round(0.05 * n))
fs = torch.from_numpy((rs.randn(fsz, fsz, fsz, 3) * 2.5).astype(np.float32)).to(dev)
ft, _ = T.zoom_tables((fsz,) * 3, np.array(shape) / fsz)
c = (np.array(shape) - 1) / 2
spec = K.DeformSpec(shape, A, c, c.astype(np.float32), True, fs, K.DeviceTables(ft, dev), device=dev)
bsz = max(1, round(0.012 * n))
bias = torch.from_numpy((rs.randn(bsz, bsz, bsz) * 0.2).astype(np.float32)).to(dev)
bt, _ = T.zoom_tables((bsz,) * 3, np.array(shape) / bsz)
btabs = K.DeviceTables(bt, dev)
mm6 = K.coords_minmax(spec)
timeit("coords_minmax", lambda: K.coords_minmax(spec), 0)
timeit("warp_f32_f32_epi", lambda: K.warp(spec, mm6, src_lin=img, src_nn=segf, gamma=1.1, bias=bias, bias_tabs=btabs), 16 * N)
timeit("warp_f32_u8_epi", lambda: K.warp(spec, mm6, src_lin=img, src_nn=seg8, gamma=1.1, bias=bias, bias_tabs=btabs), 10 * N)
timeit("warp_lin_only", lambda: K.warp(spec, mm6, src_lin=img), 8 * N)
timeit("warp_nn_f32_only", lambda: K.warp(spec, mm6, src_nn=segf), 8 * N)

specw = K.DeformSpec(shape, A, c, c.astype(np.float32), True, fs, K.DeviceTables(ft, dev), device=dev)
timeit("deform_rows_prepare", lambda: specw.prepare_rows(bias, btabs), 0)
specw.prepare_rows(bias, btabs)
timeit("coords_minmax_ws", lambda: K.coords_minmax(specw), 0)
timeit("warp_f32_f32_epi_ws", lambda: K.warp(specw, mm6, src_lin=img, src_nn=segf, gamma=1.1, bias=bias, bias_tabs=btabs), 16 * N)
timeit("warp_f32_u8_epi_ws", lambda: K.warp(specw, mm6, src_lin=img, src_nn=seg8, gamma=1.1, bias=bias, bias_tabs=btabs), 10 * N)
timeit("warp_f32_u8tof32_epi_ws", lambda: K.warp(specw, mm6, src_lin=img, src_nn=seg8, gamma=1.1, bias=bias, bias_tabs=btabs, nn_out=torch.float32), 13 * N)
timeit("warp_lin_only_ws", lambda: K.warp(specw, mm6, src_lin=img), 8 * N)
timeit("warp_nn_f32_only_ws", lambda: K.warp(specw, mm6, src_nn=segf), 8 * N)

for sigma in (0.6, 1.3, 1.77):
    taps = T.gaussian_taps(sigma)
    for axis in range(3):
        timeit(f"blur_axis{axis}_R{len(taps)//2}", lambda: K.blur_axis(img, axis, taps), 8 * N)
    timeit(f"blur_yz_fused_R{len(taps)//2}", lambda: K.blur_yz(img, taps, taps), 16 * N)

m = args.m
stds, new, fac, rtabs = T.resample_plan(shape, [0.5] * 3, [0.5 * n / m] * 3, 0.5)
rt = K.DeviceTables(rtabs, dev)
M = int(np.prod(new))
low = K.resample_noise(img, rt, noise_std=9.0, seed=3, stream_id=2)
timeit(f"resample_noise_m{new[0]}", lambda: K.resample_noise(img, rt, noise_std=9.0, seed=3, stream_id=2), 4 * N + 4 * M)
# K6 + K7 + K8 as the fused pair (csrc/fsg_blur_rs.hip), per launch
from fetalsyngen_amd import _lib as _L
import ctypes as _C
for m_rs in (96, 128, args.m, 220, 250):
    stds_r, new_r, _fr, tabs_r = T.resample_plan(shape, [0.5] * 3, [0.5 * n / m_rs] * 3, 0.5)
    taps_r = [T.gaussian_taps(float(s_)) for s_ in stds_r]
    if min(len(t_) for t_ in taps_r) < 3:
        continue
    rt_r = K.DeviceTables(tabs_r, dev)
    Mr = int(np.prod(new_r))
    mid = torch.empty((new_r[0], n, n), dtype=torch.float32, device=dev)
    outr = torch.empty(tuple(new_r), dtype=torch.float32, device=dev)
    fp = _C.POINTER(_C.c_float)
    lib = _L.load()
    st_ = K._stream(img)
    def _x():
        _L.check(lib.fsg_blur_resample_x_f32(img.data_ptr(), n, n, n, rt_r.ptrs[0], new_r[0], taps_r[0].ctypes.data_as(fp), len(taps_r[0]), mid.data_ptr(), st_), "x")
    def _yz():
        _L.check(lib.fsg_blur_resample_yz_noise_f32(mid.data_ptr(), new_r[0], n, n, rt_r.ptrs[1], rt_r.ptrs[2], new_r[1], new_r[2], taps_r[1].ctypes.data_as(fp), len(taps_r[1]),
                                                    taps_r[2].ctypes.data_as(fp), len(taps_r[2]), 2, None, 3, 2, 9.0, outr.data_ptr(), st_), "yz")
    timeit(f"blur_rs_x_m{new_r[0]}_R{len(taps_r[0])//2}", _x, 4 * N + 4 * N * new_r[0] / n)
    timeit(f"blur_rs_yz_m{new_r[0]}_R{len(taps_r[0])//2}", _yz, 4 * N * new_r[0] / n + 4 * Mr)
    def _yz0():
        _L.check(lib.fsg_blur_resample_yz_noise_f32(mid.data_ptr(), new_r[0], n, n, rt_r.ptrs[1], rt_r.ptrs[2], new_r[1], new_r[2], taps_r[1].ctypes.data_as(fp), len(taps_r[1]),
                                                    taps_r[2].ctypes.data_as(fp), len(taps_r[2]), 0, None, 3, 2, 9.0, outr.data_ptr(), st_), "yz")
    timeit(f"blur_rs_yz_nonoise_m{new_r[0]}_R{len(taps_r[0])//2}", _yz0, 4 * N * new_r[0] / n + 4 * Mr)

bt2, _ = T.zoom_tables(new, 1 / np.asarray(fac))
zt = K.DeviceTables(bt2, dev)
mm = K.zoom_minmax(low, zt)
timeit(f"zoom_minmax_m{new[0]}", lambda: K.zoom_minmax(low, zt), 4 * M)
timeit(f"zoom_normalise_m{new[0]}", lambda: K.zoom_normalise(low, zt, mm, 1), 4 * M + 4 * N)
slots = K.zoom_minmax_sharded(low, zt)
def _sharded_raw():  # slots stay allocated: the timed call is the launch alone (K.zoom_minmax_sharded also uploads the initial keys)
    from fetalsyngen_amd import _lib
    sx, sy, sz = low.shape
    dx, dy, dz = zt.lengths
    tx, ty, tz = zt.ptrs
    _lib.check(_lib.load().fsg_zoom3d_minmax_sharded_f32(low.data_ptr(), sx, sy, sz, tx, ty, tz, dx, dy, dz, slots.data_ptr(),
                                                         int(slots.shape[0]), torch.cuda.current_stream().cuda_stream), "sharded")
timeit(f"zoom_sharded_minmax_m{new[0]}", _sharded_raw, 4 * M)
timeit(f"zoom_sharded_normalise_m{new[0]}", lambda: K.zoom_normalise(low, zt, slots, 1), 4 * M + 4 * N)
timeit("reduce_minmax", lambda: K.reduce_minmax(img), 4 * N)
timeit("scale", lambda: K.scale(img, K.reduce_minmax(img), 1), 12 * N)
