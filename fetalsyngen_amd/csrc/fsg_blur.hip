// fsg_blur.hip -- K6: one axis pass of the separable Gaussian blur.
//
// Replaces one `conv3d` call of `gaussian_blur_3d` (utils/generation.py:84-110): 1-D taps along one
// axis, zero padding, no border renormalisation.  Algorithmic traffic: 4 B read + 4 B written per voxel
// per pass; taps and halos are cache/LDS traffic.
//
// Round-1 kernels:
//   blur_generic_kernel : any shape/alignment, one thread per voxel (correctness fallback).
//   blur_strided_v4     : axis 0 / 1 (stride = inner floats), inner % 4 == 0: each lane owns a float4
//                         column segment and produces TL consecutive outputs along the blur axis from
//                         TL + 2R coalesced row loads (register sliding window, fully unrolled).
//   blur_contig_lds     : axis 2 (z, contiguous): rows staged in LDS with zero halos, each lane produces
//                         4 consecutive outputs from aligned ds_read_b128 windows.
#include "fsg_common.h"

namespace {

constexpr int MAX_TAPS = 129;  // radius <= 64 (sigma <= 21)

struct Taps { float w[MAX_TAPS]; };

// ---- generic ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blur_generic_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int outer, int len, int inner,
                                                           const float* __restrict__ taps, int ntaps) {
  // volume viewed as (outer, len, inner); blur along `len`
  const size_t total = (size_t)outer * len * inner;
  const int R = ntaps >> 1;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t in = e % inner;
    const size_t rest = e / inner;
    const int l = (int)(rest % len);
    const size_t o = rest / len;
    const float* base = src + (o * len) * inner + in;
    float acc = 0.f;
    const int t0 = max(0, R - l), t1 = min(ntaps, len - l + R);
    for (int t = t0; t < t1; ++t) acc = fmaf(taps[t], base[(size_t)(l + t - R) * inner], acc);
    dst[e] = acc;
  }
}

// ---- strided axes, float4 lanes, register window ----------------------------------------------
// Volume viewed as (outer, len, inner4 float4 columns); blur along `len`.  A thread owns one float4 column
// and TL consecutive outputs: TL + 2R coalesced row loads, TL stores, taps from the kernel arguments.
// Block = 64 columns x 4 chunks (threadIdx.y), so that narrow slabs (axis 1: inner4 = nz/4 = 64) still
// fill every lane; grid: x = column groups, y = groups of 4 chunks, z = outer.
template <int R, int TL>
__global__ __launch_bounds__(256) void blur_strided_v4(const float4* __restrict__ src, float4* __restrict__ dst,
                                                       int len, int inner4, Taps T) {
  const int c = blockIdx.x * 64 + threadIdx.x;  // float4 column
  const int l0 = (blockIdx.y * 4 + threadIdx.y) * TL;
  if (c >= inner4 || l0 >= len) return;
  const size_t slab = (size_t)blockIdx.z * len * inner4;
  const float4* s = src + slab + c;
  float4 acc[TL];
#pragma unroll
  for (int o = 0; o < TL; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < TL + 2 * R; ++t) {
    const int l = l0 + t - R;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l >= 0 && l < len) v = s[(size_t)l * inner4];
#pragma unroll
    for (int o = 0; o < TL; ++o) {
      const int tap = t - o;  // input l0+t-R contributes to output l0+o with tap index t-o
      if (tap >= 0 && tap <= 2 * R) {
        const float w = T.w[tap];
        acc[o].x = fmaf(w, v.x, acc[o].x);
        acc[o].y = fmaf(w, v.y, acc[o].y);
        acc[o].z = fmaf(w, v.z, acc[o].z);
        acc[o].w = fmaf(w, v.w, acc[o].w);
      }
    }
  }
  float4* d = dst + slab + c;
#pragma unroll
  for (int o = 0; o < TL; ++o)
    if (l0 + o < len) d[(size_t)(l0 + o) * inner4] = acc[o];
}

// ---- contiguous axis: rows through LDS ----------------------------------------------------------
// block = 256 threads = 4 waves; each wave owns one row at a time: stages the row (nz floats) into LDS
// with RP zero floats either side (RP = R rounded up to 4), then every lane produces 4 outputs.
template <int R>
__global__ __launch_bounds__(256) void blur_contig_lds(const float* __restrict__ src, float* __restrict__ dst,
                                                       int rows, int nz, Taps T) {
  constexpr int RP = (R + 3) & ~3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pitch = nz + 2 * RP;  // nz % 4 == 0 -> pitch % 4 == 0
  float* row = lds + (size_t)wave * pitch;
  const int nz4 = nz >> 2;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)r * nz);
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)r * nz);
    if (lane < RP / 4) {
      reinterpret_cast<float4*>(row)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(row + RP + nz)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int q = lane; q < nz4; q += 64) reinterpret_cast<float4*>(row + RP)[q] = s4[q];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
    for (int q = lane; q < nz4; q += 64) {
      // window [4q - RP, 4q + 4 + RP) in row coordinates == LDS floats [4q, 4q + 4 + 2RP)
      float win[4 + 2 * RP];
#pragma unroll
      for (int u = 0; u < (4 + 2 * RP) / 4; ++u) {
        const float4 t = reinterpret_cast<const float4*>(row)[q + u];
        win[4 * u] = t.x; win[4 * u + 1] = t.y; win[4 * u + 2] = t.z; win[4 * u + 3] = t.w;
      }
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t <= 2 * R; ++t) {
        const float w = T.w[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(w, win[RP - R + e + t], o[e]);
      }
      d4[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// ---- long kernels (radius 9..64: BlurCortex draws sigma from a gamma distribution) -------------------------------
// Same data movement as the two kernels above with the radius as a run-time value: the input loop is not unrolled,
// every input row is offered to all TL outputs under a uniform predicate (useful work (2R+1)/(TL+2R)), taps are read
// from the kernel arguments with a wave-uniform index (scalar loads).
template <int TL>
__global__ __launch_bounds__(256) void blur_strided_long(const float4* __restrict__ src, float4* __restrict__ dst, int len,
                                                         int inner4, int R, Taps T) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int l0 = (blockIdx.y * 4 + threadIdx.y) * TL;
  if (c >= inner4 || l0 >= len) return;
  const size_t slab = (size_t)blockIdx.z * len * inner4;
  const float4* s = src + slab + c;
  float4 acc[TL];
#pragma unroll
  for (int o = 0; o < TL; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int t_lo = max(0, R - l0), t_hi = min(TL + 2 * R, len - l0 + R);  // rows inside the volume only
  for (int t = t_lo; t < t_hi; ++t) {
    const float4 v = s[(size_t)(l0 + t - R) * inner4];
#pragma unroll
    for (int o = 0; o < TL; ++o) {
      const int tap = t - o;
      if (tap >= 0 && tap <= 2 * R) {
        const float w = T.w[tap];
        acc[o].x = fmaf(w, v.x, acc[o].x);
        acc[o].y = fmaf(w, v.y, acc[o].y);
        acc[o].z = fmaf(w, v.z, acc[o].z);
        acc[o].w = fmaf(w, v.w, acc[o].w);
      }
    }
  }
  float4* d = dst + slab + c;
#pragma unroll
  for (int o = 0; o < TL; ++o)
    if (l0 + o < len) d[(size_t)(l0 + o) * inner4] = acc[o];
}

__global__ __launch_bounds__(256) void blur_contig_long(const float* __restrict__ src, float* __restrict__ dst, int rows, int nz,
                                                        int R, Taps T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int RP = (R + 3) & ~3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pitch = nz + 2 * RP;
  float* row = lds + (size_t)wave * pitch;
  const int nz4 = nz >> 2;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)r * nz);
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)r * nz);
    for (int q = lane; q < RP / 4; q += 64) {
      reinterpret_cast<float4*>(row)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(row + RP + nz)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int q = lane; q < nz4; q += 64) reinterpret_cast<float4*>(row + RP)[q] = s4[q];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int q = lane; q < nz4; q += 64) {
      const float* wnd = row + 4 * q + (RP - R);  // input index (4q + e + t - R) lives at wnd[e + t]
      float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
      float a = wnd[0], b = wnd[1], c = wnd[2];
      for (int t = 0; t <= 2 * R; ++t) {
        const float d = wnd[t + 3];
        const float w = T.w[t];
        o0 = fmaf(w, a, o0); o1 = fmaf(w, b, o1); o2 = fmaf(w, c, o2); o3 = fmaf(w, d, o3);
        a = b; b = c; c = d;
      }
      d4[q] = make_float4(o0, o1, o2, o3);
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// ---- fused y + z pass ----------------------------------------------------------------------------------------
// The y and z passes of the separable blur as ONE kernel: a workgroup owns a TY x TZ tile of one x-plane, stages the
// tile with its (Ry, Rz) halos in LDS (zeros outside the volume = the reference's zero padding), runs the y pass for
// the tile's rows over the z-halo'd width into a second LDS buffer, then the z pass from there to global memory.  The
// intermediate volume of the two-launch form (4 B/voxel written + 4 B/voxel read, and a launch) never exists.  Each pass
// accumulates its taps in ascending order with fmaf exactly as the single-axis kernels do, on the same values, so the
// result is bit-identical to running them one after the other.
constexpr int YZ_MAXR = 8;
constexpr int YZ_TL = 4;               // y outputs per thread (register sliding window); 4 beats 8 and 16 at 256^3 (25.3 / 26.7 / 32.7 us at R = 4: more workgroups per CU, the larger y halo is served by L2)
constexpr int YZ_ROWS = 4 * YZ_TL;     // y rows per workgroup

// Workgroup = YZ_ROWS consecutive y rows of one x-plane over the FULL z extent (nz <= 512), thread = (float4 z-column,
// y chunk), i.e. the body of blur_strided_v4 with TL = 8 whose results go to LDS rows (zero halo of RP floats either
// side) instead of HBM; after a barrier every wave runs the body of blur_contig_lds over its rows of that LDS tile.
template <int R>
__global__ __launch_bounds__(256) void blur_yz_fused_kernel(const float4* __restrict__ src, float* __restrict__ dst, int ny,
                                                            int nz, Taps T) {
  constexpr int RP = (R + 3) & ~3;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [YZ_ROWS][nz + 2 RP]
  const int inner4 = nz >> 2, pitch = nz + 2 * RP;
  const int tx = threadIdx.x, tyc = threadIdx.y;  // 64 x 4
  // 1-D grid of (plane, y tile) pairs, every XCD given a contiguous range of them: the two tiles that share a y halo run on
  // the same XCD one after the other, so the halo rows are served by its L2 instead of being read from HBM twice
  const int tiles_y = (ny + YZ_ROWS - 1) / YZ_ROWS;
  const int tile = xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int bx = tile / tiles_y;
  const int y0 = (tile - bx * tiles_y) * YZ_ROWS;
  const size_t plane4 = (size_t)bx * ny * inner4;
  // zero halos of every row
  for (int e = tyc * 64 + tx; e < YZ_ROWS * (2 * RP / 4); e += 256) {
    const int r = e / (2 * RP / 4), h = e - r * (2 * RP / 4);
    float* row = lds + (size_t)r * pitch;
    reinterpret_cast<float4*>(h < RP / 4 ? row : row + RP + nz)[h < RP / 4 ? h : h - RP / 4] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- y pass (axis 1) into LDS ----
  for (int c = tx; c < inner4; c += 64) {
    const int l0 = y0 + tyc * YZ_TL;
    const float4* s = src + plane4 + c;
    float4 acc[YZ_TL];
#pragma unroll
    for (int o = 0; o < YZ_TL; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < YZ_TL + 2 * R; ++t) {
      const int l = l0 + t - R;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l >= 0 && l < ny) v = s[(size_t)l * inner4];
#pragma unroll
      for (int o = 0; o < YZ_TL; ++o) {
        const int tap = t - o;
        if (tap >= 0 && tap <= 2 * R) {
          const float w = T.w[tap];
          acc[o].x = fmaf(w, v.x, acc[o].x);
          acc[o].y = fmaf(w, v.y, acc[o].y);
          acc[o].z = fmaf(w, v.z, acc[o].z);
          acc[o].w = fmaf(w, v.w, acc[o].w);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < YZ_TL; ++o)
      reinterpret_cast<float4*>(lds + (size_t)(tyc * YZ_TL + o) * pitch + RP)[c] = acc[o];
  }
  __syncthreads();
  // ---- z pass (axis 2) from LDS to HBM: wave w takes rows w, w+4, ... ----
  const int wave = tyc, lane = tx;
  for (int r = wave; r < YZ_ROWS; r += 4) {
    const int y = y0 + r;
    if (y >= ny) break;
    const float* row = lds + (size_t)r * pitch;
    float4* d4 = reinterpret_cast<float4*>(dst + ((size_t)bx * ny + y) * nz);
    for (int q = lane; q < inner4; q += 64) {
      float win[4 + 2 * RP];
#pragma unroll
      for (int u = 0; u < (4 + 2 * RP) / 4; ++u) {
        const float4 t = reinterpret_cast<const float4*>(row)[q + u];
        win[4 * u] = t.x; win[4 * u + 1] = t.y; win[4 * u + 2] = t.z; win[4 * u + 3] = t.w;
      }
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t <= 2 * R; ++t) {
        const float w = T.w[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(w, win[RP - R + e + t], o[e]);
      }
      d4[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

template <int R>
int launch_yz(const float* src, float* dst, int nx, int ny, int nz, const Taps& T, hipStream_t st) {
  constexpr int RP = (R + 3) & ~3;
  const size_t lds = (size_t)YZ_ROWS * (nz + 2 * RP) * sizeof(float);
  dim3 grid((unsigned)((ny + YZ_ROWS - 1) / YZ_ROWS) * (unsigned)nx), block(64, 4);
  hipLaunchKernelGGL(blur_yz_fused_kernel<R>, grid, block, lds, st, reinterpret_cast<const float4*>(src), dst, ny, nz, T);
  FSG_RETURN_LAUNCH();
}

template <int R>
int launch_strided(const float* src, float* dst, int outer, int len, int inner, const Taps& T, hipStream_t st) {
#ifndef FSG_BLUR_TL
#define FSG_BLUR_TL 24  // measured on MI355X at 256^3: 24 beats 16 (fewer halo re-reads) and 32 (register pressure)
#endif
  constexpr int TL = FSG_BLUR_TL;
  const int inner4 = inner / 4;
  const int chunks = (len + TL - 1) / TL;
  dim3 block(64, 4), grid((unsigned)((inner4 + 63) / 64), (unsigned)((chunks + 3) / 4), (unsigned)outer);
  hipLaunchKernelGGL((blur_strided_v4<R, TL>), grid, block, 0, st, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), len, inner4, T);
  FSG_RETURN_LAUNCH();
}

template <int R>
int launch_contig(const float* src, float* dst, int rows, int nz, const Taps& T, hipStream_t st) {
  constexpr int RP = (R + 3) & ~3;
  const size_t lds = (size_t)4 * (nz + 2 * RP) * sizeof(float);
  int grid = (rows + 3) / 4;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL((blur_contig_lds<R>), dim3(grid), dim3(256), lds, st, src, dst, rows, nz, T);
  FSG_RETURN_LAUNCH();
}

}  // namespace

extern "C" int fsg_blur_axis_f32(const float* src, float* dst, int nx, int ny, int nz, int axis, const float* taps,
                                 int ntaps, void* stream) {
  if (!src || !dst || src == dst || !taps) return FSG_E_BADARG;
  if (nx <= 0 || ny <= 0 || nz <= 0 || axis < 0 || axis > 2) return FSG_E_BADARG;
  if (ntaps <= 0 || (ntaps & 1) == 0) return FSG_E_BADARG;
  hipStream_t st = fsg_stream(stream);
  int outer, len, inner;
  if (axis == 0) { outer = 1; len = nx; inner = ny * nz; }
  else if (axis == 1) { outer = nx; len = ny; inner = nz; }
  else { outer = nx * ny; len = nz; inner = 1; }
  const size_t total = (size_t)nx * ny * nz;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(blur_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, outer, len, inner, taps,
                     ntaps);
  FSG_RETURN_LAUNCH();
}

// Fast paths take the taps by value (host pointer): no device round trip, graph-capturable.
extern "C" int fsg_blur_axis_taps_host_f32(const float* src, float* dst, int nx, int ny, int nz, int axis,
                                           const float* taps_host, int ntaps, void* stream) {
  if (!src || !dst || src == dst || !taps_host) return FSG_E_BADARG;
  if (nx <= 0 || ny <= 0 || nz <= 0 || axis < 0 || axis > 2) return FSG_E_BADARG;
  if (ntaps <= 0 || (ntaps & 1) == 0 || ntaps > MAX_TAPS) return FSG_E_BADARG;
  if ((size_t)nx * ny * nz > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  hipStream_t st = fsg_stream(stream);
  const int R = ntaps >> 1;
  Taps T;
  for (int t = 0; t < MAX_TAPS; ++t) T.w[t] = t < ntaps ? taps_host[t] : 0.f;
  const bool aligned = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  if (axis == 2) {
    const int rows = nx * ny;
    if (aligned && (nz & 3) == 0 && R >= 1 && R <= 8 && nz <= 4064) {  // 4 rows + halos within 64 KB of LDS
      switch (R) {
        case 1: return launch_contig<1>(src, dst, rows, nz, T, st);
        case 2: return launch_contig<2>(src, dst, rows, nz, T, st);
        case 3: return launch_contig<3>(src, dst, rows, nz, T, st);
        case 4: return launch_contig<4>(src, dst, rows, nz, T, st);
        case 5: return launch_contig<5>(src, dst, rows, nz, T, st);
        case 6: return launch_contig<6>(src, dst, rows, nz, T, st);
        case 7: return launch_contig<7>(src, dst, rows, nz, T, st);
        case 8: return launch_contig<8>(src, dst, rows, nz, T, st);
      }
    }
    if (aligned && (nz & 3) == 0 && R > 8 && nz + 2 * ((R + 3) & ~3) + 4 <= 4096) {  // 4 staged rows within 64 KB of LDS
      const int RP = (R + 3) & ~3;
      int grid = (rows + 3) / 4;
      if (grid > 8192) grid = 8192;
      hipLaunchKernelGGL(blur_contig_long, dim3(grid), dim3(256), (size_t)4 * (nz + 2 * RP + 4) * sizeof(float), st, src, dst,
                         rows, nz, R, T);
      FSG_RETURN_LAUNCH();
    }
    return FSG_E_ALIGN;  // caller falls back to fsg_blur_axis_f32
  }
  int outer, len, inner;
  if (axis == 0) { outer = 1; len = nx; inner = ny * nz; }
  else { outer = nx; len = ny; inner = nz; }
  if (aligned && (inner & 3) == 0 && R >= 1 && R <= 8) {
    switch (R) {
      case 1: return launch_strided<1>(src, dst, outer, len, inner, T, st);
      case 2: return launch_strided<2>(src, dst, outer, len, inner, T, st);
      case 3: return launch_strided<3>(src, dst, outer, len, inner, T, st);
      case 4: return launch_strided<4>(src, dst, outer, len, inner, T, st);
      case 5: return launch_strided<5>(src, dst, outer, len, inner, T, st);
      case 6: return launch_strided<6>(src, dst, outer, len, inner, T, st);
      case 7: return launch_strided<7>(src, dst, outer, len, inner, T, st);
      case 8: return launch_strided<8>(src, dst, outer, len, inner, T, st);
    }
  }
  if (aligned && (inner & 3) == 0 && R > 8) {
    constexpr int TL = 16;
    const int inner4 = inner / 4, chunks = (len + TL - 1) / TL;
    dim3 block(64, 4), grid((unsigned)((inner4 + 63) / 64), (unsigned)((chunks + 3) / 4), (unsigned)outer);
    hipLaunchKernelGGL(blur_strided_long<TL>, grid, block, 0, st, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), len, inner4, R, T);
    FSG_RETURN_LAUNCH();
  }
  return FSG_E_ALIGN;
}

// y pass followed by z pass in one launch (see blur_yz_fused_kernel); same result contract as two
// fsg_blur_axis_taps_host_f32 calls (axis 1 then axis 2).  Serves the isotropic case the generator produces (the same
// taps on both axes, radius 1..8), 16-byte aligned volumes, nz % 4 == 0; returns FSG_E_ALIGN otherwise (the caller then
// issues the two single-axis passes).
extern "C" int fsg_blur_yz_taps_host_f32(const float* src, float* dst, int nx, int ny, int nz, const float* taps_y_host,
                                         int ntaps_y, const float* taps_z_host, int ntaps_z, void* stream) {
  if (!src || !dst || src == dst || !taps_y_host || !taps_z_host) return FSG_E_BADARG;
  if (nx <= 0 || ny <= 0 || nz <= 0) return FSG_E_BADARG;
  if (ntaps_y <= 0 || (ntaps_y & 1) == 0 || ntaps_z <= 0 || (ntaps_z & 1) == 0) return FSG_E_BADARG;
  if ((size_t)nx * ny * nz > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  const int R = ntaps_y >> 1;
  const bool aligned = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  // one workgroup spans the whole z extent: 32 rows x (nz + halo) floats of LDS
  if (!aligned || (nz & 3) || ntaps_y != ntaps_z || R < 1 || R > YZ_MAXR || nx > 65535 || nz > 496) return FSG_E_ALIGN;
  for (int t = 0; t < ntaps_y; ++t)
    if (taps_y_host[t] != taps_z_host[t]) return FSG_E_ALIGN;
  Taps T;
  for (int t = 0; t < MAX_TAPS; ++t) T.w[t] = t < ntaps_y ? taps_y_host[t] : 0.f;
  hipStream_t st = fsg_stream(stream);
  switch (R) {
    case 1: return launch_yz<1>(src, dst, nx, ny, nz, T, st);
    case 2: return launch_yz<2>(src, dst, nx, ny, nz, T, st);
    case 3: return launch_yz<3>(src, dst, nx, ny, nz, T, st);
    case 4: return launch_yz<4>(src, dst, nx, ny, nz, T, st);
    case 5: return launch_yz<5>(src, dst, nx, ny, nz, T, st);
    case 6: return launch_yz<6>(src, dst, nx, ny, nz, T, st);
    case 7: return launch_yz<7>(src, dst, nx, ny, nz, T, st);
    default: return launch_yz<8>(src, dst, nx, ny, nz, T, st);
  }
}
