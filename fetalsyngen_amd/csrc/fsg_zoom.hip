// fsg_zoom.hip -- table-driven separable linear resampling (K2b stand-alone, K5b, K7+K8, K9, K10).
//
// Replaces `myzoom_torch` (utils/generation.py:310-397: three Python loops of slice-wise lerps, ~3k op
// dispatches per call) and the axis-aligned `fast_3D_interp_torch` call of RandResample
// (generator/augmentation/synthseg.py:87-104) with one launch; fused epilogues add the RandNoise draw
// (synthseg.py:230-233), the global-max normalisation of `resize_back` (synthseg.py:111-112) and the
// dataset's [0,1] scaling (data/datasets.py:311).
//
// The per-axis tables (index pair + weight pair per output sample) are built on the host with the very
// torch/numpy calls the reference uses, so sample positions are bit-identical; the kernel evaluates
// x, then y, then z as w_lo*a + w_hi*b without FMA.
#include "fsg_common.h"

namespace {

enum { EPI_STORE = 0, EPI_NOISE_PTR = 1, EPI_NOISE_PHILOX = 2, EPI_MINMAX = 3, EPI_NORM = 4 };

struct ZoomK {
  const float* src;
  int sx, sy, sz;
  const fsg_tap* tx;
  const fsg_tap* ty;
  const fsg_tap* tz;
  float* dst;
  int dx, dy, dz;
};

// generic multi-channel variant (API-level myzoom_torch; channel-last)
template <int NCH>
__global__ __launch_bounds__(256) void zoom_nch_kernel(ZoomK Z) {
  const int kc = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (kc >= Z.dz * NCH || j >= Z.dy) return;
  const int k = kc / NCH, ch = kc - k * NCH;
  const fsg_tap a = Z.tx[i], b = Z.ty[j], c = Z.tz[k];
  float v = 0.f;
  if (a.lo >= 0 && b.lo >= 0 && c.lo >= 0) v = fsg_tab_interp<NCH>(Z.src, Z.sy, Z.sz, ch, a, b, c);
  Z.dst[(((size_t)i * Z.dy + j) * Z.dz) * NCH + kc] = v;
}

struct EpiZ {
  const float* noise;
  uint64_t seed, stream_id;
  float noise_std;
  int32_t* mm_out;
  const int32_t* mm_in;
  int norm_mode;
};

template <int EPI>
__global__ __launch_bounds__(256) void zoom1_kernel(ZoomK Z, EpiZ E) {
  float lo = INFINITY, hi = -INFINITY;
  float inv_max = 0.f, mnq = 0.f, den = 1.f, mx = 1.f;
  if (EPI == EPI_NORM) {
    mx = fsg_key2f(E.mm_in[1]);
    const float mn = fsg_key2f(E.mm_in[0]);
    mnq = mn / mx;         // min(y/max) == min(y)/max: IEEE division is monotone
    den = 1.0f - mnq;      // max(y/max) == max/max == 1
    (void)inv_max;
  }
  const int rows = Z.dx * Z.dy;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int i = r / Z.dy, j = r - i * Z.dy;
    const fsg_tap a = Z.tx[i], b = Z.ty[j];
    const bool okr = a.lo >= 0 && b.lo >= 0;
    for (int k = threadIdx.x; k < Z.dz; k += blockDim.x) {
      const fsg_tap c = Z.tz[k];
      float v = 0.f;
      if (okr && c.lo >= 0) v = fsg_tab_interp<1>(Z.src, Z.sy, Z.sz, 0, a, b, c);
      const size_t o = (size_t)r * Z.dz + k;
      if (EPI == EPI_STORE) {
        Z.dst[o] = v;
      } else if (EPI == EPI_NOISE_PTR) {
        v = v + E.noise_std * E.noise[o];
        Z.dst[o] = v < 0.f ? 0.f : v;
      } else if (EPI == EPI_NOISE_PHILOX) {
        v = v + E.noise_std * fsg_randn1(E.seed, E.stream_id, (uint64_t)o);
        Z.dst[o] = v < 0.f ? 0.f : v;
      } else if (EPI == EPI_MINMAX) {
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
      } else {  // EPI_NORM
        float t = v / mx;
        if (E.norm_mode == 1) t = (mnq == 1.0f) ? t * 0.0f : (t - mnq) / den;  // flat image -> arr*minv
        Z.dst[o] = t;
      }
    }
  }
  if (EPI == EPI_MINMAX) {
    __shared__ float red[2][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lo = fsg_wave_min(lo);
    hi = fsg_wave_max(hi);
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
      fsg_atomic_min_key(&E.mm_out[0], lo);
      fsg_atomic_max_key(&E.mm_out[1], hi);
    }
  }
}

// ---- row-wise variant (tuned path) ------------------------------------------------------------------
// One wave per output row (i, j, :): the x- and y-interpolation of the four source rows it depends on
// is done once, coalesced, into a wave-private LDS row of sz floats; each output voxel is then one lerp
// of two LDS values.  Global traffic per output row: 4 coalesced source rows (L1/L2 hits for the
// neighbouring output rows that share them) + one coalesced store.  Same fp32 operation order as
// fsg_tab_interp<1>, so results are bit-identical to the per-voxel kernel.
constexpr int ZROWCAP = 1024;  // max source z extent handled by the row kernel

__device__ __forceinline__ void zwave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ fsg_tap zuniform_tap(const fsg_tap* t, int idx) {
  const int4 v = *reinterpret_cast<const int4*>(t + idx);
  fsg_tap r;
  r.lo = __builtin_amdgcn_readfirstlane(v.x);
  r.hi = __builtin_amdgcn_readfirstlane(v.y);
  r.w_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(v.z));
  r.w_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(v.w));
  return r;
}

template <int EPI>
__device__ __forceinline__ void zoom_emit(const ZoomK& Z, const EpiZ& E, const float* sm, bool okr, const fsg_tap& c,
                                          size_t o, float mx, float mnq, float den, float& lo, float& hi) {
  float v = 0.f;
  if (okr && c.lo >= 0) v = fsg_mix(c.w_lo, sm[c.lo], c.w_hi, sm[c.hi]);
  if (EPI == EPI_STORE) {
    Z.dst[o] = v;
  } else if (EPI == EPI_NOISE_PTR) {
    v = v + E.noise_std * E.noise[o];
    Z.dst[o] = v < 0.f ? 0.f : v;
  } else if (EPI == EPI_NOISE_PHILOX) {
    v = v + E.noise_std * fsg_randn1(E.seed, E.stream_id, (uint64_t)o);
    Z.dst[o] = v < 0.f ? 0.f : v;
  } else if (EPI == EPI_MINMAX) {
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  } else {
    float t = v / mx;
    if (E.norm_mode == 1) t = (mnq == 1.0f) ? t * 0.0f : (t - mnq) / den;
    Z.dst[o] = t;
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void zoom1_rows_kernel(ZoomK Z, EpiZ E, int rows_per_block) {
  __shared__ float sm_all[4][ZROWCAP];
  __shared__ float red[2][4];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* sm = sm_all[wave];
  float lo = INFINITY, hi = -INFINITY;
  float mnq = 0.f, den = 1.f, mx = 1.f;
  if (EPI == EPI_NORM) {
    mx = fsg_key2f(E.mm_in[1]);
    mnq = fsg_key2f(E.mm_in[0]) / mx;
    den = 1.0f - mnq;
  }
  const int nb = gridDim.x;
  const int tile = (nb & 7) == 0 ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int rows = Z.dx * Z.dy;
  const int r_begin = tile * rows_per_block;
  const int r_end = min(rows, r_begin + rows_per_block);
  const bool cached = Z.dz <= 256;
  fsg_tap ck[4];
  if (cached) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = lane + 64 * q;
      ck[q] = k < Z.dz ? Z.tz[k] : fsg_tap{-1, 0, 0.f, 0.f};
    }
  }
  for (int r = r_begin + wave; r < r_end; r += 4) {
    const int i = r / Z.dy, j = r - i * Z.dy;
    const fsg_tap a = zuniform_tap(Z.tx, i), b = zuniform_tap(Z.ty, j);
    const bool okr = a.lo >= 0 && b.lo >= 0;
    if (okr) {
      const float* p00 = Z.src + ((size_t)a.lo * Z.sy + b.lo) * Z.sz;
      const float* p10 = Z.src + ((size_t)a.hi * Z.sy + b.lo) * Z.sz;
      const float* p01 = Z.src + ((size_t)a.lo * Z.sy + b.hi) * Z.sz;
      const float* p11 = Z.src + ((size_t)a.hi * Z.sy + b.hi) * Z.sz;
      for (int zs = lane; zs < Z.sz; zs += FSG_WAVE) {
        const float t0 = fsg_mix(a.w_lo, p00[zs], a.w_hi, p10[zs]);
        const float t1 = fsg_mix(a.w_lo, p01[zs], a.w_hi, p11[zs]);
        sm[zs] = fsg_mix(b.w_lo, t0, b.w_hi, t1);
      }
    }
    zwave_sync();
    const size_t row = (size_t)r * Z.dz;
    if (cached) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        if (k < Z.dz) zoom_emit<EPI>(Z, E, sm, okr, ck[q], row + k, mx, mnq, den, lo, hi);
      }
    } else {
      for (int k = lane; k < Z.dz; k += FSG_WAVE)
        zoom_emit<EPI>(Z, E, sm, okr, Z.tz[k], row + k, mx, mnq, den, lo, hi);
    }
    zwave_sync();
  }
  if (EPI == EPI_MINMAX) {
    lo = fsg_wave_min(lo);
    hi = fsg_wave_max(hi);
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
      fsg_atomic_min_key(&E.mm_out[0], lo);
      fsg_atomic_max_key(&E.mm_out[1], hi);
    }
  }
}

// Software-pipelined variant for sz <= 256 and dz <= 256 (every low-res / full-res size of the 256^3
// configuration): the wave fetches the taps of ALL its rows with one coalesced load (lane t <- row t),
// and the four source rows of row t+1 are in flight in registers while row t is blended and emitted, so
// the dependent chain "table entry -> source rows -> LDS -> output" is paid once per wave, not per row.
template <int EPI>
__global__ __launch_bounds__(256, 8) void zoom1_rows_pf_kernel(ZoomK Z, EpiZ E, int rows_per_block) {
  __shared__ float sm_all[4][2][256];
  __shared__ float red[2][4];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float lo = INFINITY, hi = -INFINITY;
  float mnq = 0.f, den = 1.f, mx = 1.f;
  if (EPI == EPI_NORM) {
    mx = fsg_key2f(E.mm_in[1]);
    mnq = fsg_key2f(E.mm_in[0]) / mx;
    den = 1.0f - mnq;
  }
  const int nb = gridDim.x;
  const int tile = (nb & 7) == 0 ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int rows = Z.dx * Z.dy;
  const int r_begin = tile * rows_per_block;
  const int r_end = min(rows, r_begin + rows_per_block);
  fsg_tap ck[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = lane + 64 * q;
    ck[q] = k < Z.dz ? Z.tz[k] : fsg_tap{-1, 0, 0.f, 0.f};
  }
  for (int base = r_begin + wave; base < r_end; base += 4 * 64) {  // chunks of 64 rows of this wave
    const int nw = min(64, (r_end - base + 3) / 4);
    int4 ta = make_int4(-1, 0, 0, 0), tb = make_int4(-1, 0, 0, 0);
    if (lane < nw) {
      const int r = base + 4 * lane;
      const int i = r / Z.dy, j = r - i * Z.dy;
      ta = *reinterpret_cast<const int4*>(Z.tx + i);
      tb = *reinterpret_cast<const int4*>(Z.ty + j);
    }
    float cur[4][4], nxt[4][4];
    fsg_tap a, b, an, bn;
    auto taps_of = [&](int t, fsg_tap& ra, fsg_tap& rb) {
      ra.lo = __builtin_amdgcn_readlane(ta.x, t);
      ra.hi = __builtin_amdgcn_readlane(ta.y, t);
      ra.w_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ta.z, t));
      ra.w_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ta.w, t));
      rb.lo = __builtin_amdgcn_readlane(tb.x, t);
      rb.hi = __builtin_amdgcn_readlane(tb.y, t);
      rb.w_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tb.z, t));
      rb.w_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tb.w, t));
    };
    auto fetch = [&](const fsg_tap& ra, const fsg_tap& rb, float (&v)[4][4]) {
      if (ra.lo < 0 || rb.lo < 0) return;
      const float* p[4] = {Z.src + ((size_t)ra.lo * Z.sy + rb.lo) * Z.sz, Z.src + ((size_t)ra.hi * Z.sy + rb.lo) * Z.sz,
                           Z.src + ((size_t)ra.lo * Z.sy + rb.hi) * Z.sz, Z.src + ((size_t)ra.hi * Z.sy + rb.hi) * Z.sz};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int zs = lane + 64 * c;
        if (zs < Z.sz) {
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u][c] = p[u][zs];
        }
      }
    };
    taps_of(0, a, b);
    fetch(a, b, cur);
    for (int t = 0; t < nw; ++t) {
      if (t + 1 < nw) {
        taps_of(t + 1, an, bn);
        fetch(an, bn, nxt);
      }
      float* sm = sm_all[wave][t & 1];
      const bool okr = a.lo >= 0 && b.lo >= 0;
      if (okr) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int zs = lane + 64 * c;
          if (zs < Z.sz) {
            const float t0 = fsg_mix(a.w_lo, cur[0][c], a.w_hi, cur[1][c]);
            const float t1 = fsg_mix(a.w_lo, cur[2][c], a.w_hi, cur[3][c]);
            sm[zs] = fsg_mix(b.w_lo, t0, b.w_hi, t1);
          }
        }
      }
      zwave_sync();
      const size_t row = (size_t)(base + 4 * t) * Z.dz;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        if (k < Z.dz) zoom_emit<EPI>(Z, E, sm, okr, ck[q], row + k, mx, mnq, den, lo, hi);
      }
      a = an;
      b = bn;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[u][c] = nxt[u][c];
    }
    zwave_sync();
  }
  if (EPI == EPI_MINMAX) {
    lo = fsg_wave_min(lo);
    hi = fsg_wave_max(hi);
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
      fsg_atomic_min_key(&E.mm_out[0], lo);
      fsg_atomic_max_key(&E.mm_out[1], hi);
    }
  }
}

int check(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty, const fsg_tap* tz, int dx,
          int dy, int dz) {
  if (!src || !tx || !ty || !tz) return FSG_E_BADARG;
  if (sx <= 0 || sy <= 0 || sz <= 0 || dx <= 0 || dy <= 0 || dz <= 0) return FSG_E_BADARG;
  if ((size_t)sx * sy * sz > (size_t)0x7FFFFFFF / 4 || (size_t)dx * dy * dz > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  return 0;
}

template <int EPI>
int launch1(const ZoomK& Z, const EpiZ& E, void* stream) {
  const int rows = Z.dx * Z.dy;
  if (Z.sz <= ZROWCAP && !(g_tuning_flags & FSG_TUNE_GENERIC_ZOOM)) {
    // 2048 blocks when there is enough work: >= 8 rows per block keeps neighbouring rows (shared source
    // rows) on one CU, and the MINMAX variant issues only 2 atomics per block
    int nblk = (rows + 7) / 8;
    if (nblk > 2048) nblk = 2048;
    if (nblk >= 8) nblk &= ~7;  // multiple of 8 for the XCD-contiguous tile order
    const int rpb = (rows + nblk - 1) / nblk;
    if (Z.sz <= 256 && Z.dz <= 256 && !(g_tuning_flags & FSG_TUNE_NO_PREFETCH))
      hipLaunchKernelGGL(zoom1_rows_pf_kernel<EPI>, dim3(nblk), dim3(256), 0, fsg_stream(stream), Z, E, rpb);
    else
      hipLaunchKernelGGL(zoom1_rows_kernel<EPI>, dim3(nblk), dim3(256), 0, fsg_stream(stream), Z, E, rpb);
    FSG_RETURN_LAUNCH();
  }
  const int grid = rows < 4096 ? rows : 4096;
  hipLaunchKernelGGL(zoom1_kernel<EPI>, dim3(grid), dim3(256), 0, fsg_stream(stream), Z, E);
  FSG_RETURN_LAUNCH();
}

}  // namespace

extern "C" {

int fsg_zoom3d_f32(const float* src, int sx, int sy, int sz, int nch, const fsg_tap* tx, const fsg_tap* ty,
                   const fsg_tap* tz, float* dst, int dx, int dy, int dz, void* stream) {
  int rc = check(src, sx, sy, sz, tx, ty, tz, dx, dy, dz);
  if (rc) return rc;
  if (!dst || src == dst) return FSG_E_BADARG;
  ZoomK Z{src, sx, sy, sz, tx, ty, tz, dst, dx, dy, dz};
  if (nch == 1) {
    EpiZ E{};
    return launch1<EPI_STORE>(Z, E, stream);
  }
  if (nch != 3) return FSG_E_BADARG;
  dim3 grid((unsigned)((dz * 3 + 63) / 64), (unsigned)((dy + 3) / 4), (unsigned)dx);
  hipLaunchKernelGGL(zoom_nch_kernel<3>, grid, fsg_block3(), 0, fsg_stream(stream), Z);
  FSG_RETURN_LAUNCH();
}

int fsg_resample_noise_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                           const fsg_tap* tz, float* dst, int dx, int dy, int dz, int noise_mode,
                           const float* noise, uint64_t seed, uint64_t stream_id, float noise_std, void* stream) {
  int rc = check(src, sx, sy, sz, tx, ty, tz, dx, dy, dz);
  if (rc) return rc;
  if (!dst || src == dst) return FSG_E_BADARG;
  ZoomK Z{src, sx, sy, sz, tx, ty, tz, dst, dx, dy, dz};
  EpiZ E{};
  E.noise = noise; E.seed = seed; E.stream_id = stream_id; E.noise_std = noise_std;
  switch (noise_mode) {
    case 0: return launch1<EPI_STORE>(Z, E, stream);
    case 1: if (!noise) return FSG_E_BADARG; return launch1<EPI_NOISE_PTR>(Z, E, stream);
    case 2: return launch1<EPI_NOISE_PHILOX>(Z, E, stream);
    default: return FSG_E_BADARG;
  }
}

int fsg_zoom3d_minmax_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                          const fsg_tap* tz, int dx, int dy, int dz, int32_t* mm, void* stream) {
  int rc = check(src, sx, sy, sz, tx, ty, tz, dx, dy, dz);
  if (rc) return rc;
  if (!mm) return FSG_E_BADARG;
  ZoomK Z{src, sx, sy, sz, tx, ty, tz, nullptr, dx, dy, dz};
  EpiZ E{};
  E.mm_out = mm;
  return launch1<EPI_MINMAX>(Z, E, stream);
}

int fsg_zoom3d_normalise_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                             const fsg_tap* tz, float* dst, int dx, int dy, int dz, const int32_t* mm, int mode,
                             void* stream) {
  int rc = check(src, sx, sy, sz, tx, ty, tz, dx, dy, dz);
  if (rc) return rc;
  if (!dst || !mm || src == dst || (mode != 0 && mode != 1)) return FSG_E_BADARG;
  ZoomK Z{src, sx, sy, sz, tx, ty, tz, dst, dx, dy, dz};
  EpiZ E{};
  E.mm_in = mm; E.norm_mode = mode;
  return launch1<EPI_NORM>(Z, E, stream);
}

}  // extern "C"
