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
#include <cstdlib>

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
  int mm_shards;  // 0 / 1: mm = {min, max}; S > 1: S slots of FSG_MM_SLOT_STRIDE ints, slot s = {min, max, ...} (fsg_hip.h)
};

// min / max keys of a (possibly sharded) pair.  Sharded: the two keys of a min/max pass are ONE address each, and a few
// thousand workgroups ending with a gated atomic on them cost 15-25 us (profiles/r02_b_zoom_experiments.txt); with S slots
// on separate 64-byte lines a workgroup updates slot blockIdx % S and the readers reduce the S slots (<= 64: one wave load).
__device__ __forceinline__ void zoom_mm_update(const EpiZ& E, float lo, float hi) {
  if (E.mm_shards > 1) {
    // sharded: two atomics WITHOUT a returned value and without the load that gates the single-pair form -- nothing the
    // workgroup has to wait for, it retires as soon as they are issued; S slots keep the per-address rate low
    int32_t* s = E.mm_out + (int)(blockIdx.x % (unsigned)E.mm_shards) * FSG_MM_SLOT_STRIDE;
    (void)__hip_atomic_fetch_min(&s[0], fsg_f2key(lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_max(&s[1], fsg_f2key(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  fsg_atomic_min_key(&E.mm_out[0], lo);
  fsg_atomic_max_key(&E.mm_out[1], hi);
}
__device__ __forceinline__ void zoom_mm_read(const EpiZ& E, float& mn, float& mx) {
  if (E.mm_shards > 1) {
    const int lane = threadIdx.x & 63;
    int kmin = 0x7FFFFFFF, kmax = (int)0x80000000;
    if (lane < E.mm_shards) {
      kmin = E.mm_in[lane * FSG_MM_SLOT_STRIDE];
      kmax = E.mm_in[lane * FSG_MM_SLOT_STRIDE + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      kmin = min(kmin, __shfl_xor(kmin, o, FSG_WAVE));
      kmax = max(kmax, __shfl_xor(kmax, o, FSG_WAVE));
    }
    mn = fsg_key2f(__builtin_amdgcn_readfirstlane(kmin));
    mx = fsg_key2f(__builtin_amdgcn_readfirstlane(kmax));
  } else {
    mn = fsg_key2f(E.mm_in[0]);
    mx = fsg_key2f(E.mm_in[1]);
  }
}

// the same in two halves, so that the loads are in flight while the caller does other work (slab kernel: behind its x stage)
__device__ __forceinline__ void zoom_mm_issue(const EpiZ& E, int& kmin, int& kmax) {
  const int lane = threadIdx.x & 63;
  kmin = 0x7FFFFFFF;
  kmax = (int)0x80000000;
  if (E.mm_shards > 1) {
    if (lane < E.mm_shards) {
      kmin = E.mm_in[lane * FSG_MM_SLOT_STRIDE];
      kmax = E.mm_in[lane * FSG_MM_SLOT_STRIDE + 1];
    }
  } else {
    kmin = E.mm_in[0];
    kmax = E.mm_in[1];
  }
}
__device__ __forceinline__ void zoom_mm_finish(int kmin, int kmax, float& mn, float& mx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, __shfl_xor(kmin, o, FSG_WAVE));
    kmax = max(kmax, __shfl_xor(kmax, o, FSG_WAVE));
  }
  mn = fsg_key2f(__builtin_amdgcn_readfirstlane(kmin));
  mx = fsg_key2f(__builtin_amdgcn_readfirstlane(kmax));
}

template <int EPI>
__global__ __launch_bounds__(256) void zoom1_kernel(ZoomK Z, EpiZ E) {
  float lo = INFINITY, hi = -INFINITY;
  float inv_max = 0.f, mnq = 0.f, den = 1.f, mx = 1.f;
  if (EPI == EPI_NORM) {
    float mn;
    zoom_mm_read(E, mn, mx);
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
      zoom_mm_update(E, lo, hi);
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
    // den == 1 (min == 0: K8 clamps at 0, so this is the usual case): x / 1.0f == x exactly, the division is skipped
    if (E.norm_mode == 1) t = (mnq == 1.0f) ? t * 0.0f : (den == 1.0f ? t - mnq : (t - mnq) / den);
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
    float mn;
    zoom_mm_read(E, mn, mx);
    mnq = mn / mx;
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
      zoom_mm_update(E, lo, hi);
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
    float mn;
    zoom_mm_read(E, mn, mx);
    mnq = mn / mx;
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
      zoom_mm_update(E, lo, hi);
    }
  }
}


// ---- tile variant ---------------------------------------------------------------------------------------
// SQ counters of the row kernels above at 256^3 (rocprofv3 --pmc, tools/kernel_bench.py --only zoom): ~92 VALU + ~108
// SALU + 13 VMEM + 13 LDS wave-instructions per output row and a time that is linear in the number of rows and blind to
// the number of cache lines fetched -- they are bound by instruction issue (one row per wave iteration: tap unpacking,
// pointer arithmetic, LDS round trip and, with the Philox epilogue, one 10-round Philox block PER OUTPUT although a block
// yields four normals).  This kernel spends its instructions on outputs instead:
//   * workgroup = one x index and TY consecutive output y rows; the x-blended source rows those outputs can touch
//     (a window of sy/dy * TY + 2 rows) are formed ONCE, coalesced, into LDS;
//   * a thread then produces four consecutive outputs of the tile's contiguous output range: two y-blends of LDS
//     values and one z-lerp per output (x -> y -> z, the operation order of fsg_tab_interp<1>: bit-identical), one
//     Philox block per four outputs, one 16-byte store.
// Falls back to the row kernels when the window does not fit the LDS budget.
constexpr int ZT_MAX_TY = 64;

template <int EPI>
__global__ __launch_bounds__(256) void zoom_tile_kernel(ZoomK Z, EpiZ E, int TY, int cap_floats) {
  extern __shared__ __attribute__((aligned(16))) float zt_smem[];
  fsg_tap* tc = reinterpret_cast<fsg_tap*>(zt_smem);  // [dz] z taps
  float* xs = zt_smem + 4 * Z.dz;                      // [window rows][sz] x-blended source rows
  __shared__ fsg_tap tb[ZT_MAX_TY];
  __shared__ int win[2];
  __shared__ float red[2][4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_y = (Z.dy + TY - 1) / TY;
  const int i = blockIdx.x / tiles_y, jt = blockIdx.x - i * tiles_y;
  const int j0 = jt * TY, nj = min(TY, Z.dy - j0);
  float lo = INFINITY, hi = -INFINITY;
  float mnq = 0.f, den = 1.f, mx = 1.f;
  if (EPI == EPI_NORM) {
    float mn;
    zoom_mm_read(E, mn, mx);
    mnq = mn / mx;
    den = 1.0f - mnq;
  }
  // y taps of the tile and the window of source rows they reference
  if (tid < 64) {
    fsg_tap t = fsg_tap{-1, 0, 0.f, 0.f};
    if (tid < nj) {
      t = Z.ty[j0 + tid];
      tb[tid] = t;
    }
    int smin = t.lo >= 0 ? t.lo : 0x7FFFFFFF, smax = t.lo >= 0 ? t.hi : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      smin = min(smin, __shfl_xor(smin, o, FSG_WAVE));
      smax = max(smax, __shfl_xor(smax, o, FSG_WAVE));
    }
    if (tid == 0) { win[0] = smin; win[1] = smax; }
  }
  __syncthreads();
  const fsg_tap a = zuniform_tap(Z.tx, i);
  const int smin = win[0], nrows = win[1] - win[0] + 1;
  const bool okx = a.lo >= 0 && nrows > 0;
  const bool fits = nrows * Z.sz <= cap_floats;  // uniform
  for (int k = tid; k < Z.dz; k += 256) tc[k] = Z.tz[k];
  if (okx && fits) {
    const float* pa = Z.src + ((size_t)a.lo * Z.sy + smin) * Z.sz;
    const float* pb = Z.src + ((size_t)a.hi * Z.sy + smin) * Z.sz;
    const int tot = nrows * Z.sz;  // the window rows are contiguous in the source: one linear, coalesced sweep
    for (int e = tid; e < tot; e += 256) xs[e] = fsg_mix(a.w_lo, pa[e], a.w_hi, pb[e]);
  }
  __syncthreads();
  const size_t base = ((size_t)i * Z.dy + j0) * Z.dz;  // first output of the tile
  const size_t end = base + (size_t)nj * Z.dz;
  const float inv_dz = 1.0f / (float)Z.dz;
  for (size_t q = (base >> 2) + tid; (q << 2) < end; q += 256) {
    const size_t o0 = q << 2;
    float v[4];
    bool live[4];
    // (row, k) of the first live element of the quad
    const size_t first = o0 < base ? base : o0;
    int rel = (int)(first - base);
    int jj = (int)((float)rel * inv_dz);
    int k = rel - jj * Z.dz;
    if (k < 0) { --jj; k += Z.dz; }
    if (k >= Z.dz) { ++jj; k -= Z.dz; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = o0 + u;
      live[u] = o >= base && o < end;
      v[u] = 0.f;
      if (live[u]) {
        const fsg_tap b = tb[jj];
        const fsg_tap c = tc[k];
        if (okx && b.lo >= 0 && c.lo >= 0) {
          if (fits) {
            const float* xl = xs + (b.lo - smin) * Z.sz;
            const float* xh = xs + (b.hi - smin) * Z.sz;
            const float ylo = fsg_mix(b.w_lo, xl[c.lo], b.w_hi, xh[c.lo]);
            const float yhi = fsg_mix(b.w_lo, xl[c.hi], b.w_hi, xh[c.hi]);
            v[u] = fsg_mix(c.w_lo, ylo, c.w_hi, yhi);
          } else {  // window larger than the launch reserved (tables that are not a plain zoom)
            v[u] = fsg_tab_interp<1>(Z.src, Z.sy, Z.sz, 0, a, b, c);
          }
        }
        if (++k == Z.dz) { k = 0; ++jj; }
      }
    }
    if (EPI == EPI_NOISE_PHILOX) {
      const float4 z = fsg_randn4(E.seed, E.stream_id, (uint64_t)q);
      v[0] += E.noise_std * z.x; v[1] += E.noise_std * z.y; v[2] += E.noise_std * z.z; v[3] += E.noise_std * z.w;
    } else if (EPI == EPI_NOISE_PTR) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live[u]) v[u] += E.noise_std * E.noise[o0 + u];
    }
    if (EPI == EPI_NOISE_PHILOX || EPI == EPI_NOISE_PTR) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = v[u] < 0.f ? 0.f : v[u];
    } else if (EPI == EPI_NORM) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float t = v[u] / mx;
        if (E.norm_mode == 1) t = (mnq == 1.0f) ? t * 0.0f : (den == 1.0f ? t - mnq : (t - mnq) / den);
        v[u] = t;
      }
    }
    if (EPI == EPI_MINMAX) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live[u]) { lo = fminf(lo, v[u]); hi = fmaxf(hi, v[u]); }
    } else if (live[0] && live[3] && ((((uintptr_t)Z.dst) & 15) == 0)) {
      *reinterpret_cast<float4*>(Z.dst + o0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live[u]) Z.dst[o0 + u] = v[u];
    }
  }
  if (EPI == EPI_MINMAX) {
    lo = fsg_wave_min(lo);
    hi = fsg_wave_max(hi);
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
      zoom_mm_update(E, lo, hi);
    }
  }
}

// Correctly rounded v / d for a divisor that is uniform over the launch (K9b divides 16.8 M voxels by the same maximum).
// r = RN(1/d) once; per value q = RN(v r), e = v - q d (exact in one FMA for a faithful q), result RN(q + e r).  With a
// correctly rounded reciprocal this is the correctly rounded quotient (Markstein's theorem) except when d's significand is all
// ones; outside a safe exponent window for d or q the IEEE division runs instead (exec-masked, empty in practice).  ~4 VALU
// instead of the ~10 of the division expansion; bit-identical to `v / d` (tests/test_hip_parity.py::test_uniform_division...).
// per-output evaluation for the slab kernel's rare "window does not fit" path; out of line so that its registers do not
// count against the kernel's hot loop
__device__ __attribute__((noinline)) float zoom_slab_slow(const float* src, int sy, int sz, fsg_tap a, fsg_tap b, fsg_tap c) {
  return fsg_tab_interp<1>(src, sy, sz, 0, a, b, c);
}

struct UniDiv {
  float d, r;
  bool fast;
};
__device__ __forceinline__ UniDiv unidiv_make(float d) {
  UniDiv u;
  u.d = d;
  u.r = 1.0f / d;
  const unsigned bits = __builtin_bit_cast(unsigned, d);
  u.fast = d > 1e-18f && d < 1e18f && (bits & 0x7FFFFFu) != 0x7FFFFFu;
  return u;
}
__device__ __forceinline__ float unidiv(const UniDiv& u, float v) {
  const float q = v * u.r;
  const float e = __builtin_fmaf(-q, u.d, v);
  const float q1 = __builtin_fmaf(e, u.r, q);
  const bool ok = u.fast && ((q > 1e-18f && q < 1e18f) || v == 0.f);
  return ok ? q1 : v / u.d;
}

// ---- slab variant (default for the passes without a noise draw: K9a min/max, K9b normalise, plain zoom) --------
// The row kernels above are bound by a chain of dependent global round trips per output row (taps -> four source rows
// -> LDS -> outputs, ~3.6 us per row and wave: profiles/r01_r_pmc_bench_kernels.json), the tile kernel by four LDS reads
// and three blends per OUTPUT.  Here the three separable stages each run once per element they produce and only the
// first touches global memory:
//   1. x: as the tile kernel -- the window of source rows the tile's TY output rows reference, x-blended, coalesced,
//      into LDS (one phase of global loads per workgroup);
//   2. y: a wave blends two window rows into its private LDS row (sz elements, 2 LDS reads + 1 blend each);
//   3. z: every lane emits four consecutive outputs of the row (2 LDS reads + 1 lerp each; the z taps of the lane's
//      four outputs stay in registers when dz <= 256), one 16-byte store.
// Blend order x -> y -> z with separate multiplies and adds: bit-identical to fsg_tab_interp<1> and to the other kernels.
template <int EPI>
#ifndef FSG_SLAB_WAVES
#define FSG_SLAB_WAVES 1  // minimum waves per SIMD asked of the register allocator (tuning: tools/ab builds)
#endif
__device__ __forceinline__ void zoom_slab_body(const ZoomK& Z, const EpiZ& E, const int TY, const int cap_floats, const unsigned bid,
                                               const unsigned nbk) {
  // domain (checked by the launcher): sz <= 256 and dz <= 256 -- a lane owns source elements lane + 64 c (c < 4) in the
  // y stage and outputs 4 lane .. 4 lane + 3 in the z stage, everything unrolled
  extern __shared__ __attribute__((aligned(16))) float zt_smem[];
  float* yr = zt_smem;              // [4 waves][256] y-blended row of each wave (padded to the domain's 256: the y stage writes
                                    // all four of a lane's columns unconditionally, the z stage never reads beyond sz)
  float* xs = yr + 4 * 256;         // [window rows][sz] x-blended source rows
  __shared__ fsg_tap tb[ZT_MAX_TY];
  __shared__ int win[2];
  __shared__ float red[2][4];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int tiles_y = (Z.dy + TY - 1) / TY;
  const int nb = (int)nbk;
  const int tile = (nb & 7) == 0 ? (int)(bid & 7) * (nb >> 3) + (int)(bid >> 3) : (int)bid;  // XCD-contiguous x slabs
  const int i = tile / tiles_y, jt = tile - i * tiles_y;
  const int j0 = jt * TY, nj = min(TY, Z.dy - j0);
  float lo = INFINITY, hi = -INFINITY;
  float mnq = 0.f, den = 1.f, mx = 1.f;
  int kmin_l = 0, kmax_l = 0;
  if (EPI == EPI_NORM) zoom_mm_issue(E, kmin_l, kmax_l);  // consumed behind the x stage: the round trip hides under its loads
  // the z taps of this lane's four outputs: the same for every row of every tile
  int zlo[4], zhi[4];
  float zwl[4], zwh[4];
  bool zok[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = min(lane * 4 + u, Z.dz - 1);
    const fsg_tap c = Z.tz[k];
    zok[u] = c.lo >= 0 && lane * 4 + u < Z.dz;
    zlo[u] = zok[u] ? c.lo : 0;
    zhi[u] = zok[u] ? c.hi : 0;
    zwl[u] = c.w_lo;
    zwh[u] = c.w_hi;
  }
  const bool dst16 = (Z.dz & 3) == 0 && ((((uintptr_t)Z.dst) & 15) == 0);
  const bool full = lane * 4 + 3 < Z.dz;
  const bool row4 = (Z.dz & 3) == 0;
  float* y = yr + wave * 256;
  if (tid < 64) {
    fsg_tap t = fsg_tap{-1, 0, 0.f, 0.f};
    if (tid < nj) {
      t = Z.ty[j0 + tid];
      tb[tid] = t;
    }
    int smin = t.lo >= 0 ? t.lo : 0x7FFFFFFF, smax = t.lo >= 0 ? t.hi : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      smin = min(smin, __shfl_xor(smin, o, FSG_WAVE));
      smax = max(smax, __shfl_xor(smax, o, FSG_WAVE));
    }
    if (tid == 0) { win[0] = smin; win[1] = smax; }
  }
  const fsg_tap a = zuniform_tap(Z.tx, i);
  __syncthreads();
  const int smin = win[0], nrows = win[1] - win[0] + 1;
  const bool okx = a.lo >= 0 && nrows > 0;
  const bool fits = nrows * Z.sz <= cap_floats;  // uniform; false only for tables that are not a plain zoom
  if (okx && fits) {
    const float* pa = Z.src + ((size_t)a.lo * Z.sy + smin) * Z.sz;
    const float* pb = Z.src + ((size_t)a.hi * Z.sy + smin) * Z.sz;
    const int tot = nrows * Z.sz;  // the window rows are contiguous in the source: one linear, coalesced sweep
    for (int e = tid; e < tot; e += 256) xs[e] = fsg_mix(a.w_lo, pa[e], a.w_hi, pb[e]);
  }
  if (EPI == EPI_NORM) {
    float mn;
    zoom_mm_finish(kmin_l, kmax_l, mn, mx);
    mnq = mn / mx;
    den = 1.0f - mnq;
  }
  const UniDiv ud = unidiv_make(mx);
  const UniDiv udd = unidiv_make(den);  // the scaling's second divisor (1 - min/max) is uniform as well
  // which arithmetic follows the first quotient (uniform): 0 none, 1 "* 0" (min == max), 2 "- min/max" (den == 1), 3 "(- min/max) / den"
  // (min == 0, the usual case: "- 0" leaves every float as it is -> 0)
  const int nmode = (EPI == EPI_NORM && E.norm_mode == 1) ? (mnq == 1.0f ? 1 : (den == 1.0f ? (mnq == 0.0f ? 0 : 2) : 3)) : 0;
#ifdef FSG_NO_UNIDIV
  const bool uni_ok = false;
#else
  const bool uni_ok = ud.fast && (nmode != 3 || udd.fast);
#endif
  __syncthreads();
  auto row_tap = [&](int jj) {
    fsg_tap b = tb[jj];
    b.lo = __builtin_amdgcn_readfirstlane(b.lo);
    b.hi = __builtin_amdgcn_readfirstlane(b.hi);
    b.w_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.w_lo)));
    b.w_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.w_hi)));
    return b;
  };
  auto emit = [&](int jj, float (&v)[4]) {
    const size_t o0 = ((size_t)i * Z.dy + j0 + jj) * Z.dz + (size_t)lane * 4;
    if (EPI == EPI_NOISE_PHILOX) {
      if (full && (o0 & 3) == 0) {
        const float4 z = fsg_randn4(E.seed, E.stream_id, (uint64_t)(o0 >> 2));
        v[0] += E.noise_std * z.x; v[1] += E.noise_std * z.y; v[2] += E.noise_std * z.z; v[3] += E.noise_std * z.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (lane * 4 + u < Z.dz) v[u] += E.noise_std * fsg_randn1(E.seed, E.stream_id, (uint64_t)(o0 + u));
      }
    } else if (EPI == EPI_NOISE_PTR) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane * 4 + u < Z.dz) v[u] += E.noise_std * E.noise[o0 + u];
    }
    if (EPI == EPI_NOISE_PHILOX || EPI == EPI_NOISE_PTR) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = v[u] < 0.f ? 0.f : v[u];
    } else if (EPI == EPI_NORM) {
      // The four quotients straight-line, ONE uniform test per row for "some lane left the fast division's domain" (then
      // the row is redone by IEEE division: the same values, unidiv's contract).  Per element `ok ? q1 : v / d` compiled
      // to a branch around an IEEE division per element and quotient: ~100 branch instructions per row in this loop.
      float t[4];
      bool bad = !uni_ok;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float q = v[u] * ud.r;
        const float e = __builtin_fmaf(-q, ud.d, v[u]);
        t[u] = __builtin_fmaf(e, ud.r, q);
        bad |= !((q > 1e-18f && q < 1e18f) || v[u] == 0.f);
      }
      if (nmode == 3) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float w = t[u] - mnq;
          const float q = w * udd.r;
          const float e = __builtin_fmaf(-q, udd.d, w);
          t[u] = __builtin_fmaf(e, udd.r, q);
          bad |= !((q > 1e-18f && q < 1e18f) || w == 0.f);
        }
      } else if (nmode == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = t[u] - mnq;
      } else if (nmode == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = t[u] * 0.0f;
      }
      if (__builtin_expect(__ballot(bad) != 0ull, 0)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float tt = v[u] / mx;
          if (E.norm_mode == 1) tt = (mnq == 1.0f) ? tt * 0.0f : (den == 1.0f ? tt - mnq : (tt - mnq) / den);
          t[u] = tt;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = t[u];
    }
    if (EPI == EPI_MINMAX) {
      if (row4) {  // uniform: dz % 4 == 0, a lane's four outputs are all inside the row or all outside -- min3 / max3 tree,
                   // then one select against the identity (element by element this was 7 vector instructions per output:
                   // fminf / fmaxf re-canonicalise both operands every time)
        const float m0 = fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), m1 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        lo = fminf(lo, full ? m0 : INFINITY);
        hi = fmaxf(hi, full ? m1 : -INFINITY);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (lane * 4 + u < Z.dz) { lo = fminf(lo, v[u]); hi = fmaxf(hi, v[u]); }
      }
    } else if (dst16) {  // uniform (includes dz % 4 == 0: a lane's four outputs are all inside the row or all outside)
      if (full) *reinterpret_cast<float4*>(Z.dst + o0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane * 4 + u < Z.dz) Z.dst[o0 + u] = v[u];
    }
  };
  if (fits || !okx) {
    for (int jj = wave; jj < nj; jj += 4) {
      const fsg_tap b = row_tap(jj);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (okx && b.lo >= 0) {
        const float* xl = xs + (b.lo - smin) * Z.sz;
        const float* xh = xs + (b.hi - smin) * Z.sz;
        float l4[4], h4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // all eight reads in flight before the first blend.  Columns >= sz read on into the next
          l4[c] = xl[lane + 64 * c];   // window row or the 256 floats reserved behind the window: whatever they hold only reaches
          h4[c] = xh[lane + 64 * c];   // the padding of y, which nobody reads -- one address per source row, the column an immediate
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) y[lane + 64 * c] = fsg_mix(b.w_lo, l4[c], b.w_hi, h4[c]);  // (columns >= sz: padding, never read)
        zwave_sync();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float t = fsg_mix(zwl[u], y[zlo[u]], zwh[u], y[zhi[u]]);
          v[u] = zok[u] ? t : 0.f;
        }
        zwave_sync();  // the wave's row is rewritten by its next iteration
      }
      emit(jj, v);
    }
  } else {  // window larger than the launch reserved (tables that are not a plain zoom): per-output evaluation
    for (int jj = wave; jj < nj; jj += 4) {
      const fsg_tap b = row_tap(jj);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (b.lo >= 0) {
        for (int u = 0; u < 4; ++u)
          if (zok[u]) v[u] = zoom_slab_slow(Z.src, Z.sy, Z.sz, a, b, fsg_tap{zlo[u], zhi[u], zwl[u], zwh[u]});
      }
      emit(jj, v);
    }
  }
  if (EPI == EPI_MINMAX) {
    lo = fsg_wave_min(lo);
    hi = fsg_wave_max(hi);
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
#ifdef FSG_DIAG
    if (tid == 0 && E.norm_mode != 77) {
#else
    if (tid == 0) {
#endif
      for (int w = 1; w < 4; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
      zoom_mm_update(E, lo, hi);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, FSG_SLAB_WAVES) void zoom_slab_kernel(ZoomK Z, EpiZ E, int TY, int cap_floats) {
  zoom_slab_body<EPI>(Z, E, TY, cap_floats, blockIdx.x, gridDim.x);
}



// ---- wave variant (r03; opt-in, FSG_TUNE_WAVE_ZOOM: bit-identical, slower than the slab kernel -- see launch1) -----------
// Four INDEPENDENT waves per workgroup, no barrier: a wave owns ZW_ROWS consecutive output rows of one x-plane.  It forms the
// window of x-blended source rows those outputs reference in its own LDS rows (one phase of global loads: the window is a
// contiguous run of the two source planes), then every output is evaluated straight from that window -- two y-blends of LDS
// values and one z-lerp, exactly the tile kernel's arithmetic (x -> y -> z, separate multiplies and adds: bit-identical to
// fsg_tab_interp<1>) -- with the z taps of the lane's four outputs in registers.  Against the slab kernel: no y-blended row
// is written to LDS and read back (a dependent write -> sync -> read per output row), no workgroup-wide phases (the slab
// kernel's fixed 10-12 us per pass: taps -> barrier -> window -> barrier, profiles/r02_b_zoom_experiments.txt), and a wave
// that waits only ever waits for its own loads.  Same lesson as csrc/fsg_blur_rs.hip's y,z kernel.
constexpr int ZW_ROWS = 8;   // output rows per wave
constexpr int ZW_B = 12;    // trips of the window sweep whose loads are in flight together
constexpr int ZW_XR = 12;    // window rows a wave can hold (ZW_ROWS outputs of an up-sampling table reference <= ZW_ROWS + 1)

template <int EPI>
__global__ __launch_bounds__(256) void zoom_wave_kernel(ZoomK Z, EpiZ E) {
  extern __shared__ __attribute__((aligned(16))) float zw_smem[];
  const int tx = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* xs = zw_smem + (size_t)wave * ZW_XR * Z.sz;  // this wave's window: [ZW_XR][sz]
  const int tiles_y = (Z.dy + 4 * ZW_ROWS - 1) / (4 * ZW_ROWS);
  const int nb = gridDim.x;
  const int tile = (nb & 7) == 0 ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;  // XCD-contiguous x slabs
  const int i = tile / tiles_y;
  const int j0 = (tile - i * tiles_y) * 4 * ZW_ROWS + wave * ZW_ROWS;
  if (j0 >= Z.dy) return;  // whole wave; there is no barrier in this kernel
  const int nj = min(ZW_ROWS, Z.dy - j0);
  int kmin_l = 0, kmax_l = 0;
  if (EPI == EPI_NORM) zoom_mm_issue(E, kmin_l, kmax_l);  // consumed behind the window loads
  // requested now: the y taps of the wave's rows (one per lane), the z taps of this lane's four outputs
  int4 yt = make_int4(-1, 0, 0, 0);
  if (tx < nj) yt = *reinterpret_cast<const int4*>(Z.ty + j0 + tx);
  int zlo[4], zhi[4];
  float zwl[4], zwh[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = tx * 4 + u;
    const int4 c = *reinterpret_cast<const int4*>(Z.tz + min(k, Z.dz - 1));
    const bool ok = k < Z.dz && c.x >= 0;
    zlo[u] = ok ? c.x : 0; zhi[u] = ok ? c.y : 0;
    zwl[u] = ok ? __builtin_bit_cast(float, c.z) : 0.f;  // "outside": 0 * v + 0 * v = 0
    zwh[u] = ok ? __builtin_bit_cast(float, c.w) : 0.f;
  }
  const fsg_tap a = zuniform_tap(Z.tx, i);
  int smin = yt.x >= 0 ? yt.x : 0x7FFFFFFF, smax = yt.x >= 0 ? yt.y : -1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    smin = min(smin, __shfl_xor(smin, o, FSG_WAVE));
    smax = max(smax, __shfl_xor(smax, o, FSG_WAVE));
  }
  smin = __builtin_amdgcn_readfirstlane(smin);
  smax = __builtin_amdgcn_readfirstlane(smax);
  const int nrows = smax - smin + 1;
  const bool okx = a.lo >= 0 && nrows > 0;
  const bool fits = nrows <= ZW_XR;  // uniform; false only for tables that are not a plain zoom
  if (okx && fits) {
    const float* pa = Z.src + ((size_t)a.lo * Z.sy + smin) * Z.sz;
    const float* pb = Z.src + ((size_t)a.hi * Z.sy + smin) * Z.sz;
    const int tot = nrows * Z.sz;  // contiguous in the source: one linear, coalesced sweep
    // ZW_B trips of the sweep at a time, every load unconditional (index clamped): all 2 ZW_B loads of a batch are in
    // flight together.  A plain `for (e = tx; e < tot; e += 64)` is a chain of ~20 round trips to memory per wave.
    for (int e0 = 0; e0 < tot; e0 += 64 * ZW_B) {
      float va[ZW_B], vb[ZW_B];
#pragma unroll
      for (int u = 0; u < ZW_B; ++u) {
        const int e = min(e0 + u * 64 + tx, tot - 1);
        va[u] = pa[e];
        vb[u] = pb[e];
      }
#pragma unroll
      for (int u = 0; u < ZW_B; ++u) {
        const int e = e0 + u * 64 + tx;
        if (e < tot) xs[e] = fsg_mix(a.w_lo, va[u], a.w_hi, vb[u]);
      }
    }
  }
  float mnq = 0.f, den = 1.f, mx = 1.f;
  if (EPI == EPI_NORM) {
    float mn;
    zoom_mm_finish(kmin_l, kmax_l, mn, mx);
    mnq = mn / mx;
    den = 1.0f - mnq;
  }
  const UniDiv ud = unidiv_make(mx);
  const UniDiv udd = unidiv_make(den);
  zwave_sync();
  float lo = INFINITY, hi = -INFINITY;
  const bool dst16 = (Z.dz & 3) == 0 && ((((uintptr_t)Z.dst) & 15) == 0);
  const bool full = tx * 4 + 3 < Z.dz;
  for (int jj = 0; jj < nj; ++jj) {
    const int blo = __builtin_amdgcn_readlane(yt.x, jj), bhi = __builtin_amdgcn_readlane(yt.y, jj);
    const float bwl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(yt.z, jj));
    const float bwh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(yt.w, jj));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (okx && blo >= 0) {
      if (fits) {
        const float* xl = xs + (blo - smin) * Z.sz;
        const float* xh = xs + (bhi - smin) * Z.sz;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float ylo = fsg_mix(bwl, xl[zlo[u]], bwh, xh[zlo[u]]);
          const float yhi = fsg_mix(bwl, xl[zhi[u]], bwh, xh[zhi[u]]);
          v[u] = fsg_mix(zwl[u], ylo, zwh[u], yhi);
        }
      } else {
        for (int u = 0; u < 4; ++u)
          if (tx * 4 + u < Z.dz && (zwl[u] != 0.f || zwh[u] != 0.f))
            v[u] = zoom_slab_slow(Z.src, Z.sy, Z.sz, a, fsg_tap{blo, bhi, bwl, bwh}, fsg_tap{zlo[u], zhi[u], zwl[u], zwh[u]});
      }
    }
    const size_t o0 = ((size_t)i * Z.dy + j0 + jj) * Z.dz + (size_t)tx * 4;
    if (EPI == EPI_NORM) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#ifdef FSG_NO_UNIDIV
        float t = v[u] / mx;
#else
        float t = unidiv(ud, v[u]);
#endif
        if (E.norm_mode == 1) t = (mnq == 1.0f) ? t * 0.0f : (den == 1.0f ? t - mnq : unidiv(udd, t - mnq));
        v[u] = t;
      }
    }
    if (EPI == EPI_MINMAX) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tx * 4 + u < Z.dz) { lo = fminf(lo, v[u]); hi = fmaxf(hi, v[u]); }
    } else if (full && dst16) {
      *reinterpret_cast<float4*>(Z.dst + o0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tx * 4 + u < Z.dz) Z.dst[o0 + u] = v[u];
    }
  }
  if (EPI == EPI_MINMAX) {
    lo = fsg_wave_min(lo);
    hi = fsg_wave_max(hi);
    if (tx == 0) {
      if (E.mm_shards > 1) {  // slot by wave, not by workgroup: the four waves of a workgroup finish at different times
        int32_t* s_ = E.mm_out + (int)((blockIdx.x * 4u + (unsigned)wave) % (unsigned)E.mm_shards) * FSG_MM_SLOT_STRIDE;
        (void)__hip_atomic_fetch_min(&s_[0], fsg_f2key(lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_max(&s_[1], fsg_f2key(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        fsg_atomic_min_key(&E.mm_out[0], lo);
        fsg_atomic_max_key(&E.mm_out[1], hi);
      }
    }
  }
}

int g_zoom_ty = 16;            // output y rows per workgroup of zoom_tile_kernel
int g_zoom_cap = 12288;        // LDS floats for the x-blended source window (48 KB)

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
  // measured at 256^3: the tile kernel wins where the Philox epilogue dominates (down-sampling + noise, K7: 27 vs 33 us
  // at m = 171, 42 vs 68 us at m = 256); the row kernels stay ahead for the up-sampling passes of K9 (30 vs 35-38 us)
  // K7 at large low-res sizes: the slab kernel is ahead from ~200 outputs per row on (m = 220: 32 vs 41 us, m = 256: 39 vs 47 us;
  // m = 171: 36 vs 27, m = 128: 20 vs 18 -- profiles/r02_b_zoom_experiments.txt), so the noise epilogues switch there
  // (only for row lengths that are a multiple of 4: the slab kernel draws one Philox block per aligned quad of the OUTPUT
  // index, which needs every row to start on a multiple of 4 -- otherwise it draws per element, 45-55 us: tools/k7_sweep.py)
  const bool noise_big = (EPI == EPI_NOISE_PHILOX || EPI == EPI_NOISE_PTR) && Z.dz >= 196 && (Z.dz & 3) == 0 && Z.sz <= 256 &&
                         Z.dz <= 256;
  if ((EPI == EPI_NOISE_PHILOX || EPI == EPI_NOISE_PTR || (g_tuning_flags & FSG_TUNE_TILE_ZOOM)) && !noise_big &&
      !(g_tuning_flags & (FSG_TUNE_GENERIC_ZOOM | FSG_TUNE_ROW_ZOOM | FSG_TUNE_SLAB_ZOOM))) {
    // window estimate for the tile kernel: TY rows advance sy/dy source rows each (+2 for the pair and rounding)
    int TY = g_zoom_ty < ZT_MAX_TY ? g_zoom_ty : ZT_MAX_TY;
    if (TY > Z.dy) TY = Z.dy;
    const long long est = ((long long)TY * Z.sy / Z.dy + 3) * Z.sz;
    if (TY >= 1 && est <= g_zoom_cap && est + 4LL * Z.dz <= 16000) {  // window + z taps within the 64 KB dynamic-LDS limit
      const int tiles_y = (Z.dy + TY - 1) / TY;
      const size_t lds = ((size_t)est + 4 * (size_t)Z.dz) * sizeof(float);  // window + z taps: as many workgroups per CU as fit
      hipLaunchKernelGGL(zoom_tile_kernel<EPI>, dim3((unsigned)(Z.dx * tiles_y)), dim3(256), lds, fsg_stream(stream), Z, E, TY,
                         (int)est);
      FSG_RETURN_LAUNCH();
    }
  }
  const bool noise_epi = EPI == EPI_NOISE_PHILOX || EPI == EPI_NOISE_PTR;
  // r03: the wave kernel (independent waves, outputs straight from the x-blended window) for the passes without a noise draw
  // Opt-in (FSG_TUNE_WAVE_ZOOM): measured SLOWER than the slab kernel at 256^3 (K9b 36.6 vs 27.5 us at m = 171, K9a 23.7 vs 17.4:
  // profiles/r03_c_zoom_wave_pmc.json) -- both are bound by instruction issue and LDS waits (~160 VALU + ~140 SALU per output
  // row and wave, half of it the exact-division epilogue), and the direct form doubles the LDS bank conflicts.
  if ((g_tuning_flags & FSG_TUNE_WAVE_ZOOM) && !noise_epi && !noise_big && Z.dz <= 256 && Z.sz <= 1024 &&
      !(g_tuning_flags & (FSG_TUNE_GENERIC_ZOOM | FSG_TUNE_ROW_ZOOM | FSG_TUNE_TILE_ZOOM | FSG_TUNE_SLAB_ZOOM))) {
    const size_t lds = (size_t)4 * ZW_XR * Z.sz * sizeof(float);
    // window estimate as for the tile / slab kernels: ZW_ROWS outputs advance sy/dy source rows each (+2 for the pair and rounding)
    if (lds <= 64000 && (long long)ZW_ROWS * Z.sy / Z.dy + 3 <= ZW_XR) {
      const int tiles_y = (Z.dy + 4 * ZW_ROWS - 1) / (4 * ZW_ROWS);
      hipLaunchKernelGGL(zoom_wave_kernel<EPI>, dim3((unsigned)(Z.dx * tiles_y)), dim3(256), lds, fsg_stream(stream), Z, E);
      FSG_RETURN_LAUNCH();
    }
  }
  // measured at 256^3 (profiles/r02_b_zoom_experiments.txt): the slab kernel wins for the passes that store (K9b 28 vs 32.5 us);
  // the min/max pass ends every workgroup with two gated atomics, which cost the slab kernel's 4 096 workgroups more than the
  // row kernel's 2 048 (38 vs 31 us; 22 us with the atomics removed) whether or not the keys are sharded over slots
  if (((!noise_epi && (EPI != EPI_MINMAX || E.mm_shards > 1)) || noise_big || (g_tuning_flags & FSG_TUNE_SLAB_ZOOM)) &&
      !(g_tuning_flags & (FSG_TUNE_GENERIC_ZOOM | FSG_TUNE_ROW_ZOOM | FSG_TUNE_TILE_ZOOM))) {
    // tile height: the kernel's fixed part per workgroup (taps, window bounds, x stage: a chain of global round trips) costs
    // 10-12 us of its 19-27 us at 16 rows per tile (profiles/r02_b_zoom_experiments.txt, ablation).  Twice the rows per
    // workgroup pays for the pass that stores nothing (K9a 19.0 -> 17.3 us); the storing passes lose it again (27.4 -> 28.5)
    // r03_h, with the row loop at ~45 instead of ~100 vector instructions: four / two times the rows win again where the window
    // still fits (m = 128: K9a 15.5 -> 14.3 us at 64 rows, K9b 26.4 -> 24.3 at 32; m = 171: 18.5 -> 17.6 / same; m = 220: same).
    // The tallest tile whose window fits the LDS budget: 4x, 2x, 1x the tuned height for the min/max pass, 2x, 1x otherwise.
    int TY = 0;
    long long est = 0, total = 0;
    for (int mult = (EPI == EPI_MINMAX ? 4 : 2); mult >= 1; mult >>= 1) {
      TY = mult * g_zoom_ty < ZT_MAX_TY ? mult * g_zoom_ty : ZT_MAX_TY;
      if (TY > Z.dy) TY = Z.dy;
      est = ((long long)TY * Z.sy / Z.dy + 3) * Z.sz;
      total = est + 5LL * 256;  // window (+ 256 floats behind it: unclamped column reads) + the four waves' rows (256 each)
      if (est <= g_zoom_cap && total <= 16000) break;
    }
    if (TY >= 1 && est <= g_zoom_cap && total <= 16000 && Z.sz <= 256 && Z.dz <= 256) {
      const int tiles_y = (Z.dy + TY - 1) / TY;
      const unsigned ntiles = (unsigned)(Z.dx * tiles_y);
      hipLaunchKernelGGL(zoom_slab_kernel<EPI>, dim3(ntiles), dim3(256), (size_t)total * sizeof(float),
                         fsg_stream(stream), Z, E, TY, (int)est);
      FSG_RETURN_LAUNCH();
    }
  }
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

int fsg_zoom_set_tuning(int y_rows, int cap_floats) {
  if (y_rows < 1 || y_rows > ZT_MAX_TY || cap_floats < 256 || cap_floats > 16000) return FSG_E_BADARG;
  g_zoom_ty = y_rows;
  g_zoom_cap = cap_floats;
  return 0;
}

int fsg_zoom3d_minmax_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                          const fsg_tap* tz, int dx, int dy, int dz, int32_t* mm, void* stream) {
  int rc = check(src, sx, sy, sz, tx, ty, tz, dx, dy, dz);
  if (rc) return rc;
  if (!mm) return FSG_E_BADARG;
  ZoomK Z{src, sx, sy, sz, tx, ty, tz, nullptr, dx, dy, dz};
  EpiZ E{};
  E.mm_out = mm;
  E.mm_shards = 1;
#ifdef FSG_DIAG  // timing ablation (results are wrong): only in a -DFSG_DIAG build, never in the shipped library
  if (getenv("FSG_DIAG_NO_MM_ATOMICS")) E.norm_mode = 77;
#endif
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
  E.mm_in = mm; E.norm_mode = mode; E.mm_shards = 1;
  return launch1<EPI_NORM>(Z, E, stream);
}

int fsg_zoom3d_minmax_sharded_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                                  const fsg_tap* tz, int dx, int dy, int dz, int32_t* slots, int nslots, void* stream) {
  int rc = check(src, sx, sy, sz, tx, ty, tz, dx, dy, dz);
  if (rc) return rc;
  if (!slots || nslots < 2 || nslots > 64) return FSG_E_BADARG;
  ZoomK Z{src, sx, sy, sz, tx, ty, tz, nullptr, dx, dy, dz};
  EpiZ E{};
  E.mm_out = slots;
  E.mm_shards = nslots;
  return launch1<EPI_MINMAX>(Z, E, stream);
}

int fsg_zoom3d_normalise_sharded_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                                     const fsg_tap* tz, float* dst, int dx, int dy, int dz, const int32_t* slots, int nslots,
                                     int mode, void* stream) {
  int rc = check(src, sx, sy, sz, tx, ty, tz, dx, dy, dz);
  if (rc) return rc;
  if (!dst || !slots || src == dst || (mode != 0 && mode != 1) || nslots < 2 || nslots > 64) return FSG_E_BADARG;
  ZoomK Z{src, sx, sy, sz, tx, ty, tz, dst, dx, dy, dz};
  EpiZ E{};
  E.mm_in = slots; E.norm_mode = mode; E.mm_shards = nslots;
  return launch1<EPI_NORM>(Z, E, stream);
}

}  // extern "C"
