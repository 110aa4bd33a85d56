// fsg_blur_rs.hip -- K6 + K7 (+ K8) fused per axis: Gaussian blur and axis-aligned down-sampling as ONE operator per axis.
//
// Reference (paths relative to /root/reference/fetalsyngen/): RandResample.__call__ = gaussian_blur_3d (utils/generation.py:
// 84-110: conv3d along x, then y, then z, zero padded, no border renormalisation) followed by fast_3D_interp_torch on an
// axis-aligned grid (generator/augmentation/synthseg.py:84-104; lerp chain x, then y, then z, strict > 0 validity:
// utils/generation.py:227-278), then RandNoise at the low resolution (synthseg.py:230-233).
// Both steps are separable linear operators, so  L_z L_y L_x B_z B_y B_x  =  (L_z B_z)(L_y B_y)(L_x B_x)  in exact arithmetic:
// operators of different axes commute.  Per axis, output j is
//        out[j] = w_lo[j] * B[lo[j]] + w_hi[j] * B[hi[j]],     B[i] = sum_t k[t] * in[i + t - R]   (zeros outside),
// so the blurred row B[i] of an axis is only ever needed by the one or two outputs that reference it: it can live in
// registers for the few instructions between its last tap and the lerp, and the blurred full-resolution volume never exists:
//        unfused (r02):  blur x 8N + blur y,z 8N + K7 (4N + 4M)            = 21 B/voxel at mu = 0.3
//        here:           x pass 4N + 4N (m/n)  +  y,z pass 4N (m/n) + 4M   = 10.5 B/voxel
// Rounding differs from the reference's order by a few ulp of the 0..255 values (the blur's own tolerance is atol 1e-3:
// conv3d's summation order is unspecified); quirks kept: zero-padded un-renormalised borders, an output whose position is
// outside (0, n-1] is 0 (lo < 0 in the table), noise added after, negatives clamped.
//
// Per axis the arithmetic is the reference's own: B[i] accumulated over ascending taps (fmaf, as fsg_blur.hip), then
// w_lo * B[lo] + w_hi * B[hi] with separate multiplies and an add (fsg_mix, as fsg_zoom.hip); only the order ACROSS axes differs.
//
//   blur_rs_x_kernel<R>  : axis 0.  Thread = one float4 column and a chunk of TL = 16 input rows: the body of blur_strided_v4
//                          (TL + 1 blurred rows from TL + 1 + 2R coalesced row loads, register sliding window, everything
//                          unrolled), then the outputs whose lower neighbour lies in the chunk are emitted straight from those
//                          registers -- the row index is wave-uniform, so "register lo - l0" is a uniform switch (scalar
//                          branches), not a gather.  The first output of a chunk is found by a ballot over the tap table.
//   blur_rs_yz_kernel<R> : axes 1 + 2 of one x-plane and 16 input y rows: the same body along y (4 + 1 blurred rows per wave)
//                          emitting its output rows into LDS, then per row the z blur on aligned ds_read_b128 windows into a
//                          second LDS buffer, then the z lerp + noise + clamp over the tile's contiguous output range (one
//                          Philox block per aligned quad of the flat output index, exactly K7's indexing), 16-byte stores.
#include "fsg_common.h"

namespace {

constexpr int RS_MAXR = 8;
constexpr int RS_KCAP = 2 * RS_MAXR + 1;
struct TapsK { float w[RS_KCAP + 3]; };

__device__ __forceinline__ fsg_tap rs_uniform_tap(const fsg_tap* t, int idx) {
  const int4 v = *reinterpret_cast<const int4*>(t + idx);
  fsg_tap r;
  r.lo = __builtin_amdgcn_readfirstlane(v.x);
  r.hi = __builtin_amdgcn_readfirstlane(v.y);
  r.w_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(v.z));
  r.w_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(v.w));
  return r;
}

// First output j (of m) whose lower neighbour lo[j] >= l0, for a table whose lo is non-decreasing in j (every plain
// resampling table: positions delta + j * n / m).  Outputs with lo < 0 ("outside": value 0) sort first.  Wave-uniform
// result; `guess` may be anything, a good one makes it one ballot.
__device__ __forceinline__ int rs_first_output(const fsg_tap* __restrict__ tab, int m, int l0, int guess) {
  const int lane = threadIdx.x & 63;
  int jg = min(max(guess, 0), m);
  while (jg > 0 && __builtin_amdgcn_readfirstlane(tab[jg - 1].lo) >= l0) jg = max(jg - 48, 0);  // guess too high
  for (;;) {
    const int j = jg + lane;
    const bool hit = j >= m || tab[j].lo >= l0;
    const unsigned long long b = __ballot(hit);
    if (b) return jg + (int)__builtin_ctzll(b);
    jg += 64;
  }
}

__device__ __forceinline__ float4 rs_mix4(float wl, const float4& a, float wh, const float4& b) {
  return make_float4(fsg_mix(wl, a.x, wh, b.x), fsg_mix(wl, a.y, wh, b.y), fsg_mix(wl, a.z, wh, b.z), fsg_mix(wl, a.w, wh, b.w));
}

// acc[d] / acc[d + 1] with a wave-uniform d: a uniform switch -- one case runs, on scalar branches
template <int NB>
__device__ __forceinline__ float4 rs_lerp_rows(const float4 (&acc)[NB], int d, bool same, float wl, float wh) {
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#define RS_CASE(K)                                                                              \
  case K:                                                                                       \
    if (K < NB) out = rs_mix4(wl, acc[K < NB ? K : 0], wh, same ? acc[K < NB ? K : 0] : acc[K + 1 < NB ? K + 1 : 0]); \
    break;
  switch (d) {
    RS_CASE(0) RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
    RS_CASE(9) RS_CASE(10) RS_CASE(11) RS_CASE(12) RS_CASE(13) RS_CASE(14) RS_CASE(15) RS_CASE(16)
    default: break;
  }
#undef RS_CASE
  return out;
}

// ---- axis 0 ------------------------------------------------------------------------------------------------------------
constexpr int RSX_TL = 16;

template <int R>
__global__ __launch_bounds__(256) void blur_rs_x_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n, int m,
                                                        int inner4, const fsg_tap* __restrict__ tab, TapsK K) {
  constexpr int TL = RSX_TL, NB = TL + 1;  // NB blurred rows l0 .. l0 + TL: the last one only ever serves as an upper neighbour
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c = blockIdx.x * 64 + tx;
  const int l0 = (blockIdx.y * 4 + ty) * TL;
  if (l0 >= n) return;  // whole wave
  const bool live = c < inner4;
  const float4* s = src + (live ? c : 0);
  float4 acc[NB];
#pragma unroll
  for (int o = 0; o < NB; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < NB + 2 * R; ++t) {
    const int l = l0 + t - R;
    // unconditional load of a clamped row, zeroed afterwards when outside the axis (zero padding): a predicated load hides
    // the number of loads in flight from the compiler, which then waits for all of them before every use
    float4 v = s[(size_t)min(max(l, 0), n - 1) * inner4];
    const bool in = l >= 0 && l < n;
    v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
#pragma unroll
    for (int o = 0; o < NB; ++o) {
      const int tap = t - o;
      if (tap >= 0 && tap <= 2 * R) {
        const float w = K.w[tap];
        acc[o].x = fmaf(w, v.x, acc[o].x);
        acc[o].y = fmaf(w, v.y, acc[o].y);
        acc[o].z = fmaf(w, v.z, acc[o].z);
        acc[o].w = fmaf(w, v.w, acc[o].w);
      }
    }
  }
  // outputs whose lower neighbour is in [l0, l0 + TL); chunk 0 also owns the "outside" outputs (lo < 0 -> 0)
  const float f = (float)m / (float)n;
  int j = l0 == 0 ? 0 : rs_first_output(tab, m, l0, (int)(((float)l0 - 0.5f / f) * f) - 2);
  float4* d = dst + c;
  for (; j < m; ++j) {
    const fsg_tap a = rs_uniform_tap(tab, j);
    if (a.lo >= l0 + TL) break;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.lo >= 0) out = rs_lerp_rows<NB>(acc, a.lo - l0, a.hi == a.lo, a.w_lo, a.w_hi);
    if (live) d[(size_t)j * inner4] = out;
  }
}

// ---- axes 1 + 2 ----------------------------------------------------------------------------------------------------------
constexpr int RSY_TL = 4, RSY_IN = 4 * RSY_TL, RSY_ROWS = RSY_IN + 2;  // input y rows per workgroup; output rows it can emit (m <= n)

struct NoiseK {
  int mode;  // 0 none, 1 pointer, 2 Philox
  const float* noise;
  uint64_t seed, stream_id;
  float std;
};

template <int R>
__global__ __launch_bounds__(256) void blur_rs_yz_kernel(const float4* __restrict__ src, float* __restrict__ dst, int ny, int nz,
                                                         int m1, int m2, const fsg_tap* __restrict__ taby,
                                                         const fsg_tap* __restrict__ tabz, TapsK Ky, TapsK Kz, NoiseK NZ) {
  constexpr int RP = (R + 3) & ~3, TL = RSY_TL, NB = TL + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int inner4 = nz >> 2, pitch = nz + 2 * RP;
  float* bufA = lds;                                   // [RSY_ROWS][pitch]  y-resampled rows, zero z halos
  float* bufD = bufA + RSY_ROWS * pitch;               // [RSY_ROWS][nz]     z-blurred rows
  fsg_tap* tc = reinterpret_cast<fsg_tap*>(bufD + RSY_ROWS * nz);  // [m2] z taps
  __shared__ int s_j[5];                               // first output row of every wave's chunk; [4] = end of the tile
  const int tx = threadIdx.x, tyc = threadIdx.y, tid = tyc * 64 + tx;
  const int tiles_y = (ny + RSY_IN - 1) / RSY_IN;
  const int tile = xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int bx = tile / tiles_y;
  const int yin0 = (tile - bx * tiles_y) * RSY_IN;
  const size_t plane4 = (size_t)bx * ny * inner4;
  for (int k = tid; k < m2; k += 256) tc[k] = tabz[k];
  for (int e = tid; e < RSY_ROWS * (2 * RP / 4); e += 256) {  // zero z halos of every row of A
    const int r = e / (2 * RP / 4), h = e - r * (2 * RP / 4);
    float* row = bufA + (size_t)r * pitch;
    reinterpret_cast<float4*>(h < RP / 4 ? row : row + RP + nz)[h < RP / 4 ? h : h - RP / 4] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- y: blur in registers, resampled rows into A ----
  const int l0 = yin0 + tyc * TL;
  const float f = (float)m1 / (float)ny;
  const int jb = l0 == 0 ? 0 : rs_first_output(taby, m1, l0, (int)(((float)l0 - 0.5f / f) * f) - 2);
  if (tx == 0) s_j[tyc] = jb;
  if (tyc == 3 && tx == 0) {
    const int lend = yin0 + RSY_IN;
    s_j[4] = lend >= ny ? m1 : rs_first_output(taby, m1, lend, (int)(((float)lend - 0.5f / f) * f) - 2);
  }
  // (wave 3's second search is executed by the whole wave -- rs_first_output is a wave-level routine; only lane 0 stores)
  __syncthreads();
  const int jt0 = s_j[0], nj = min(s_j[4] - s_j[0], RSY_ROWS);
  for (int c = tx; c < inner4; c += 64) {
    const float4* s = src + plane4 + c;
    float4 acc[NB];
#pragma unroll
    for (int o = 0; o < NB; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < NB + 2 * R; ++t) {
      const int l = l0 + t - R;
      float4 v = s[(size_t)min(max(l, 0), ny - 1) * inner4];
      const bool in = l >= 0 && l < ny;
      v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
#pragma unroll
      for (int o = 0; o < NB; ++o) {
        const int tap = t - o;
        if (tap >= 0 && tap <= 2 * R) {
          const float w = Ky.w[tap];
          acc[o].x = fmaf(w, v.x, acc[o].x);
          acc[o].y = fmaf(w, v.y, acc[o].y);
          acc[o].z = fmaf(w, v.z, acc[o].z);
          acc[o].w = fmaf(w, v.w, acc[o].w);
        }
      }
    }
    if (l0 < ny) {
      for (int j = jb; j < m1 && j - jt0 < RSY_ROWS; ++j) {
        const fsg_tap a = rs_uniform_tap(taby, j);
        if (a.lo >= l0 + TL) break;
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.lo >= 0) out = rs_lerp_rows<NB>(acc, a.lo - l0, a.hi == a.lo, a.w_lo, a.w_hi);
        reinterpret_cast<float4*>(bufA + (size_t)(j - jt0) * pitch + RP)[c] = out;
      }
    }
  }
  __syncthreads();
  // ---- z blur: wave w takes rows w, w + 4, ... of A into D (taps ascending with fmaf, as blur_contig_lds) ----
  for (int r = tyc; r < nj; r += 4) {
    const float* row = bufA + (size_t)r * pitch;
    float4* d4 = reinterpret_cast<float4*>(bufD + (size_t)r * nz);
    for (int q = tx; q < inner4; q += 64) {
      float win[4 + 2 * RP];
#pragma unroll
      for (int u = 0; u < (4 + 2 * RP) / 4; ++u) {
        const float4 t = reinterpret_cast<const float4*>(row)[q + u];
        win[4 * u] = t.x; win[4 * u + 1] = t.y; win[4 * u + 2] = t.z; win[4 * u + 3] = t.w;
      }
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t <= 2 * R; ++t) {
        const float w = Kz.w[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(w, win[RP - R + e + t], o[e]);
      }
      d4[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();
  // ---- z resample + noise + clamp over the tile's contiguous output range (flat quads: one Philox block each) ----
  const size_t obase = ((size_t)bx * m1 + jt0) * m2;
  const size_t oend = obase + (size_t)max(nj, 0) * m2;
  const float inv_m2 = 1.0f / (float)m2;
  for (size_t q = (obase >> 2) + tid; (q << 2) < oend; q += 256) {
    const size_t o0 = q << 2;
    float v[4];
    bool live[4];
    const size_t first = o0 < obase ? obase : o0;
    int rel = (int)(first - obase);
    int jj = (int)((float)rel * inv_m2);
    int k = rel - jj * m2;
    if (k < 0) { --jj; k += m2; }
    if (k >= m2) { ++jj; k -= m2; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = o0 + u;
      live[u] = o >= obase && o < oend;
      v[u] = 0.f;
      if (live[u]) {
        const fsg_tap cz = tc[k];
        if (cz.lo >= 0) {
          const float* rd = bufD + (size_t)jj * nz;
          v[u] = fsg_mix(cz.w_lo, rd[cz.lo], cz.w_hi, rd[cz.hi]);
        }
        if (++k == m2) { k = 0; ++jj; }
      }
    }
    if (NZ.mode == 2) {
      const float4 z = fsg_randn4(NZ.seed, NZ.stream_id, (uint64_t)q);
      v[0] += NZ.std * z.x; v[1] += NZ.std * z.y; v[2] += NZ.std * z.z; v[3] += NZ.std * z.w;
    } else if (NZ.mode == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live[u]) v[u] += NZ.std * NZ.noise[o0 + u];
    }
    if (NZ.mode != 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = v[u] < 0.f ? 0.f : v[u];
    }
    if (live[0] && live[3] && ((((uintptr_t)dst) & 15) == 0)) {
      *reinterpret_cast<float4*>(dst + o0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live[u]) dst[o0 + u] = v[u];
    }
  }
}

size_t rs_yz_lds(int nz, int m2, int R) {
  const int RP = (R + 3) & ~3;
  return ((size_t)RSY_ROWS * (nz + 2 * RP) + (size_t)RSY_ROWS * nz + 4 * (size_t)m2) * sizeof(float);
}

template <int R>
int launch_rs_yz(const float* src, float* dst, int m0, int ny, int nz, int m1, int m2, const fsg_tap* ty, const fsg_tap* tz,
                 const TapsK& Ky, const TapsK& Kz, const NoiseK& NZ, hipStream_t st) {
  const size_t lds = rs_yz_lds(nz, m2, R);
  if (lds > 64000) return FSG_E_ALIGN;
  const unsigned tiles_y = (unsigned)((ny + RSY_IN - 1) / RSY_IN);
  hipLaunchKernelGGL(blur_rs_yz_kernel<R>, dim3(tiles_y * (unsigned)m0), dim3(64, 4), lds, st,
                     reinterpret_cast<const float4*>(src), dst, ny, nz, m1, m2, ty, tz, Ky, Kz, NZ);
  FSG_RETURN_LAUNCH();
}

template <int R>
int launch_rs_x(const float* src, float* dst, int n, int m, int inner4, const fsg_tap* tab, const TapsK& K, hipStream_t st) {
  const int chunks = (n + RSX_TL - 1) / RSX_TL;
  dim3 grid((unsigned)((inner4 + 63) / 64), (unsigned)((chunks + 3) / 4)), block(64, 4);
  hipLaunchKernelGGL(blur_rs_x_kernel<R>, grid, block, 0, st, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n, m, inner4, tab, K);
  FSG_RETURN_LAUNCH();
}

// taps centred in a set of radius R (>= their own radius): zero weights outside -- fmaf(0, v, acc) == acc, so the padded
// blur is bit-identical to the unpadded one
int rs_fill_taps(const float* taps_host, int ntaps, int R, TapsK& K) {
  if (!taps_host || ntaps < 3 || (ntaps & 1) == 0 || ntaps > RS_KCAP || (ntaps >> 1) > R || R > RS_MAXR) return FSG_E_ALIGN;
  const int pad = R - (ntaps >> 1);
  for (int t = 0; t < RS_KCAP + 3; ++t) K.w[t] = (t >= pad && t < pad + ntaps) ? taps_host[t - pad] : 0.f;
  return 0;
}

}  // namespace

extern "C" {

int fsg_blur_resample_supported(int n0, int n1, int n2, int m0, int m1, int m2, int ntaps_x, int ntaps_y, int ntaps_z) {
  if (g_tuning_flags & FSG_TUNE_NO_BLUR_RS) return 0;  // A/B switch: every caller then takes the unfused sequence
  if (n0 <= 0 || n1 <= 0 || n2 <= 0 || m0 <= 0 || m1 <= 0 || m2 <= 0) return 0;
  if (m0 > n0 || m1 > n1 || m2 > n2) return 0;  // down-sampling tables (lo non-decreasing, at most one output per input row + 1)
  for (int nt : {ntaps_x, ntaps_y, ntaps_z})
    if (nt < 3 || (nt & 1) == 0 || nt > RS_KCAP) return 0;
  if ((n2 & 3) || n2 > 512 || ((long long)n1 * n2 & 3)) return 0;
  if ((size_t)n0 * n1 * n2 > (size_t)0x7FFFFFFF) return 0;
  const int Ryz = (ntaps_y > ntaps_z ? ntaps_y : ntaps_z) >> 1;
  if (rs_yz_lds(n2, m2, Ryz) > 64000) return 0;
  return 1;
}

int fsg_blur_resample_x_f32(const float* src, int n0, int n1, int n2, const fsg_tap* tx, int m0, const float* taps_host,
                            int ntaps, float* dst, void* stream) {
  if (!src || !dst || src == dst || !tx) return FSG_E_BADARG;
  if (n0 <= 0 || n1 <= 0 || n2 <= 0 || m0 <= 0) return FSG_E_BADARG;
  if (m0 > n0) return FSG_E_ALIGN;
  const int R = ntaps >> 1;
  TapsK K;
  int rc = rs_fill_taps(taps_host, ntaps, R, K);
  if (rc) return rc;
  const long long inner = (long long)n1 * n2;
  if ((inner & 3) || ((((uintptr_t)src) | ((uintptr_t)dst)) & 15)) return FSG_E_ALIGN;
  if ((size_t)n0 * inner > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  const int inner4 = (int)(inner >> 2);
  hipStream_t st = fsg_stream(stream);
  switch (R) {
    case 1: return launch_rs_x<1>(src, dst, n0, m0, inner4, tx, K, st);
    case 2: return launch_rs_x<2>(src, dst, n0, m0, inner4, tx, K, st);
    case 3: return launch_rs_x<3>(src, dst, n0, m0, inner4, tx, K, st);
    case 4: return launch_rs_x<4>(src, dst, n0, m0, inner4, tx, K, st);
    case 5: return launch_rs_x<5>(src, dst, n0, m0, inner4, tx, K, st);
    case 6: return launch_rs_x<6>(src, dst, n0, m0, inner4, tx, K, st);
    case 7: return launch_rs_x<7>(src, dst, n0, m0, inner4, tx, K, st);
    case 8: return launch_rs_x<8>(src, dst, n0, m0, inner4, tx, K, st);
    default: return FSG_E_ALIGN;
  }
}

int fsg_blur_resample_yz_noise_f32(const float* src, int m0, int n1, int n2, const fsg_tap* ty, const fsg_tap* tz, int m1,
                                   int m2, const float* taps_y_host, int ntaps_y, const float* taps_z_host, int ntaps_z,
                                   int noise_mode, const float* noise, uint64_t seed, uint64_t stream_id, float noise_std,
                                   float* dst, void* stream) {
  if (!src || !dst || src == dst || !ty || !tz) return FSG_E_BADARG;
  if (m0 <= 0 || n1 <= 0 || n2 <= 0 || m1 <= 0 || m2 <= 0 || noise_mode < 0 || noise_mode > 2) return FSG_E_BADARG;
  if (noise_mode == 1 && !noise) return FSG_E_BADARG;
  if (m1 > n1 || m2 > n2) return FSG_E_ALIGN;
  const int R = (ntaps_y > ntaps_z ? ntaps_y : ntaps_z) >> 1;  // one radius for both axes: the narrower tap set is zero-padded
  TapsK Ky, Kz;
  int rc = rs_fill_taps(taps_y_host, ntaps_y, R, Ky);
  if (rc) return rc;
  rc = rs_fill_taps(taps_z_host, ntaps_z, R, Kz);
  if (rc) return rc;
  if ((n2 & 3) || n2 > 512 || (((uintptr_t)src) & 15)) return FSG_E_ALIGN;
  if ((size_t)m0 * n1 * n2 > (size_t)0x7FFFFFFF || (size_t)m0 * m1 * m2 > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  NoiseK NZ{noise_mode, noise, seed, stream_id, noise_std};
  hipStream_t st = fsg_stream(stream);
  switch (R) {
    case 1: return launch_rs_yz<1>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 2: return launch_rs_yz<2>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 3: return launch_rs_yz<3>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 4: return launch_rs_yz<4>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 5: return launch_rs_yz<5>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 6: return launch_rs_yz<6>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 7: return launch_rs_yz<7>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 8: return launch_rs_yz<8>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    default: return FSG_E_ALIGN;
  }
}

}  // extern "C"
