// fsg_common.h -- device helpers shared by the gfx950 kernels of libfsg_hip.so.
// CDNA4 only: wave = 64 lanes, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fsg_hip.h"

// The reference evaluates every interpolation as separate fp32 multiplies and adds (ATen CPU
// element-wise ops).  Contraction to FMA would change sampling positions by 1 ulp and flip
// nearest-neighbour labels, so it is off for the whole library; the blur asks for fmaf explicitly.
#pragma clang fp contract(off)

#define FSG_WAVE 64

extern int g_tuning_flags;

#define FSG_RETURN_LAUNCH()                 \
  do {                                      \
    hipError_t e_ = hipGetLastError();      \
    return (int)e_;                         \
  } while (0)

static inline hipStream_t fsg_stream(void* s) { return (hipStream_t)s; }

// ---------------------------------------------------------------------------------------------
// order-preserving float <-> int32 keys (so integer atomicMin/atomicMax reduce floats, -0 < +0)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int32_t fsg_f2key(float f) {
  int32_t b = __builtin_bit_cast(int32_t, f);
  return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__host__ __device__ __forceinline__ float fsg_key2f(int32_t k) {
  int32_t b = k >= 0 ? k : (k ^ 0x7FFFFFFF);
  return __builtin_bit_cast(float, b);
}

// Global min / max of order keys.  The targets only ever move one way, so a (possibly stale) relaxed read
// that already dominates the candidate proves the atomic is unnecessary: after the first few blocks almost
// every block skips it (same-address atomics serialise at ~11 ns each and 2048 of them cost more than the
// reduction itself).
__device__ __forceinline__ void fsg_atomic_min_key(int32_t* p, float v) {
  const int32_t k = fsg_f2key(v);
  if (k < __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(p, k);
}
__device__ __forceinline__ void fsg_atomic_max_key(int32_t* p, float v) {
  const int32_t k = fsg_f2key(v);
  if (k > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, k);
}

// wave-level min / max (64 lanes) by butterfly shuffles
__device__ __forceinline__ float fsg_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, FSG_WAVE));
  return v;
}
__device__ __forceinline__ float fsg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, FSG_WAVE));
  return v;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: element e of stream (seed, stream_id) is lane e%4 of block e/4
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 fsg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                   uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    // three-input xor in one instruction (v_bitop3_b32, truth table 0x96; gfx950): 20 instead of 40 xors per block
    uint32_t n0 = __builtin_amdgcn_bitop3_b32(h1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(h0, c3, k1, 0x96);
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += W0; k1 += W1;
  }
  return make_uint4(c0, c1, c2, c3);
}

// four standard normals for counter block `blk`.  UNIFORM_KEY: the seed is wave-uniform (a kernel argument) and the call sits
// in a loop -- the two key words pass through an empty asm so that the ten round keys are ten scalar adds per call instead of
// twenty loop-invariant scalar registers (which the register allocator spilled to vector lanes: 8 v_readlane + 11 s_nop per
// GMM group, 16-130 spilled registers in the blur + resample kernels).
template <bool UNIFORM_KEY = false>
__device__ __forceinline__ float4 fsg_randn4(uint64_t seed, uint64_t stream_id, uint64_t blk) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  if (UNIFORM_KEY) asm volatile("" : "+s"(k0), "+s"(k1));
  uint4 r = fsg_philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)stream_id,
                              (uint32_t)(stream_id >> 32), k0, k1);
  const float S = 5.9604644775390625e-08f;  // 2^-24
  float u0 = (float)((r.x >> 8) + 1u) * S;  // (0, 1]
  float u1 = (float)((r.z >> 8) + 1u) * S;
  float t0 = (float)(r.y >> 8) * S;         // [0, 1) revolutions
  float t1 = (float)(r.w >> 8) * S;
  // -2 ln u = -2 ln2 * log2 u ; v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 (input in revolutions)
  float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
  float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  float4 z;
  z.x = r0 * __builtin_amdgcn_cosf(t0);
  z.y = r0 * __builtin_amdgcn_sinf(t0);
  z.z = r1 * __builtin_amdgcn_cosf(t1);
  z.w = r1 * __builtin_amdgcn_sinf(t1);
  return z;
}
__device__ __forceinline__ float fsg_randn1(uint64_t seed, uint64_t stream_id, uint64_t e) {
  float4 z = fsg_randn4(seed, stream_id, e >> 2);
  switch ((int)(e & 3)) {
    case 0: return z.x;
    case 1: return z.y;
    case 2: return z.z;
    default: return z.w;
  }
}

// K1 voxel loop shared by gmm_x4_kernel and the fused head of a sample (fsg_deform.hip): workgroup `blk` of `nblk`,
// tables already in LDS.
__device__ __forceinline__ void fsg_gmm_x4_loop(const uint8_t* __restrict__ l0, const uint8_t* __restrict__ l1,
                                                const uint8_t* __restrict__ l2, const uint8_t* __restrict__ l3, size_t n,
                                                const float* s_mu, const float* s_sg, const float* __restrict__ noise,
                                                uint64_t seed, uint64_t stream_id, float* __restrict__ out, unsigned blk,
                                                unsigned nblk) {
  const size_t ngrp = (n + 3) >> 2;
  if (!noise && (n & 3) == 0 && n <= ((size_t)1 << 30)) {
    // The usual case on a lean path (r03): whole groups only, 32-bit offsets against uniform bases (one address register per
    // access instead of a 64-bit add per pointer), no tail or injected-noise branches inside the loop.  Same values.
    const uint8_t *q1 = l1 ? l1 : l0, *q2 = l2 ? l2 : l0, *q3 = l3 ? l3 : l0;
    const uint32_t k1 = l1 ? 0xFFFFFFFFu : 0u, k2 = l2 ? 0xFFFFFFFFu : 0u, k3 = l3 ? 0xFFFFFFFFu : 0u;
    const uint32_t ng = (uint32_t)ngrp, step = nblk * blockDim.x;
    // (two groups per trip with their eight label words in flight together measured the same: the kernel runs at the cold-HBM
    // rate of its 8 B/voxel, profiles/r03_g_notes.txt)
    for (uint32_t g = blk * blockDim.x + threadIdx.x; g < ng; g += step) {
      const uint32_t e = g << 2;
      const uint32_t w0 = *reinterpret_cast<const uint32_t*>(l0 + e), w1 = *reinterpret_cast<const uint32_t*>(q1 + e);
      const uint32_t w2 = *reinterpret_cast<const uint32_t*>(q2 + e), w3 = *reinterpret_cast<const uint32_t*>(q3 + e);
      const uint32_t w = w0 + (w1 & k1) + (w2 & k2) + (w3 & k3);
      const float4 r = fsg_randn4<true>(seed, stream_id, (uint64_t)g);
      const float z[4] = {r.x, r.y, r.z, r.w};
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int l = (int)((w >> (8 * q)) & 255u);
        v[q] = fmaxf(s_mu[l] + s_sg[l] * z[q], 0.f);  // (no NaN, and no -0: mu >= +0)
      }
      *reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + (size_t)(g << 4)) = make_float4(v[0], v[1], v[2], v[3]);
    }
    return;
  }
  // The four label words of a group are requested TOGETHER: with `if (l1) w += *l1` the compiler put each load, its wait and
  // its add into a branch of their own -- four dependent round trips to memory per group, and the kernel ran at that latency
  // (36 us) whatever the arithmetic did; the same draw with the loads issued back to back is memory bound at 21 us
  // (tools/ubench/stream_shapes.hip, profiles/r02_j_gmm_head.txt).  An absent volume reads l0 again under a zero mask.
  const uint8_t *p1 = l1 ? l1 : l0, *p2 = l2 ? l2 : l0, *p3 = l3 ? l3 : l0;
  const uint32_t m1 = l1 ? 0xFFFFFFFFu : 0u, m2 = l2 ? 0xFFFFFFFFu : 0u, m3 = l3 ? 0xFFFFFFFFu : 0u;
  for (size_t g = (size_t)blk * blockDim.x + threadIdx.x; g < ngrp; g += (size_t)nblk * blockDim.x) {
    const size_t e = g << 2;
    uint32_t w = 0;
    if (e + 3 < n) {
      const uint32_t w0 = *reinterpret_cast<const uint32_t*>(l0 + e), w1 = *reinterpret_cast<const uint32_t*>(p1 + e);
      const uint32_t w2 = *reinterpret_cast<const uint32_t*>(p2 + e), w3 = *reinterpret_cast<const uint32_t*>(p3 + e);
      w = w0 + (w1 & m1) + (w2 & m2) + (w3 & m3);
    } else {
      for (int q = 0; q < 4 && e + q < n; ++q) {
        uint32_t b = l0[e + q];
        if (l1) b += l1[e + q];
        if (l2) b += l2[e + q];
        if (l3) b += l3[e + q];
        w |= (b & 255u) << (8 * q);
      }
    }
    float z[4];
    if (noise) {
#pragma unroll
      for (int q = 0; q < 4; ++q) z[q] = (e + q < n) ? noise[e + q] : 0.f;
    } else {
      const float4 r = fsg_randn4<true>(seed, stream_id, (uint64_t)g);
      z[0] = r.x; z[1] = r.y; z[2] = r.z; z[3] = r.w;
    }
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int l = (int)((w >> (8 * q)) & 255u);
      const float t = s_mu[l] + s_sg[l] * z[q];
      v[q] = t < 0.f ? 0.f : t;
    }
    if (e + 3 < n) {
      *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int q = 0; q < 4 && e + q < n; ++q) out[e + q] = v[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// separable linear interpolation from a small/any grid with per-axis tables, reference order:
//   x: t = wl*a + wh*b (per y,z corner), then y, then z   (utils/generation.py:376-386)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fsg_mix(float wl, float a, float wh, float b) { return wl * a + wh * b; }

template <int NCH>
__device__ __forceinline__ float fsg_tab_interp(const float* __restrict__ src, int sy, int sz, int ch,
                                                const fsg_tap& a, const fsg_tap& b, const fsg_tap& c) {
  const size_t rx0 = (size_t)a.lo * sy, rx1 = (size_t)a.hi * sy;
  const float* p00 = src + ((rx0 + b.lo) * sz) * NCH + ch;
  const float* p01 = src + ((rx0 + b.hi) * sz) * NCH + ch;
  const float* p10 = src + ((rx1 + b.lo) * sz) * NCH + ch;
  const float* p11 = src + ((rx1 + b.hi) * sz) * NCH + ch;
  const int z0 = c.lo * NCH, z1 = c.hi * NCH;
  float t00 = fsg_mix(a.w_lo, p00[z0], a.w_hi, p10[z0]);  // y = b.lo, z = c.lo
  float t10 = fsg_mix(a.w_lo, p01[z0], a.w_hi, p11[z0]);  // y = b.hi, z = c.lo
  float t01 = fsg_mix(a.w_lo, p00[z1], a.w_hi, p10[z1]);  // y = b.lo, z = c.hi
  float t11 = fsg_mix(a.w_lo, p01[z1], a.w_hi, p11[z1]);  // y = b.hi, z = c.hi
  float u0 = fsg_mix(b.w_lo, t00, b.w_hi, t10);
  float u1 = fsg_mix(b.w_lo, t01, b.w_hi, t11);
  return fsg_mix(c.w_lo, u0, c.w_hi, u1);
}

// kernel-argument copy of fsg_deform
struct FsgDeformK {
  int n0, n1, n2;
  float A[9];
  float cen[3];
  float c2[3];
  int flip;
  int f0, f1, f2;
  const float* field;
  const fsg_tap* tx;
  const fsg_tap* ty;
  const fsg_tap* tz;
  const float* rows;  // optional: per-(i,j) x/y-interpolated coarse rows (fsg_deform_rows_f32), else null
  int row_stride;
};

// clamped, un-shifted sampling position of grid point (i,j,k)  (affine_nonrigid.py:331-347)
__device__ __forceinline__ void fsg_position(const FsgDeformK& D, int i, int j, int k, float& x, float& y, float& z) {
  float px = (float)i - D.cen[0], py = (float)j - D.cen[1], pz = (float)k - D.cen[2];
  if (D.field) {
    const fsg_tap a = D.tx[i], b = D.ty[j], c = D.tz[k];
    px = px + fsg_tab_interp<3>(D.field, D.f1, D.f2, 0, a, b, c);
    py = py + fsg_tab_interp<3>(D.field, D.f1, D.f2, 1, a, b, c);
    pz = pz + fsg_tab_interp<3>(D.field, D.f1, D.f2, 2, a, b, c);
  }
  x = D.A[0] * px + D.A[1] * py + D.A[2] * pz + D.c2[0];
  y = D.A[3] * px + D.A[4] * py + D.A[5] * pz + D.c2[1];
  z = D.A[6] * px + D.A[7] * py + D.A[8] * pz + D.c2[2];
  const float hx = (float)(D.n0 - 1), hy = (float)(D.n1 - 1), hz = (float)(D.n2 - 1);
  if (x < 0.f) x = 0.f;
  if (y < 0.f) y = 0.f;
  if (z < 0.f) z = 0.f;
  if (x > hx) x = hx;
  if (y > hy) y = hy;
  if (z > hz) z = hz;
}

static inline int fsg_fill_deform(const fsg_deform* d, FsgDeformK& K) {
  if (!d) return FSG_E_BADARG;
  K.n0 = d->shape[0]; K.n1 = d->shape[1]; K.n2 = d->shape[2];
  if (K.n0 <= 0 || K.n1 <= 0 || K.n2 <= 0) return FSG_E_BADARG;
  if ((size_t)K.n0 * K.n1 * K.n2 > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  for (int t = 0; t < 9; ++t) K.A[t] = d->A[t];
  for (int t = 0; t < 3; ++t) { K.cen[t] = d->centre[t]; K.c2[t] = d->c2[t]; }
  K.flip = d->flip;
  K.f0 = d->field_dims[0]; K.f1 = d->field_dims[1]; K.f2 = d->field_dims[2];
  const bool has = K.f0 > 0 && K.f1 > 0 && K.f2 > 0;
  K.field = has ? d->field : nullptr;
  K.tx = d->tx; K.ty = d->ty; K.tz = d->tz;
  if (has && (!d->field || !d->tx || !d->ty || !d->tz)) return FSG_E_BADARG;
  K.rows = d->rows; K.row_stride = d->row_stride;
  if (K.rows && K.row_stride < 3 * K.f2) return FSG_E_BADARG;
  return 0;
}

// kernel-argument copy of fsg_epilogue
struct EpiK {
  float gamma;
  int b0, b1, b2;
  const float* bias;
  const fsg_tap* bx;
  const fsg_tap* by;
  const fsg_tap* bz;
};

static inline int fill_epilogue(const fsg_epilogue* e, EpiK& K) {
  K.gamma = 0.f; K.bias = nullptr; K.bx = K.by = K.bz = nullptr; K.b0 = K.b1 = K.b2 = 0;
  if (!e) return 0;
  K.gamma = e->gamma;
  if (e->bias) {
    if (!e->bx || !e->by || !e->bz) return FSG_E_BADARG;
    K.bias = e->bias; K.bx = e->bx; K.by = e->by; K.bz = e->bz;
    K.b0 = e->bias_dims[0]; K.b1 = e->bias_dims[1]; K.b2 = e->bias_dims[2];
    if (K.b0 <= 0 || K.b1 <= 0 || K.b2 <= 0) return FSG_E_BADARG;
  }
  return 0;
}

// floor(min) of the clamped coordinates per axis (affine_nonrigid.py:350-358), from the order keys of the reduction
struct Margins { float mx, my, mz; };

__device__ __forceinline__ Margins load_margins(const int32_t* mm6) {
  Margins m;
  m.mx = floorf(fsg_key2f(mm6[0]));
  m.my = floorf(fsg_key2f(mm6[1]));
  m.mz = floorf(fsg_key2f(mm6[2]));
  return m;
}

// logical tile id: hardware deals consecutive workgroups round-robin over the 8 XCDs; give every XCD a
// contiguous range of tiles (= a slab of x planes) so the source planes it gathers from stay in ITS L2.
// Speed only: any mapping gives the same result.
__device__ __forceinline__ int xcd_tile(int b, int nb) {
  return (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}

// fsg_warp_lean.hip: the lean fused warp (-> FSG_E_ALIGN when the configuration is outside its domain).
// label_in_bytes / label_out_bytes: 4 = float32, 1 = uint8.
int fsg_launch_warp_lean(const FsgDeformK& D, const EpiK& E, const int32_t* mm6, const float* src_lin, float* out_lin,
                         const void* src_nn, void* out_nn, int label_in_bytes, int label_out_bytes, bool fast,
                         void* stream);

// launch geometry: x = 64 lanes along z, y = 4 rows along y; grid (z chunks, y chunks, x)
static inline dim3 fsg_block3() { return dim3(64, 4, 1); }
static inline dim3 fsg_grid3(int n0, int n1, int n2) {
  return dim3((unsigned)((n2 + 63) / 64), (unsigned)((n1 + 3) / 4), (unsigned)n0);
}
