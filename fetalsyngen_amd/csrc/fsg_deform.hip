// fsg_deform.hip -- K2/K3/K4(/K5): deformation coordinates and the fused warp.
//
// Replaces, for one volume: `SpatialDeformation.generate_deformation_and_flip` +
// `apply_deformation_and_flip` (generator/deformation/affine_nonrigid.py:122-193, :299-366) and the
// `fast_3D_interp_torch` gathers (utils/generation.py:204-288) of the reference, plus optionally the
// RandGamma / RandBiasField pointwise stages (generator/augmentation/synthseg.py:274, :178-182).
//
// HBM layout: source and destination volumes (n0,n1,n2) fp32 (labels optionally uint8), z fastest.
// The 192 MiB coordinate volumes and the 192 MiB up-sampled field of the reference are never
// materialised: each thread re-evaluates the coarse field (<= 40 KB, L1/L2 resident) at its voxel.
#include "fsg_common.h"
#include "fsg_ride.h"

int g_tuning_flags = 0;  // FSG_TUNE_* bits, see fsg_set_tuning
int g_warp_variant = 0;   // fsg_warp_set_variant

namespace {

// ---- min/max of the clamped coordinates (affine_nonrigid.py:350-355) -------------------------
__global__ __launch_bounds__(256) void coords_minmax_kernel(FsgDeformK D, int32_t* __restrict__ mm6) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int rows = D.n0 * D.n1;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int i = r / D.n1, j = r - i * D.n1;
    for (int k = threadIdx.x; k < D.n2; k += blockDim.x) {
      float x, y, z;
      fsg_position(D, i, j, k, x, y, z);
      lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
      lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
      lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
    }
  }
  __shared__ float red[6][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = fsg_wave_min(lo[a]), h = fsg_wave_max(hi[a]);
    if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = red[a][0];
    for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, red[a][w]) : fmaxf(v, red[a][w]);
    // -0.0 must order below +0.0 like torch.min would see it only by sign; keys keep the sign bit
    if (a < 3) fsg_atomic_min_key(&mm6[a], v);
    else fsg_atomic_max_key(&mm6[a], v);
  }
}

// ---- materialised coordinates (public API of SpatialDeformation) ------------------------------
__global__ __launch_bounds__(256) void coords_kernel(FsgDeformK D, const int32_t* __restrict__ mm6,
                                                     float* __restrict__ xx, float* __restrict__ yy,
                                                     float* __restrict__ zz) {
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (k >= D.n2 || j >= D.n1) return;
  const Margins m = load_margins(mm6);
  float x, y, z;
  fsg_position(D, i, j, k, x, y, z);
  const size_t o = ((size_t)i * D.n1 + j) * D.n2 + k;
  xx[o] = x - m.mx;
  yy[o] = y - m.my;
  zz[o] = z - m.mz;
}

// ---- samplers ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T fetch(const T* __restrict__ s, const FsgDeformK& D, int x, int y, int z) {
  const int xs = D.flip ? (D.n0 - 1 - x) : x;  // torch.flip(.., [0]) folded into the index
  return s[((size_t)xs * D.n1 + y) * D.n2 + z];
}

__device__ __forceinline__ float sample_linear(const float* __restrict__ s, const FsgDeformK& D, float x,
                                               float y, float z) {
  const float hx = (float)(D.n0 - 1), hy = (float)(D.n1 - 1), hz = (float)(D.n2 - 1);
  const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= hx) && (y <= hy) && (z <= hz);
  if (!ok) return 0.f;
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const int x1 = min(x0 + 1, D.n0 - 1), y1 = min(y0 + 1, D.n1 - 1), z1 = min(z0 + 1, D.n2 - 1);
  const float bx = x - fx, by = y - fy, bz = z - fz;
  const float ax = 1.f - bx, ay = 1.f - by, az = 1.f - bz;
  const float c000 = fetch(s, D, x0, y0, z0), c100 = fetch(s, D, x1, y0, z0);
  const float c010 = fetch(s, D, x0, y1, z0), c110 = fetch(s, D, x1, y1, z0);
  const float c001 = fetch(s, D, x0, y0, z1), c101 = fetch(s, D, x1, y0, z1);
  const float c011 = fetch(s, D, x0, y1, z1), c111 = fetch(s, D, x1, y1, z1);
  const float c00 = c000 * ax + c100 * bx;
  const float c01 = c001 * ax + c101 * bx;
  const float c10 = c010 * ax + c110 * bx;
  const float c11 = c011 * ax + c111 * bx;
  const float c0 = c00 * ay + c10 * by;
  const float c1 = c01 * ay + c11 * by;
  return c0 * az + c1 * bz;
}

template <typename T>
__device__ __forceinline__ T sample_nearest(const T* __restrict__ s, const FsgDeformK& D, float x, float y,
                                            float z) {
  int xi = (int)rintf(x), yi = (int)rintf(y), zi = (int)rintf(z);  // round half to even
  xi = min(max(xi, 0), D.n0 - 1);
  yi = min(max(yi, 0), D.n1 - 1);
  zi = min(max(zi, 0), D.n2 - 1);
  return fetch(s, D, xi, yi, zi);
}

// =================================================================================================
// Row-wise kernels (the tuned path).  One wave owns one output row (i, j, all k) at a time:
//   1. the 64 lanes evaluate the x- and y-interpolation of the coarse displacement grid (and of the
//      coarse bias grid) ONCE per row -- 3*f2 + b2 values -- into a wave-private LDS slot;
//   2. every voxel of the row then needs only the z-interpolation: two LDS reads + one lerp per
//      channel, instead of 8 global loads + 7 lerps per channel.
// The arithmetic per voxel is the same sequence of fp32 operations as fsg_tab_interp (x, then y, then
// z; products then sum, no FMA), so positions are bit-identical to the per-voxel kernels.
// =================================================================================================
constexpr int ROWCAP = 512;  // floats of LDS per wave (3*f2 + b2 must fit; else per-voxel fallback)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x/y part of the separable interpolation for one row: dst[c*s2 + zs], c < NCH
template <int NCH>
__device__ __forceinline__ void fill_row_xy(const float* __restrict__ grid, int s1, int s2, const fsg_tap a,
                                            const fsg_tap b, float* dst, int lane) {
  const float* p00 = grid + (((size_t)a.lo * s1 + b.lo) * s2) * NCH;
  const float* p10 = grid + (((size_t)a.hi * s1 + b.lo) * s2) * NCH;
  const float* p01 = grid + (((size_t)a.lo * s1 + b.hi) * s2) * NCH;
  const float* p11 = grid + (((size_t)a.hi * s1 + b.hi) * s2) * NCH;
  for (int e = lane; e < NCH * s2; e += FSG_WAVE) {
    const int c = e / s2, zs = e - c * s2;
    const int o = zs * NCH + c;
    const float t0 = fsg_mix(a.w_lo, p00[o], a.w_hi, p10[o]);  // y = b.lo
    const float t1 = fsg_mix(a.w_lo, p01[o], a.w_hi, p11[o]);  // y = b.hi
    dst[e] = fsg_mix(b.w_lo, t0, b.w_hi, t1);
  }
}

__device__ __forceinline__ fsg_tap uniform_tap(const fsg_tap* t, int idx) {
  // idx is wave-uniform: broadcast through SGPRs so the 16-byte load is issued once
  const int4 v = *reinterpret_cast<const int4*>(t + idx);
  fsg_tap r;
  r.lo = __builtin_amdgcn_readfirstlane(v.x);
  r.hi = __builtin_amdgcn_readfirstlane(v.y);
  r.w_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(v.z));
  r.w_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(v.w));
  return r;
}

// position of voxel (i,j,k) from the row's LDS slot (same op order as fsg_position)
__device__ __forceinline__ void row_position(const FsgDeformK& D, const float* sm, int i, int j, int k,
                                             const fsg_tap& c, float& x, float& y, float& z) {
  float px = (float)i - D.cen[0], py = (float)j - D.cen[1], pz = (float)k - D.cen[2];
  if (D.field) {
    px = px + fsg_mix(c.w_lo, sm[c.lo], c.w_hi, sm[c.hi]);
    py = py + fsg_mix(c.w_lo, sm[D.f2 + c.lo], c.w_hi, sm[D.f2 + c.hi]);
    pz = pz + fsg_mix(c.w_lo, sm[2 * D.f2 + c.lo], c.w_hi, sm[2 * D.f2 + c.hi]);
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

typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));

// trilinear gather with the two z-neighbours fetched by one 8-byte load (needs n2 >= 2).
// Same blend order as sample_linear; when z0 == n2-1 the ceil neighbour is clamped onto z0 as in the
// reference (its weight is then exactly 0).
__device__ __forceinline__ float sample_linear_pairs(const float* __restrict__ s, const FsgDeformK& D, float x,
                                                     float y, float z) {
  const float hx = (float)(D.n0 - 1), hy = (float)(D.n1 - 1), hz = (float)(D.n2 - 1);
  const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= hx) && (y <= hy) && (z <= hz);
  if (!ok) return 0.f;
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  int x1 = min(x0 + 1, D.n0 - 1);
  const int y1 = min(y0 + 1, D.n1 - 1);
  const float bx = x - fx, by = y - fy, bz = z - fz;
  const float ax = 1.f - bx, ay = 1.f - by, az = 1.f - bz;
  int xs0 = x0, xs1 = x1;
  if (D.flip) { xs0 = D.n0 - 1 - x0; xs1 = D.n0 - 1 - x1; }
  const int zb = min(z0, D.n2 - 2);
  const bool hi0 = z0 != zb;  // only when z0 == n2-1: both neighbours are the pair's upper element
  const float* r00 = s + ((size_t)xs0 * D.n1 + y0) * D.n2 + zb;
  const float* r10 = s + ((size_t)xs1 * D.n1 + y0) * D.n2 + zb;
  const float* r01 = s + ((size_t)xs0 * D.n1 + y1) * D.n2 + zb;
  const float* r11 = s + ((size_t)xs1 * D.n1 + y1) * D.n2 + zb;
  const float2_u p00 = *reinterpret_cast<const float2_u*>(r00);
  const float2_u p10 = *reinterpret_cast<const float2_u*>(r10);
  const float2_u p01 = *reinterpret_cast<const float2_u*>(r01);
  const float2_u p11 = *reinterpret_cast<const float2_u*>(r11);
  const float c000 = hi0 ? p00.y : p00.x, c001 = p00.y;
  const float c100 = hi0 ? p10.y : p10.x, c101 = p10.y;
  const float c010 = hi0 ? p01.y : p01.x, c011 = p01.y;
  const float c110 = hi0 ? p11.y : p11.x, c111 = p11.y;
  const float c00 = c000 * ax + c100 * bx;
  const float c01 = c001 * ax + c101 * bx;
  const float c10 = c010 * ax + c110 * bx;
  const float c11 = c011 * ax + c111 * bx;
  const float c0 = c00 * ay + c10 * by;
  const float c1 = c01 * ay + c11 * by;
  return c0 * az + c1 * bz;
}

// logical tile id: hardware deals consecutive workgroups round-robin over the 8 XCDs; give every XCD a
// contiguous range of tiles (= a slab of x planes) so the source planes it gathers from stay in ITS L2.
// Speed only: any mapping gives the same result.

// ---- per-row coarse values, precomputed once per deformation ------------------------------------------
// rows[(i*n1 + j)*stride + e]: e < 3*f2 -> x/y-interpolated displacement (channel-major, then z index of
// the coarse grid); 3*f2 <= e < 3*f2 + b2 -> x/y-interpolated bias grid.  One thread per entry.  With this
// buffer the warp / min-max kernels start a row with ONE coalesced load instead of a dependent chain
// (table entry -> four coarse-grid loads -> LDS).
__device__ __forceinline__ void deform_rows_body(const FsgDeformK& D, const EpiK& E, float* __restrict__ rows, int stride,
                                                 int i, int t) {  // t = (j, e) of x plane i
  const int nf = D.field ? 3 * D.f2 : 0;
  const int need = nf + (E.bias ? E.b2 : 0);
  if (t >= D.n1 * need) return;
  const int j = t / need, e = t - j * need;
  float v;
  if (e < nf) {
    const fsg_tap a = D.tx[i], b = D.ty[j];
    const int c = e / D.f2, zs = e - c * D.f2;
    const int o = zs * 3 + c;
    const float* g = D.field;
    const float f00 = g[((a.lo * D.f1 + b.lo) * D.f2) * 3 + o];
    const float f10 = g[((a.hi * D.f1 + b.lo) * D.f2) * 3 + o];
    const float f01 = g[((a.lo * D.f1 + b.hi) * D.f2) * 3 + o];
    const float f11 = g[((a.hi * D.f1 + b.hi) * D.f2) * 3 + o];
    v = fsg_mix(b.w_lo, fsg_mix(a.w_lo, f00, a.w_hi, f10), b.w_hi, fsg_mix(a.w_lo, f01, a.w_hi, f11));
  } else {
    const fsg_tap a = E.bx[i], b = E.by[j];
    const int zs = e - nf;
    const float* g = E.bias;
    const float f00 = g[(a.lo * E.b1 + b.lo) * E.b2 + zs];
    const float f10 = g[(a.hi * E.b1 + b.lo) * E.b2 + zs];
    const float f01 = g[(a.lo * E.b1 + b.hi) * E.b2 + zs];
    const float f11 = g[(a.hi * E.b1 + b.hi) * E.b2 + zs];
    v = fsg_mix(b.w_lo, fsg_mix(a.w_lo, f00, a.w_hi, f10), b.w_hi, fsg_mix(a.w_lo, f01, a.w_hi, f11));
  }
  rows[((size_t)i * D.n1 + j) * stride + e] = v;
}

// Workgroup form of the same values.  The per-entry kernel is bound by dependent round trips, not by bytes: 11 000 workgroups
// each live for a tap load -> four gathers -> one store, and only 2 048 of them are resident at a time.  Here a workgroup
// owns one x plane and RB_J consecutive y rows: ONE phase of global loads (the x taps, the two coarse planes they select,
// the y taps of its rows), then every entry is a y-blend of two LDS values.  Blend order x then y as above: bit-identical.
constexpr int RB_J = 64;        // y rows per workgroup
constexpr int RB_SLAB_F = 2048; // floats: f1 * f2 * 3 of the coarse displacement grid
constexpr int RB_SLAB_B = 256;  // floats: b1 * b2 of the coarse bias grid
constexpr int RB_NEED = 512;    // entries per row

struct RowsSmem {
  float sf[RB_SLAB_F];
  float sb[RB_SLAB_B];
  int tab[RB_NEED];
  int4 ty[RB_J];
  int4 by[RB_J];
};

__host__ __device__ inline bool rows_block_fits(const FsgDeformK& D, const EpiK& E) {
  const int nf = D.field ? 3 * D.f2 : 0, nb = E.bias ? E.b2 : 0;
  return (D.field ? D.f1 * D.f2 * 3 : 0) <= RB_SLAB_F && (E.bias ? E.b1 * E.b2 : 0) <= RB_SLAB_B && nf + nb <= RB_NEED;
}

__device__ __forceinline__ void deform_rows_block(const FsgDeformK& D, const EpiK& E, float* __restrict__ rows, int stride,
                                                  int i, int j0, RowsSmem& S) {
  const int tid = threadIdx.x;
  const int nf = D.field ? 3 * D.f2 : 0, nb = E.bias ? E.b2 : 0, need = nf + nb;
  const int slab_f = D.field ? D.f1 * D.f2 * 3 : 0, slab_b = E.bias ? E.b1 * E.b2 : 0;
  const int jn = min(RB_J, D.n1 - j0);
  if (tid < jn) {
    if (D.field) S.ty[tid] = *reinterpret_cast<const int4*>(D.ty + j0 + tid);
    if (E.bias) S.by[tid] = *reinterpret_cast<const int4*>(E.by + j0 + tid);
  }
  if (D.field) {
    const fsg_tap a = uniform_tap(D.tx, i);
    const float* g0 = D.field + (size_t)a.lo * slab_f;
    const float* g1 = D.field + (size_t)a.hi * slab_f;
    for (int t = tid; t < slab_f; t += 256) S.sf[t] = fsg_mix(a.w_lo, g0[t], a.w_hi, g1[t]);
  }
  if (E.bias) {
    const fsg_tap a = uniform_tap(E.bx, i);
    const float* g0 = E.bias + (size_t)a.lo * slab_b;
    const float* g1 = E.bias + (size_t)a.hi * slab_b;
    for (int t = tid; t < slab_b; t += 256) S.sb[t] = fsg_mix(a.w_lo, g0[t], a.w_hi, g1[t]);
  }
  for (int e = tid; e < need; e += 256) {
    if (e < nf) {
      const int c = e / D.f2, zs = e - c * D.f2;
      S.tab[e] = zs * 3 + c;
    } else {
      S.tab[e] = e - nf;
    }
  }
  __syncthreads();
  const int rf = D.f2 * 3;
  for (int jj = tid >> 6; jj < jn; jj += 4) {
    float* dst = rows + ((size_t)i * D.n1 + j0 + jj) * stride;
    for (int e = tid & 63; e < need; e += 64) {
      const int o = S.tab[e];
      float v;
      if (e < nf) {
        const int4 b = S.ty[jj];
        v = fsg_mix(__builtin_bit_cast(float, b.z), S.sf[b.x * rf + o], __builtin_bit_cast(float, b.w), S.sf[b.y * rf + o]);
      } else {
        const int4 b = S.by[jj];
        v = fsg_mix(__builtin_bit_cast(float, b.z), S.sb[b.x * E.b2 + o], __builtin_bit_cast(float, b.w),
                    S.sb[b.y * E.b2 + o]);
      }
      dst[e] = v;
    }
  }
}

__global__ __launch_bounds__(256) void deform_rows_block_kernel(FsgDeformK D, EpiK E, float* __restrict__ rows, int stride) {
  __shared__ RowsSmem S;
  deform_rows_block(D, E, rows, stride, blockIdx.y, blockIdx.x * RB_J, S);
}

__global__ __launch_bounds__(256) void deform_rows_kernel(FsgDeformK D, EpiK E, float* __restrict__ rows, int stride) {
  deform_rows_body(D, E, rows, stride, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- floor(min) of the clamped coordinates with a boundary shortcut --------------------------------------
// Only floor(min) per axis is consumed downstream (margin subtraction, affine_nonrigid.py:350-358).  The
// coordinates are clamped to [0, n-1], so as soon as ONE voxel has a coordinate < 1 on an axis, that axis'
// floor(min) is 0 whatever the other voxels do.  In practice such voxels sit on the faces of the grid: pass 1
// evaluates the six faces only (2.3 % of a 256^3 grid); pass 2 (all voxels) runs only for the rare
// deformations where some axis is still unresolved -- every block of it first checks the three keys and
// exits when all are below key(1.0).  The result has the same floor as the exact minimum.
constexpr int FACE_STEP = 4;
__host__ __device__ inline int faces_samples(int n) { return (n - 1 + FACE_STEP - 1) / FACE_STEP + 1; }  // 0, 4, .., n-1
__host__ __device__ inline int faces_total(int n0, int n1, int n2) {
  const int m0 = faces_samples(n0), m1 = faces_samples(n1), m2 = faces_samples(n2);
  return 2 * (m1 * m2 + m0 * m2 + m0 * m1);
}

__device__ __forceinline__ void coords_faces_min_body(const FsgDeformK& D, int32_t* __restrict__ mm3, float (*red)[4],
                                                      int blk, int nblk) {
  // every FACE_STEP-th voxel of each face in both directions, the last index always included (corners and edges are in):
  // ANY subset of voxels is a valid first pass -- a coordinate < 1 found settles its axis, an axis left open sends the
  // sample through the exact full pass
  const int n0 = D.n0, n1 = D.n1, n2 = D.n2;
  const int m0 = faces_samples(n0), m1 = faces_samples(n1), m2 = faces_samples(n2);
  const int fa = m1 * m2, fb = m0 * m2, fc = m0 * m1;
  const int total = 2 * (fa + fb + fc);
  float lo[3] = {INFINITY, INFINITY, INFINITY};
  for (int t = blk * blockDim.x + threadIdx.x; t < total; t += nblk * blockDim.x) {
    int i, j, k, u = t;
    if (u < 2 * fa) {
      i = (u >= fa) ? n0 - 1 : 0; u %= fa; j = u / m2; k = u - j * m2;
      j = min(j * FACE_STEP, n1 - 1); k = min(k * FACE_STEP, n2 - 1);
    } else if ((u -= 2 * fa) < 2 * fb) {
      j = (u >= fb) ? n1 - 1 : 0; u %= fb; i = u / m2; k = u - i * m2;
      i = min(i * FACE_STEP, n0 - 1); k = min(k * FACE_STEP, n2 - 1);
    } else {
      u -= 2 * fb; k = (u >= fc) ? n2 - 1 : 0; u %= fc; i = u / m1; j = u - i * m1;
      i = min(i * FACE_STEP, n0 - 1); j = min(j * FACE_STEP, n1 - 1);
    }
    float x, y, z;
    fsg_position(D, i, j, k, x, y, z);
    lo[0] = fminf(lo[0], x);
    lo[1] = fminf(lo[1], y);
    lo[2] = fminf(lo[2], z);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float l = fsg_wave_min(lo[a]);
    if (lane == 0) red[a][wave] = l;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    float v = red[a][0];
    for (int w = 1; w < 4; ++w) v = fminf(v, red[a][w]);
    fsg_atomic_min_key(&mm3[a], v);
  }
}

__global__ __launch_bounds__(256) void coords_faces_min_kernel(FsgDeformK D, int32_t* __restrict__ mm3) {
  __shared__ float red[3][4];
  coords_faces_min_body(D, mm3, red, blockIdx.x, gridDim.x);
}

// ---- head of a sample as ONE launch ----------------------------------------------------------------------------
// The GMM draw (K1), the per-row coarse values and the six-face minimum do not depend on one another, and the two small ones
// cost a launch each mostly for the launch itself (~6 us of dispatch + end-of-kernel write-back against 2-4 us of work).
// Workgroups [0, nfaces) reduce the faces, [nfaces, nfaces + nrows) fill the rows, the rest draw the intensities (VALU-bound,
// so the small latency-bound work hides beside it).  The min/max keys must already be initialised (they arrive with the
// parameter arena): no workgroup of this launch may reset what another one reduces into.
struct HeadGmm {
  const uint8_t* l0; const uint8_t* l1; const uint8_t* l2; const uint8_t* l3;
  size_t n;
  const float* mus; const float* sigmas; int ntab;
  const float* noise; uint64_t seed, stream_id;
  float* out;
  // codes mode (ntuples > 0): l0 = the uint16 code volume, l1 = the tuples, sel = the four selected bytes of a row (8 bits each)
  int ntuples, stride;
  uint32_t sel;
};

// 8 waves/SIMD (<= 64 VGPRs) as the stand-alone GMM kernel has: the face job alone would take 73 and cap the launch at 6
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void sample_head_kernel(HeadGmm G, FsgDeformK D, EpiK E, float* __restrict__ rows,
                                                          int stride, int rows_gx, int nrows, int nfaces,
                                                          int32_t* __restrict__ mm3) {
  __shared__ float s_mu[256], s_sg[256];
  __shared__ float red[3][4];
  union HeadSmem {  // a workgroup does one job: the rows job's block and the codes mode's (mu, sigma) table share the space
    RowsSmem rows;
    float2 code_ms[FSG_CODES_MAX];
  };
  __shared__ HeadSmem U;
  RowsSmem& S = U.rows;
  int b = blockIdx.x;  // workgroup-uniform branches: a workgroup does exactly one of the three jobs
  if (b < nfaces) {  // the longest dependent chain first
    coords_faces_min_body(D, mm3, red, b, nfaces);
    return;
  }
  b -= nfaces;
  if (b < nrows) {
    deform_rows_block(D, E, rows, stride, b / rows_gx, (b % rows_gx) * RB_J, S);
    return;
  }
  b -= nrows;
  if (G.ntuples > 0) {  // codes mode (fsg_ride.h)
    const fsg_ride::GmmCodesK C{G.l0, G.l1, G.ntuples, G.stride, G.sel, G.n, G.mus, G.sigmas, G.ntab, G.seed, G.stream_id, G.out};
    fsg_ride::gmm_codes_job(C, U.code_ms, (unsigned)b, gridDim.x - (unsigned)(nrows + nfaces));
    return;
  }
  for (int t = threadIdx.x; t < 256; t += blockDim.x) {
    s_mu[t] = t < G.ntab ? G.mus[t] : 0.f;
    s_sg[t] = t < G.ntab ? G.sigmas[t] : 0.f;
  }
  __syncthreads();
  fsg_gmm_x4_loop(G.l0, G.l1, G.l2, G.l3, G.n, s_mu, s_sg, G.noise, G.seed, G.stream_id, G.out, (unsigned)b,
                  gridDim.x - (unsigned)(nrows + nfaces));
}

constexpr int WARP_ROWS_PER_WAVE = 4;

// Tuning knobs measured on MI355X (tools/kernel_bench.py, 256^3, rot 12 deg): batch 2 + skipping the gathers
// of voxels that sample outside the volume is the fastest of {1,2,4} x {skip, no skip} for the full kernel.
#ifndef FSG_WARP_BATCH
#define FSG_WARP_BATCH 2
#endif
#ifndef FSG_WARP_NO_SKIP_OUTSIDE
#define FSG_WARP_SKIP_OUTSIDE 1
#endif

template <typename LT, bool HAS_LIN, bool HAS_NN, bool FAST, int Q0, int NQ>
__device__ __forceinline__ void warp_emitN(const FsgDeformK& D, const EpiK& E, const Margins& m, const float* sm,
                                           int nf, int i, int j, int kbase, const fsg_tap (&ck)[4],
                                           const fsg_tap (&cbk)[4], size_t row, const float* __restrict__ src_lin,
                                           float* __restrict__ out_lin, const LT* __restrict__ src_nn,
                                           LT* __restrict__ out_nn) {
  const float hx = (float)(D.n0 - 1), hy = (float)(D.n1 - 1), hz = (float)(D.n2 - 1);
  const unsigned sx = (unsigned)D.n1 * (unsigned)D.n2, sy = (unsigned)D.n2;
  float x[NQ], y[NQ], z[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int k = min(kbase + 64 * (Q0 + q), D.n2 - 1);
    row_position(D, sm, i, j, k, ck[Q0 + q], x[q], y[q], z[q]);
    x[q] = x[q] - m.mx;
    y[q] = y[q] - m.my;
    z[q] = z[q] - m.mz;
  }
  LT nn[NQ];
  if (HAS_NN) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int xi = (int)rintf(x[q]), yi = (int)rintf(y[q]), zi = (int)rintf(z[q]);
      xi = min(max(xi, 0), D.n0 - 1);
      yi = min(max(yi, 0), D.n1 - 1);
      zi = min(max(zi, 0), D.n2 - 1);
      if (D.flip) xi = D.n0 - 1 - xi;
      nn[q] = src_nn[(unsigned)xi * sx + (unsigned)yi * sy + (unsigned)zi];
    }
  }
  float2_u p00[NQ], p10[NQ], p01[NQ], p11[NQ];
  float bx[NQ], by[NQ], bz[NQ];
  bool ok[NQ], hi0[NQ];
  if (HAS_LIN) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      ok[q] = (x[q] > 0.f) && (y[q] > 0.f) && (z[q] > 0.f) && (x[q] <= hx) && (y[q] <= hy) && (z[q] <= hz);
      const float fx = floorf(x[q]), fy = floorf(y[q]), fz = floorf(z[q]);
      bx[q] = x[q] - fx;
      by[q] = y[q] - fy;
      bz[q] = z[q] - fz;
      int x0 = min(max((int)fx, 0), D.n0 - 1), y0 = min(max((int)fy, 0), D.n1 - 1), z0 = min(max((int)fz, 0), D.n2 - 1);
      int x1 = min(x0 + 1, D.n0 - 1);
      const int y1 = min(y0 + 1, D.n1 - 1);
      if (D.flip) { x0 = D.n0 - 1 - x0; x1 = D.n0 - 1 - x1; }
      const int zb = min(z0, D.n2 - 2);
      hi0[q] = z0 != zb;
      const unsigned o00 = (unsigned)x0 * sx + (unsigned)y0 * sy + (unsigned)zb;
      const unsigned o10 = (unsigned)x1 * sx + (unsigned)y0 * sy + (unsigned)zb;
      const unsigned o01 = (unsigned)x0 * sx + (unsigned)y1 * sy + (unsigned)zb;
      const unsigned o11 = (unsigned)x1 * sx + (unsigned)y1 * sy + (unsigned)zb;
#ifdef FSG_WARP_SKIP_OUTSIDE
      p00[q] = p10[q] = p01[q] = p11[q] = float2_u{0.f, 0.f};
      if (ok[q])  // voxels that sample outside the volume (clamped onto a 0-face) need no data
#endif
      {
        p00[q] = *reinterpret_cast<const float2_u*>(src_lin + o00);
        p10[q] = *reinterpret_cast<const float2_u*>(src_lin + o10);
        p01[q] = *reinterpret_cast<const float2_u*>(src_lin + o01);
        p11[q] = *reinterpret_cast<const float2_u*>(src_lin + o11);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int k = kbase + 64 * (Q0 + q);
    const bool live = k < D.n2;
    if (HAS_NN && live) out_nn[row + k] = nn[q];
    if (HAS_LIN) {
      const float ax = 1.f - bx[q], ay = 1.f - by[q], az = 1.f - bz[q];
      const float c000 = hi0[q] ? p00[q].y : p00[q].x, c001 = p00[q].y;
      const float c100 = hi0[q] ? p10[q].y : p10[q].x, c101 = p10[q].y;
      const float c010 = hi0[q] ? p01[q].y : p01[q].x, c011 = p01[q].y;
      const float c110 = hi0[q] ? p11[q].y : p11[q].x, c111 = p11[q].y;
      const float c00 = c000 * ax + c100 * bx[q];
      const float c01 = c001 * ax + c101 * bx[q];
      const float c10 = c010 * ax + c110 * bx[q];
      const float c11 = c011 * ax + c111 * bx[q];
      const float c0 = c00 * ay + c10 * by[q];
      const float c1 = c01 * ay + c11 * by[q];
      float v = ok[q] ? (c0 * az + c1 * bz[q]) : 0.f;
      if (E.gamma > 0.f) {
        // 300*(v/300)^g.  FAST: 300 * 2^(g*(log2 v - log2 300)) on v_log_f32 / v_exp_f32; else OCML powf and
        // an IEEE division like ATen.  (|FAST - precise| < 1e-4 on the 0..255 scale, tests/test_hip_parity.py)
        if (FAST) v = 300.0f * __builtin_amdgcn_exp2f(E.gamma * (__builtin_amdgcn_logf(v) - 8.2288186904958804f));
        else v = 300.0f * powf(v / 300.0f, E.gamma);
      }
      if (E.bias) {
        const fsg_tap cb = cbk[Q0 + q];
        const float bval = fsg_mix(cb.w_lo, sm[nf + cb.lo], cb.w_hi, sm[nf + cb.hi]);
        v = v * (FAST ? __builtin_amdgcn_exp2f(bval * 1.4426950408889634f) : expf(bval));
      }
      if (live) out_lin[row + k] = v;
    }
  }
}

template <typename LT, bool HAS_LIN, bool HAS_NN, bool FAST>
__device__ __forceinline__ void warp_emit4(const FsgDeformK& D, const EpiK& E, const Margins& m, const float* sm,
                                           int nf, int i, int j, int kbase, const fsg_tap (&ck)[4],
                                           const fsg_tap (&cbk)[4], size_t row, const float* __restrict__ src_lin,
                                           float* __restrict__ out_lin, const LT* __restrict__ src_nn,
                                           LT* __restrict__ out_nn) {
#define FSG_EMIT(Q0, NQ) \
  warp_emitN<LT, HAS_LIN, HAS_NN, FAST, Q0, NQ>(D, E, m, sm, nf, i, j, kbase, ck, cbk, row, src_lin, out_lin, src_nn, out_nn)
#if FSG_WARP_BATCH == 4
  FSG_EMIT(0, 4);
#elif FSG_WARP_BATCH == 2
  FSG_EMIT(0, 2);
  FSG_EMIT(2, 2);
#else
  FSG_EMIT(0, 1);
  FSG_EMIT(1, 1);
  FSG_EMIT(2, 1);
  FSG_EMIT(3, 1);
#endif
#undef FSG_EMIT
}

// stage one row's coarse values into the wave's LDS slot
__device__ __forceinline__ void stage_row(const FsgDeformK& D, const EpiK& E, int i, int j, int nf, int need,
                                          const fsg_tap& ax, const fsg_tap& abx, float* sm, int lane) {
  if (D.rows) {
    const float* r = D.rows + ((size_t)i * D.n1 + j) * D.row_stride;
    for (int e = lane; e < need; e += FSG_WAVE) sm[e] = r[e];
  } else {
    if (D.field) fill_row_xy<3>(D.field, D.f1, D.f2, ax, uniform_tap(D.ty, j), sm, lane);
    if (E.bias) fill_row_xy<1>(E.bias, E.b1, E.b2, abx, uniform_tap(E.by, j), sm + nf, lane);
  }
}

template <typename LT, bool HAS_LIN, bool HAS_NN, bool FAST>
__global__ __launch_bounds__(256) void warp_rows_kernel(FsgDeformK D, const int32_t* __restrict__ mm6,
                                                        const float* __restrict__ src_lin,
                                                        float* __restrict__ out_lin, const LT* __restrict__ src_nn,
                                                        LT* __restrict__ out_nn, EpiK E) {
  __shared__ float sm_all[4][2][ROWCAP];  // per wave, double buffered
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // Row -> wave mapping (FSG_WARP_MAP).  An L1-missing cache line costs ~10.7 cycles of the CU's fill path
  // (profiles/r01_gather_cost_ubench.txt), so the four waves of a block should gather from ADJACENT source
  // rows at the same time: 0 = wave w owns rows 4w..4w+3 (no sharing), 1 = wave w owns rows w, w+4, w+8,
  // w+12 (the block sweeps 4 adjacent rows per step), 2 = 2x2 patches of (x,y) rows.
#ifndef FSG_WARP_MAP
#define FSG_WARP_MAP 1
#endif
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
#if FSG_WARP_MAP == 2
  const int tiles_j = (D.n1 + 2 * WARP_ROWS_PER_WAVE - 1) / (2 * WARP_ROWS_PER_WAVE);
  const int i = min(2 * (tile / tiles_j) + (wave >> 1), D.n0 - 1);
  const bool dead_i = 2 * (tile / tiles_j) + (wave >> 1) >= D.n0;
  const int jbase = (tile % tiles_j) * (2 * WARP_ROWS_PER_WAVE) + (wave & 1);
  const int jstep = 2;
#else
  const int tiles_j = (D.n1 + 4 * WARP_ROWS_PER_WAVE - 1) / (4 * WARP_ROWS_PER_WAVE);
  const int i = tile / tiles_j;
  const bool dead_i = false;
#if FSG_WARP_MAP == 1
  const int jbase = (tile - i * tiles_j) * (4 * WARP_ROWS_PER_WAVE) + wave;
  const int jstep = 4;
#else
  const int jbase = (tile - i * tiles_j) * (4 * WARP_ROWS_PER_WAVE) + wave * WARP_ROWS_PER_WAVE;
  const int jstep = 1;
#endif
#endif
  const Margins m = load_margins(mm6);
  const int nf = D.field ? 3 * D.f2 : 0;
  const int need = nf + (E.bias ? E.b2 : 0);
  const fsg_tap none = fsg_tap{0, 0, 0.f, 0.f};
  const bool onfly = D.rows == nullptr;
  const fsg_tap ax = (onfly && D.field) ? uniform_tap(D.tx, i) : none;
  const fsg_tap abx = (onfly && E.bias) ? uniform_tap(E.bx, i) : none;
  // the z tables are the same for every row: keep this lane's entries in registers (n2 <= 256)
  const bool cached = D.n2 <= 256;
  fsg_tap ck[4], cbk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = min(lane + 64 * q, D.n2 - 1);
    ck[q] = D.field ? D.tz[k] : none;
    cbk[q] = E.bias ? E.bz[k] : none;
  }
  if (dead_i) return;
  if (jbase < D.n1 && need) stage_row(D, E, i, jbase, nf, need, ax, abx, sm_all[wave][0], lane);
  for (int r = 0; r < WARP_ROWS_PER_WAVE; ++r) {
    const int j = jbase + r * jstep;
    if (j >= D.n1) break;
    const float* sm = sm_all[wave][r & 1];
    wave_lds_sync();
    // prefetch the next row's coarse values into the other buffer while this row's gathers are in flight
    if (r + 1 < WARP_ROWS_PER_WAVE && j + jstep < D.n1 && need)
      stage_row(D, E, i, j + jstep, nf, need, ax, abx, sm_all[wave][(r + 1) & 1], lane);
    const size_t row = ((size_t)i * D.n1 + j) * D.n2;
    if (cached) {
      warp_emit4<LT, HAS_LIN, HAS_NN, FAST>(D, E, m, sm, nf, i, j, lane, ck, cbk, row, src_lin, out_lin, src_nn,
                                            out_nn);
    } else {
      for (int kb = 0; kb < D.n2; kb += 256) {
        fsg_tap c4[4], b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = min(kb + lane + 64 * q, D.n2 - 1);
          c4[q] = D.field ? D.tz[k] : none;
          b4[q] = E.bias ? E.bz[k] : none;
        }
        warp_emit4<LT, HAS_LIN, HAS_NN, FAST>(D, E, m, sm, nf, i, j, kb + lane, c4, b4, row, src_lin, out_lin,
                                              src_nn, out_nn);
      }
    }
  }
}

// ---- lean per-voxel body on buffer addressing (patch kernel) ------------------------------------------------
// The patch kernel is instruction-issue bound (~290 wave-instructions per 64 voxels, one per quad-cycle per
// SIMD; profiles/r01_pmc_warp.md), so this body spends as few instructions as the arithmetic allows:
//   * raw buffer loads (SGPR descriptor + 32-bit byte offset): no 64-bit address arithmetic, and out-of-range
//     offsets return 0 instead of faulting, so no index is ever clamped (indices are in range by construction;
//     the clamp only guarded against NaN positions);
//   * ONE base offset per voxel; the 2x2x2 neighbours are base +- plane stride, + row stride, and the upper z
//     neighbour is the second half of each 8-byte load.  Where the reference clamps a neighbour onto the base
//     (coordinate exactly on the last plane/row/column) its weight is exactly 0, so whichever finite value (or
//     the 0 of an out-of-range read) sits there contributes +-0: same result.  Only the z edge is re-read with
//     4-byte loads, because the 8-byte load of the buffer's very last element would straddle its end.
struct WarpBuf {
  __amdgpu_buffer_rsrc_t lin, nn;
  unsigned sx, sy;   // element strides of x and y
  int dx_bytes;      // +-plane stride in bytes (sign: flip)
};

template <typename LT>
__device__ __forceinline__ LT buf_load_label(__amdgpu_buffer_rsrc_t r, unsigned elem);
template <>
__device__ __forceinline__ float buf_load_label<float>(__amdgpu_buffer_rsrc_t r, unsigned elem) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, elem * 4u, 0, 0));
}
template <>
__device__ __forceinline__ uint8_t buf_load_label<uint8_t>(__amdgpu_buffer_rsrc_t r, unsigned elem) {
  return __builtin_amdgcn_raw_buffer_load_b8(r, elem, 0, 0);
}

template <typename LT, bool HAS_LIN, bool HAS_NN, bool FAST>
__device__ __forceinline__ void warp_emit_buf(const FsgDeformK& D, const EpiK& E, const Margins& m, const WarpBuf& B,
                                              const float* sm, int nf, int i, int j, int k, bool live,
                                              const fsg_tap& c, const fsg_tap& cb, size_t row,
                                              float* __restrict__ out_lin, LT* __restrict__ out_nn) {
  typedef float f2v __attribute__((ext_vector_type(2)));
  float x, y, z;
  row_position(D, sm, i, j, k, c, x, y, z);
  x = x - m.mx;
  y = y - m.my;
  z = z - m.mz;
  if (HAS_NN) {
    int xi = (int)rintf(x);
    const int yi = (int)rintf(y), zi = (int)rintf(z);
    if (D.flip) xi = D.n0 - 1 - xi;
    const LT l = buf_load_label<LT>(B.nn, (unsigned)xi * B.sx + (unsigned)yi * B.sy + (unsigned)zi);
    if (live) out_nn[row + k] = l;
  }
  if (HAS_LIN) {
    const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f);  // x <= n-1 etc. hold by construction (clamped)
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    int x0 = (int)fx;
    const int y0 = (int)fy, z0 = (int)fz;
    const float bx = x - fx, by = y - fy, bz = z - fz;
    const float ax = 1.f - bx, ay = 1.f - by, az = 1.f - bz;
    if (D.flip) x0 = D.n0 - 1 - x0;
    const unsigned o = ((unsigned)x0 * B.sx + (unsigned)y0 * B.sy + (unsigned)z0) * 4u;
    const unsigned oy = B.sy * 4u;
    f2v p00 = {0.f, 0.f}, p10 = {0.f, 0.f}, p01 = {0.f, 0.f}, p11 = {0.f, 0.f};
    if (ok) {  // voxels that sample outside the volume (clamped onto a 0-face) need no data
      p00 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(B.lin, o, 0, 0));
      p10 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(B.lin, o + (unsigned)B.dx_bytes, 0, 0));
      p01 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(B.lin, o + oy, 0, 0));
      p11 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(B.lin, o + (unsigned)B.dx_bytes + oy, 0, 0));
    }
    if (ok && z0 >= D.n2 - 1) {  // rare (z exactly on the last column): single-element reads, upper neighbour unused
      p00.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(B.lin, o, 0, 0));
      p10.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(B.lin, o + (unsigned)B.dx_bytes, 0, 0));
      p01.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(B.lin, o + oy, 0, 0));
      p11.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(B.lin, o + (unsigned)B.dx_bytes + oy, 0, 0));
      p00.y = p00.x; p10.y = p10.x; p01.y = p01.x; p11.y = p11.x;
    }
    const float c00 = p00.x * ax + p10.x * bx;
    const float c01 = p00.y * ax + p10.y * bx;
    const float c10 = p01.x * ax + p11.x * bx;
    const float c11 = p01.y * ax + p11.y * bx;
    const float c0 = c00 * ay + c10 * by;
    const float c1 = c01 * ay + c11 * by;
    float v = ok ? (c0 * az + c1 * bz) : 0.f;
    if (E.gamma > 0.f) {
      if (FAST) v = 300.0f * __builtin_amdgcn_exp2f(E.gamma * (__builtin_amdgcn_logf(v) - 8.2288186904958804f));
      else v = 300.0f * powf(v / 300.0f, E.gamma);
    }
    if (E.bias) {
      const float bval = fsg_mix(cb.w_lo, sm[nf + cb.lo], cb.w_hi, sm[nf + cb.hi]);
      v = v * (FAST ? __builtin_amdgcn_exp2f(bval * 1.4426950408889634f) : expf(bval));
    }
    if (live) out_lin[row + k] = v;
  }
}

// ---- patch variant: 16 waves sweep a 4 x 4 patch of adjacent rows in lockstep ----------------------------
// Cost model (profiles/r01_gather_cost_ubench.txt): a gather costs ~10.7 cycles of the CU's fill path per
// 128-B line that misses L1, whatever its width.  Output rows (i..i+3, j..j+3) read source rows that
// overlap almost entirely (their 2x2 neighbourhoods interleave), so when the 16 waves of ONE workgroup
// advance through z together -- one 64-voxel chunk per wave per step, a barrier per step -- a line missed
// by one wave is an L1 hit for the others.  Same per-voxel arithmetic as the row kernel (warp_emitN).
constexpr int PATCH = 4;
constexpr int PATCH_ROWCAP = 128;
#ifndef FSG_PATCH_NQ
#define FSG_PATCH_NQ 1
#endif
constexpr int PATCH_NQ = FSG_PATCH_NQ;  // 64-voxel chunks per lane and lockstep step (2: 174 us against 135 us -- the L1 working set doubles)
constexpr int PATCH_TZCAP = 512;  // z extent whose taps are staged in LDS (2 x 8 KB)

template <typename LT, bool HAS_LIN, bool HAS_NN, bool FAST, bool BUF>
__global__ __launch_bounds__(1024, 8) void warp_patch_kernel(FsgDeformK D, const int32_t* __restrict__ mm6,
                                                          const float* __restrict__ src_lin,
                                                          float* __restrict__ out_lin,
                                                          const LT* __restrict__ src_nn, LT* __restrict__ out_nn,
                                                          EpiK E) {
  __shared__ float sm_all[PATCH * PATCH][PATCH_ROWCAP];
  // the z taps of the whole row (displacement and bias grids), staged once per workgroup: read from the tables inside
  // the chunk loop they put a dependent global round trip in front of every chunk's gathers
  __shared__ int4 s_tz[PATCH_TZCAP], s_bz[PATCH_TZCAP];
  const bool taps_lds = D.n2 <= PATCH_TZCAP;
  if (taps_lds) {
    for (int t = threadIdx.x; t < D.n2; t += 1024) {
      if (D.field) s_tz[t] = *reinterpret_cast<const int4*>(D.tz + t);
      if (E.bias) s_bz[t] = *reinterpret_cast<const int4*>(E.bz + t);
    }
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tiles_j = (D.n1 + PATCH - 1) / PATCH;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int i_raw = (tile / tiles_j) * PATCH + (wave >> 2), j_raw = (tile % tiles_j) * PATCH + (wave & 3);
  const bool live_row = i_raw < D.n0 && j_raw < D.n1;
  const int i = min(i_raw, D.n0 - 1), j = min(j_raw, D.n1 - 1);
  const Margins m = load_margins(mm6);
  const int nf = D.field ? 3 * D.f2 : 0;
  const int need = nf + (E.bias ? E.b2 : 0);
  const fsg_tap none = fsg_tap{0, 0, 0.f, 0.f};
  float* sm = sm_all[wave];
  if (need) {
    const bool onfly = D.rows == nullptr;
    const fsg_tap ax = (onfly && D.field) ? uniform_tap(D.tx, i) : none;
    const fsg_tap abx = (onfly && E.bias) ? uniform_tap(E.bx, i) : none;
    stage_row(D, E, i, j, nf, need, ax, abx, sm, lane);
  }
  __syncthreads();  // rows are wave-private, the z taps were written by every wave
  const size_t row = ((size_t)i * D.n1 + j) * D.n2;
  const unsigned nvox = (unsigned)D.n0 * (unsigned)D.n1 * (unsigned)D.n2;
  WarpBuf B;
  B.lin = __builtin_amdgcn_make_buffer_rsrc((void*)src_lin, 0, HAS_LIN ? nvox * 4u : 0u, 0x00020000);
  B.nn = __builtin_amdgcn_make_buffer_rsrc((void*)src_nn, 0, HAS_NN ? nvox * (unsigned)sizeof(LT) : 0u, 0x00020000);
  B.sy = (unsigned)D.n2;
  B.sx = (unsigned)D.n1 * (unsigned)D.n2;
  B.dx_bytes = D.flip ? -(int)(B.sx * 4u) : (int)(B.sx * 4u);
  auto tap_at = [&](const int4* lds, const fsg_tap* tab, bool on, int k) {
    // NB: written as `if`, not `cond ? table[k] : none`: the ternary makes hipcc scalarise the 16-byte entry
    // into eight branchy dword loads (+35 % kernel time, measured)
    fsg_tap c = none;
    if (on) {
      if (taps_lds) { const int4 v = lds[k]; c = fsg_tap{v.x, v.y, __builtin_bit_cast(float, v.z), __builtin_bit_cast(float, v.w)}; }
      else c = tab[k];
    }
    return c;
  };
  if (BUF) {
    for (int kb = 0; kb < D.n2; kb += FSG_WAVE) {
      const int kk = kb + lane;
      const int k = min(kk, D.n2 - 1);
      const fsg_tap c = tap_at(s_tz, D.tz, D.field != nullptr, k), cb = tap_at(s_bz, E.bz, E.bias != nullptr, k);
      warp_emit_buf<LT, HAS_LIN, HAS_NN, FAST>(D, E, m, B, sm, nf, i, j, k, live_row && kk < D.n2, c, cb, row, out_lin,
                                               out_nn);
      __syncthreads();  // keep the 16 waves on the same z chunk: bounded L1 working set, shared misses
    }
  } else {
    // PATCH_NQ chunks of 64 voxels per step and lane: all their gathers are in flight before the first blend
    for (int kb = 0; kb < D.n2; kb += PATCH_NQ * FSG_WAVE) {
      const int kk = kb + lane;
      fsg_tap ck[4] = {none, none, none, none}, cbk[4] = {none, none, none, none};
#pragma unroll
      for (int q = 0; q < PATCH_NQ; ++q) {
        const int k = min(kk + 64 * q, D.n2 - 1);
        ck[q] = tap_at(s_tz, D.tz, D.field != nullptr, k);
        cbk[q] = tap_at(s_bz, E.bz, E.bias != nullptr, k);
      }
      if (live_row)
        warp_emitN<LT, HAS_LIN, HAS_NN, FAST, 0, PATCH_NQ>(D, E, m, sm, nf, i, j, kk, ck, cbk, row, src_lin, out_lin, src_nn,
                                                           out_nn);
      __syncthreads();  // keep the 16 waves on the same z chunks: bounded L1 working set, shared misses
    }
  }
}

// ---- tile variant: the 16 waves sweep a PI x PJ patch of rows in lockstep, KZ voxels of every row per step ------
// The patch kernel above gives a wave 64 consecutive z voxels of ONE row: at the rotations the generator draws
// (<= 20 degrees about every axis) the source positions of such a run drift by up to 64 * sin(theta) = 13-20 source
// rows in x and in y, so one gather touches ~30 different 128-B lines and uses a few voxels of each.  Here a wave
// covers RJ = 64 / KZ adjacent rows x KZ voxels: the drift inside a step is KZ * sin(theta), the footprint of the
// workgroup's PI x PJ x KZ brick of outputs is a compact block of source rows that the next step (same rows, next
// KZ voxels) continues along the same cache lines.  Same per-voxel arithmetic (warp_emitN), bit-identical output.
template <typename LT, bool HAS_LIN, bool HAS_NN, bool FAST, int KZ, int WI>
__global__ __launch_bounds__(1024, 8) void warp_tile_kernel(FsgDeformK D, const int32_t* __restrict__ mm6,
                                                         const float* __restrict__ src_lin,
                                                         float* __restrict__ out_lin,
                                                         const LT* __restrict__ src_nn, LT* __restrict__ out_nn,
                                                         EpiK E) {
  constexpr int RJ = 64 / KZ;   // rows (along j) per wave
  constexpr int WJ = 16 / WI;   // waves along j
  constexpr int PI = WI, PJ = WJ * RJ;
  __shared__ float sm_all[PI * PJ][PATCH_ROWCAP];
  __shared__ int4 s_tz[PATCH_TZCAP], s_bz[PATCH_TZCAP];
  const bool taps_lds = D.n2 <= PATCH_TZCAP;
  if (taps_lds) {
    for (int t = threadIdx.x; t < D.n2; t += 1024) {
      if (D.field) s_tz[t] = *reinterpret_cast<const int4*>(D.tz + t);
      if (E.bias) s_bz[t] = *reinterpret_cast<const int4*>(E.bz + t);
    }
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tiles_j = (D.n1 + PJ - 1) / PJ;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int wi = wave / WJ, wj = wave - wi * WJ;
  const int i_raw = (tile / tiles_j) * PI + wi;
  const int j_wave = (tile % tiles_j) * PJ + wj * RJ;
  const int rj = lane / KZ, kz = lane - rj * KZ;
  const int j_raw = j_wave + rj;
  const bool live_row = i_raw < D.n0 && j_raw < D.n1;
  const int i = min(i_raw, D.n0 - 1), j = min(j_raw, D.n1 - 1);
  const Margins m = load_margins(mm6);
  const int nf = D.field ? 3 * D.f2 : 0;
  const int need = nf + (E.bias ? E.b2 : 0);
  const fsg_tap none = fsg_tap{0, 0, 0.f, 0.f};
  if (need) {
    const bool onfly = D.rows == nullptr;
    const fsg_tap ax = (onfly && D.field) ? uniform_tap(D.tx, i) : none;
    const fsg_tap abx = (onfly && E.bias) ? uniform_tap(E.bx, i) : none;
#pragma unroll
    for (int r = 0; r < RJ; ++r)  // every row of the wave is staged by all 64 lanes
      stage_row(D, E, i, min(j_wave + r, D.n1 - 1), nf, need, ax, abx, sm_all[wave * RJ + r], lane);
  }
  __syncthreads();
  const float* sm = sm_all[wave * RJ + rj];
  const size_t row = ((size_t)i * D.n1 + j) * D.n2;
  auto tap_at = [&](const int4* lds, const fsg_tap* tab, bool on, int k) {
    fsg_tap c = none;
    if (on) {
      if (taps_lds) { const int4 v = lds[k]; c = fsg_tap{v.x, v.y, __builtin_bit_cast(float, v.z), __builtin_bit_cast(float, v.w)}; }
      else c = tab[k];
    }
    return c;
  };
  for (int kb = 0; kb < D.n2; kb += KZ) {
    const int kk = kb + kz;
    const int k = min(kk, D.n2 - 1);
    fsg_tap ck[4] = {none, none, none, none}, cbk[4] = {none, none, none, none};
    ck[0] = tap_at(s_tz, D.tz, D.field != nullptr, k);
    cbk[0] = tap_at(s_bz, E.bz, E.bias != nullptr, k);
    // lanes past the end of the row or of the patch recompute a live voxel and do not store (kbase >= n2 disables stores)
    warp_emitN<LT, HAS_LIN, HAS_NN, FAST, 0, 1>(D, E, m, sm, nf, i, j, (live_row && kk < D.n2) ? kk : (D.n2 + kk), ck, cbk, row,
                                                src_lin, out_lin, src_nn, out_nn);
    __syncthreads();  // the 16 waves stay on the same z slab: the brick's source block is what L1 holds
  }
}

template <bool MIN_ONLY>
__device__ __forceinline__ void coords_minmax_rows_body(const FsgDeformK& D, int32_t* __restrict__ mm6, int rows_per_block, int blk);

template <bool MIN_ONLY>
__global__ __launch_bounds__(256) void coords_minmax_rows_kernel(FsgDeformK D, int32_t* __restrict__ mm6,
                                                                 int rows_per_block) {
  coords_minmax_rows_body<MIN_ONLY>(D, mm6, rows_per_block, (int)blockIdx.x);
}

// The conditional floor(min) pass with the NEXT sample's keyed draw job beside it (fsg_sample_plan::ride_draw): both are a few
// workgroups of latency, and as launches of their own each costs ~5 us of the step.  Workgroups [0, nrest): the pass;
// [nrest, nrest + draw workgroups): the draw (fsg_ride.h), which touches nothing of this sample.
__global__ __launch_bounds__(256) void floormin_ride_kernel(FsgDeformK D, int32_t* __restrict__ mm6, int rows_per_block, int nrest,
                                                            const fsg_ride::DrawK P) {
  if ((int)blockIdx.x < nrest) coords_minmax_rows_body<true>(D, mm6, rows_per_block, (int)blockIdx.x);
  else fsg_ride::keyed_draw_body(P, (int)blockIdx.x - nrest);
}

template <bool MIN_ONLY>
__device__ __forceinline__ void coords_minmax_rows_body(const FsgDeformK& D, int32_t* __restrict__ mm6, int rows_per_block, int blk) {
  if (MIN_ONLY) {  // floor(min) already known to be 0 on every axis: nothing to add (block-uniform exit)
    const int32_t one = fsg_f2key(1.0f);
    const int32_t k0 = __hip_atomic_load(&mm6[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t k1 = __hip_atomic_load(&mm6[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t k2 = __hip_atomic_load(&mm6[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k0 < one && k1 < one && k2 < one) return;
  }
  __shared__ float sm_all[4][ROWCAP];
  __shared__ float red[6][4];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* sm = sm_all[wave];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int rows = D.n0 * D.n1;
  const int r_begin = blk * rows_per_block;
  const int r_end = min(rows, r_begin + rows_per_block);
  const fsg_tap none = fsg_tap{0, 0, 0.f, 0.f};
  const bool cached = D.n2 <= 256;
  fsg_tap ck[4];
  if (cached) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = lane + 64 * q;
      ck[q] = (D.field && k < D.n2) ? D.tz[k] : none;
    }
  }
  for (int r = r_begin + wave; r < r_end; r += 4) {
    const int i = r / D.n1, j = r - i * D.n1;
    if (D.rows) {
      const float* rr = D.rows + (size_t)r * D.row_stride;
      for (int e = lane; e < 3 * D.f2; e += FSG_WAVE) sm[e] = rr[e];
    } else if (D.field) {
      fill_row_xy<3>(D.field, D.f1, D.f2, uniform_tap(D.tx, i), uniform_tap(D.ty, j), sm, lane);
    }
    wave_lds_sync();
    if (cached) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        if (k < D.n2) {
          float x, y, z;
          row_position(D, sm, i, j, k, ck[q], x, y, z);
          lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
          lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
          lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
        }
      }
    } else {
      for (int k = lane; k < D.n2; k += FSG_WAVE) {
        float x, y, z;
        row_position(D, sm, i, j, k, D.field ? D.tz[k] : none, x, y, z);
        lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
        lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
        lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
      }
    }
    wave_lds_sync();
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float l = fsg_wave_min(lo[a]), h = fsg_wave_max(hi[a]);
    if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = red[a][0];
    for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, red[a][w]) : fmaxf(v, red[a][w]);
    if (a < 3) fsg_atomic_min_key(&mm6[a], v);
    else if (!MIN_ONLY) fsg_atomic_max_key(&mm6[a], v);
  }
}

// ---- fused warp -------------------------------------------------------------------------------
template <typename LT>
__global__ __launch_bounds__(256) void warp_kernel(FsgDeformK D, const int32_t* __restrict__ mm6,
                                                   const float* __restrict__ src_lin, float* __restrict__ out_lin,
                                                   const LT* __restrict__ src_nn, LT* __restrict__ out_nn, EpiK E) {
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (k >= D.n2 || j >= D.n1) return;
  const Margins m = load_margins(mm6);
  float x, y, z;
  fsg_position(D, i, j, k, x, y, z);
  x = x - m.mx;
  y = y - m.my;
  z = z - m.mz;
  const size_t o = ((size_t)i * D.n1 + j) * D.n2 + k;
  if (src_nn) out_nn[o] = sample_nearest<LT>(src_nn, D, x, y, z);
  if (src_lin) {
    float v = sample_linear(src_lin, D, x, y, z);
    if (E.gamma > 0.f) v = 300.0f * powf(v / 300.0f, E.gamma);
    if (E.bias) {
      const float b = fsg_tab_interp<1>(E.bias, E.b1, E.b2, 0, E.bx[i], E.by[j], E.bz[k]);
      v = v * expf(b);
    }
    out_lin[o] = v;
  }
}

// =================================================================================================
// Brick warp kernel: LDS staging of the source bounding box.
//
// A block of 256 threads produces an 8 x 8 x 16 brick of the output grid (4 consecutive z per thread):
//   0. the brick's 8+8+16 table entries and the <= 4^3 coarse-grid nodes it touches go to LDS;
//   1. every thread evaluates the sampling position of its 4 voxels (reference operation order) and the
//      block reduces the integer bounding box of all 2x2x2 neighbourhoods (wave shuffles -> LDS atomics);
//   2. the bounding box of the intensity volume (and of the uint8 label volume) is copied to LDS with
//      coalesced 16-byte loads -- every source cache line is fetched once per brick instead of once per
//      gather instruction that touches it (profiles/r01_pmc_warp.md);
//   3. the trilinear neighbourhoods and the nearest label are read from LDS, blended in the reference's
//      order, gamma/bias applied, and 4 voxels are stored with one 16-byte store.
// Bricks whose box does not fit (extreme deformations) or whose coarse window exceeds 4 nodes per axis take
// the same arithmetic through direct global gathers.  Results are bit-identical to the other warp kernels.
// STATUS (r01): correct but not yet faster -- 255-265 us vs 175 us for the patch kernel at 256^3: with
// 48 KiB of LDS only 3 workgroups fit a CU and each brick is a chain of 2 dependent global round trips
// + 3 barriers, and an 8x8x16 brick's box is ~7x its own volume.  Opt-in via FSG_TUNE_BRICK; the plan in
// DESIGN.md section 7 (persistent workgroups that prefetch the next brick's box) builds on this kernel.
// =================================================================================================
constexpr int BRI = 8, BRJ = 8, BRK = 16;
constexpr int BR_CAP = 9216;   // floats of LDS for the intensity box (36 KiB)
constexpr int BR_W = 4;        // coarse nodes per axis held in LDS

struct BrickLds {
  float box[BR_CAP];
  uint32_t lab[BR_CAP / 4];
  float nodes_f[BR_W * BR_W * BR_W * 3];
  float nodes_b[BR_W * BR_W * BR_W];
  fsg_tap tap[32];   // [0,8) x, [8,16) y, [16,32) z of the displacement tables
  fsg_tap btap[32];  // same for the bias tables
  int bb[6];         // min x,y,z, max x,y,z of the neighbourhoods
};

__device__ __forceinline__ float sel4(const float (&u)[BR_W], int idx) {
  float r = u[0];
  r = idx == 1 ? u[1] : r;
  r = idx == 2 ? u[2] : r;
  r = idx == 3 ? u[3] : r;
  return r;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, FSG_WAVE));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, FSG_WAVE));
  return v;
}

// x/y interpolation of the LDS node window for the nodes zs = 0..nzw-1 of one channel
template <int NCH>
__device__ __forceinline__ void window_xy(const float* nodes, int nyw, int nzw, int ch, const fsg_tap& a, int alo,
                                          int ahi, const fsg_tap& b, int blo, int bhi, float (&u)[BR_W]) {
#pragma unroll
  for (int zs = 0; zs < BR_W; ++zs) {
    u[zs] = 0.f;
    if (zs < nzw) {
      const float f00 = nodes[((alo * nyw + blo) * nzw + zs) * NCH + ch];
      const float f10 = nodes[((ahi * nyw + blo) * nzw + zs) * NCH + ch];
      const float f01 = nodes[((alo * nyw + bhi) * nzw + zs) * NCH + ch];
      const float f11 = nodes[((ahi * nyw + bhi) * nzw + zs) * NCH + ch];
      u[zs] = fsg_mix(b.w_lo, fsg_mix(a.w_lo, f00, a.w_hi, f10), b.w_hi, fsg_mix(a.w_lo, f01, a.w_hi, f11));
    }
  }
}

__device__ __forceinline__ void affine_clamp(const FsgDeformK& D, float px, float py, float pz, float& x, float& y,
                                             float& z) {
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

template <typename LD, bool HAS_LIN, bool HAS_NN, bool FAST>
__global__ __launch_bounds__(256) void warp_brick_kernel(FsgDeformK D, const int32_t* __restrict__ mm6,
                                                         const float* __restrict__ src_lin,
                                                         float* __restrict__ out_lin,
                                                         const uint8_t* __restrict__ src_nn,
                                                         LD* __restrict__ out_nn, EpiK E) {
  __shared__ __attribute__((aligned(16))) BrickLds S;
  const int t = threadIdx.x;
  const int nbk = (D.n2 + BRK - 1) / BRK, nbj = (D.n1 + BRJ - 1) / BRJ;
  int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int kb = tile % nbk;
  tile /= nbk;
  const int jb = tile % nbj, ib = tile / nbj;
  const int i0 = ib * BRI, j0 = jb * BRJ, k0 = kb * BRK;
  const int k4 = t & 3, jj = (t >> 2) & 7, ii = t >> 5;
  const int i = min(i0 + ii, D.n0 - 1), j = min(j0 + jj, D.n1 - 1);
  const bool live_ij = (i0 + ii < D.n0) && (j0 + jj < D.n1);
  const Margins m = load_margins(mm6);

  // ---- 0. tables and coarse nodes of this brick -------------------------------------------------
  if (t < 32) {
    const fsg_tap none = fsg_tap{0, 0, 0.f, 0.f};
    fsg_tap a = none, b = none;
    if (t < 8) {
      const int q = min(i0 + t, D.n0 - 1);
      if (D.field) a = D.tx[q];
      if (E.bias) b = E.bx[q];
    } else if (t < 16) {
      const int q = min(j0 + t - 8, D.n1 - 1);
      if (D.field) a = D.ty[q];
      if (E.bias) b = E.by[q];
    } else {
      const int q = min(k0 + t - 16, D.n2 - 1);
      if (D.field) a = D.tz[q];
      if (E.bias) b = E.bz[q];
    }
    S.tap[t] = a;
    S.btap[t] = b;
  }
  if (t < 3) { S.bb[t] = 0x7FFFFFFF; S.bb[3 + t] = -1; }
  __syncthreads();
  // windows (tables are non-decreasing in the output index)
  const int xw0 = S.tap[0].lo, yw0 = S.tap[8].lo, zw0 = S.tap[16].lo;
  const int nxw = S.tap[7].hi - xw0 + 1, nyw = S.tap[15].hi - yw0 + 1, nzw = S.tap[31].hi - zw0 + 1;
  const int bxw0 = S.btap[0].lo, byw0 = S.btap[8].lo, bzw0 = S.btap[16].lo;
  const int nbxw = S.btap[7].hi - bxw0 + 1, nbyw = S.btap[15].hi - byw0 + 1, nbzw = S.btap[31].hi - bzw0 + 1;
  const bool win_ok = (!D.field || (nxw <= BR_W && nyw <= BR_W && nzw <= BR_W)) &&
                      (!E.bias || (nbxw <= BR_W && nbyw <= BR_W && nbzw <= BR_W));
  if (win_ok) {
    if (D.field) {
      const int nn3 = nxw * nyw * nzw * 3;
      if (t < nn3) {
        const int c = t % 3;
        int r = t / 3;
        const int zz = r % nzw;
        r /= nzw;
        const int yy = r % nyw, xx = r / nyw;
        S.nodes_f[t] = D.field[(((size_t)(xw0 + xx) * D.f1 + (yw0 + yy)) * D.f2 + (zw0 + zz)) * 3 + c];
      }
    }
    if (E.bias) {
      const int nb3 = nbxw * nbyw * nbzw;
      if (t >= 192 && t - 192 < nb3) {
        int r = t - 192;
        const int zz = r % nbzw;
        r /= nbzw;
        const int yy = r % nbyw, xx = r / nbyw;
        S.nodes_b[t - 192] = E.bias[((size_t)(bxw0 + xx) * E.b1 + (byw0 + yy)) * E.b2 + (bzw0 + zz)];
      }
    }
  }
  __syncthreads();

  // ---- 1. positions of this thread's 4 voxels ------------------------------------------------------
  float x[4], y[4], z[4], bval[4];
  {
    float ux[BR_W], uy[BR_W], uz[BR_W], ub[BR_W];
    const fsg_tap a = S.tap[ii], b = S.tap[8 + jj];
    if (D.field && win_ok) {
      window_xy<3>(S.nodes_f, nyw, nzw, 0, a, a.lo - xw0, a.hi - xw0, b, b.lo - yw0, b.hi - yw0, ux);
      window_xy<3>(S.nodes_f, nyw, nzw, 1, a, a.lo - xw0, a.hi - xw0, b, b.lo - yw0, b.hi - yw0, uy);
      window_xy<3>(S.nodes_f, nyw, nzw, 2, a, a.lo - xw0, a.hi - xw0, b, b.lo - yw0, b.hi - yw0, uz);
    }
    if (E.bias && win_ok) {
      const fsg_tap ab = S.btap[ii], bb = S.btap[8 + jj];
      window_xy<1>(S.nodes_b, nbyw, nbzw, 0, ab, ab.lo - bxw0, ab.hi - bxw0, bb, bb.lo - byw0, bb.hi - byw0, ub);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = 4 * k4 + q;
      const int k = min(k0 + kq, D.n2 - 1);
      bval[q] = 0.f;
      if (win_ok) {
        float px = (float)i - D.cen[0], py = (float)j - D.cen[1], pz = (float)k - D.cen[2];
        if (D.field) {
          const fsg_tap c = S.tap[16 + kq];
          const int cl = c.lo - zw0, ch = c.hi - zw0;
          px = px + fsg_mix(c.w_lo, sel4(ux, cl), c.w_hi, sel4(ux, ch));
          py = py + fsg_mix(c.w_lo, sel4(uy, cl), c.w_hi, sel4(uy, ch));
          pz = pz + fsg_mix(c.w_lo, sel4(uz, cl), c.w_hi, sel4(uz, ch));
        }
        affine_clamp(D, px, py, pz, x[q], y[q], z[q]);
        if (E.bias) {
          const fsg_tap cb = S.btap[16 + kq];
          bval[q] = fsg_mix(cb.w_lo, sel4(ub, cb.lo - bzw0), cb.w_hi, sel4(ub, cb.hi - bzw0));
        }
      } else {
        fsg_position(D, i, j, k, x[q], y[q], z[q]);
        if (E.bias) bval[q] = fsg_tab_interp<1>(E.bias, E.b1, E.b2, 0, E.bx[i], E.by[j], E.bz[k]);
      }
      x[q] = x[q] - m.mx;
      y[q] = y[q] - m.my;
      z[q] = z[q] - m.mz;
    }
  }

  // ---- 1b. bounding box of all neighbourhoods --------------------------------------------------------
  int lo0 = 0x7FFFFFFF, lo1 = 0x7FFFFFFF, lo2 = 0x7FFFFFFF, hi0 = -1, hi1 = -1, hi2 = -1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a0 = min(max((int)floorf(x[q]), 0), D.n0 - 1), a1 = min(max((int)floorf(y[q]), 0), D.n1 - 1),
              a2 = min(max((int)floorf(z[q]), 0), D.n2 - 1);
    lo0 = min(lo0, a0); hi0 = max(hi0, min(a0 + 1, D.n0 - 1));
    lo1 = min(lo1, a1); hi1 = max(hi1, min(a1 + 1, D.n1 - 1));
    lo2 = min(lo2, a2); hi2 = max(hi2, min(a2 + 1, D.n2 - 1));
  }
  lo0 = wave_min_i(lo0); lo1 = wave_min_i(lo1); lo2 = wave_min_i(lo2);
  hi0 = wave_max_i(hi0); hi1 = wave_max_i(hi1); hi2 = wave_max_i(hi2);
  if ((t & 63) == 0) {
    atomicMin(&S.bb[0], lo0); atomicMin(&S.bb[1], lo1); atomicMin(&S.bb[2], lo2);
    atomicMax(&S.bb[3], hi0); atomicMax(&S.bb[4], hi1); atomicMax(&S.bb[5], hi2);
  }
  __syncthreads();
  const int X0 = S.bb[0], Y0 = S.bb[1], Z0 = S.bb[2] & ~3;  // z start aligned to 16 bytes
  const int ex = S.bb[3] - X0 + 1, ey = S.bb[4] - Y0 + 1;
  const int ez4 = ((S.bb[5] - Z0) >> 2) + 1;                // 16-byte chunks per row
  const int pitch = ez4 * 4;
  const int nchunk = ex * ey * ez4;
  const bool fits = nchunk * 4 <= BR_CAP;

  // ---- 2. copy the box to LDS ----------------------------------------------------------------------------
  if (fits) {
    const float inv_e = 1.0f / (float)ez4, inv_y = 1.0f / (float)ey;
    for (int c = t; c < nchunk; c += 256) {
      int r = (int)((float)c * inv_e);
      if (r * ez4 > c) --r; else if ((r + 1) * ez4 <= c) ++r;
      const int q = c - r * ez4;
      int rx = (int)((float)r * inv_y);
      if (rx * ey > r) --rx; else if ((rx + 1) * ey <= r) ++rx;
      const int ry = r - rx * ey;
      int xs = X0 + rx;
      if (D.flip) xs = D.n0 - 1 - xs;
      const size_t g = ((size_t)xs * D.n1 + (Y0 + ry)) * D.n2 + Z0 + 4 * q;
      if (HAS_LIN) reinterpret_cast<float4*>(S.box)[c] = *reinterpret_cast<const float4*>(src_lin + g);
      if (HAS_NN) S.lab[c] = *reinterpret_cast<const uint32_t*>(src_nn + g);
    }
  }
  __syncthreads();

  // ---- 3. gather, blend, epilogue, store ------------------------------------------------------------------
  const float hx = (float)(D.n0 - 1), hy = (float)(D.n1 - 1), hz = (float)(D.n2 - 1);
  float v[4];
  uint32_t lab4 = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (HAS_NN) {
      int xi = (int)rintf(x[q]), yi = (int)rintf(y[q]), zi = (int)rintf(z[q]);
      xi = min(max(xi, 0), D.n0 - 1);
      yi = min(max(yi, 0), D.n1 - 1);
      zi = min(max(zi, 0), D.n2 - 1);
      uint32_t l;
      if (fits) {
        // (explicit LDS address space: with plain pointers the optimiser merges this branch and the global one below into ONE
        // load through a selected generic pointer -- a flat instruction on the vector-memory path for every "LDS" read)
        const int idx = ((xi - X0) * ey + (yi - Y0)) * pitch + (zi - Z0);
        l = (((const __attribute__((address_space(3))) uint32_t*)S.lab)[idx >> 2] >> (8 * (idx & 3))) & 255u;
      } else {
        const int xs = D.flip ? D.n0 - 1 - xi : xi;
        l = src_nn[((size_t)xs * D.n1 + yi) * D.n2 + zi];
      }
      lab4 |= l << (8 * q);
    }
    if (HAS_LIN) {
      const bool ok = (x[q] > 0.f) && (y[q] > 0.f) && (z[q] > 0.f) && (x[q] <= hx) && (y[q] <= hy) && (z[q] <= hz);
      float r = 0.f;
      if (ok) {
        const float fx = floorf(x[q]), fy = floorf(y[q]), fz = floorf(z[q]);
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
        const int dx = min(x0 + 1, D.n0 - 1) - x0, dy = min(y0 + 1, D.n1 - 1) - y0, dz = min(z0 + 1, D.n2 - 1) - z0;
        const float bx = x[q] - fx, by = y[q] - fy, bz = z[q] - fz;
        const float ax = 1.f - bx, ay = 1.f - by, az = 1.f - bz;
        float c000, c100, c010, c110, c001, c101, c011, c111;
        if (fits) {
          const int b00 = ((x0 - X0) * ey + (y0 - Y0)) * pitch + (z0 - Z0);
          const int sxl = dx * ey * pitch, syl = dy * pitch;
          const __attribute__((address_space(3))) float* B = (const __attribute__((address_space(3))) float*)S.box;
          c000 = B[b00];             c001 = B[b00 + dz];
          c100 = B[b00 + sxl];       c101 = B[b00 + sxl + dz];
          c010 = B[b00 + syl];       c011 = B[b00 + syl + dz];
          c110 = B[b00 + sxl + syl]; c111 = B[b00 + sxl + syl + dz];
        } else {
          int xs0 = x0, xs1 = x0 + dx;
          if (D.flip) { xs0 = D.n0 - 1 - xs0; xs1 = D.n0 - 1 - xs1; }
          const float* r00 = src_lin + ((size_t)xs0 * D.n1 + y0) * D.n2 + z0;
          const float* r10 = src_lin + ((size_t)xs1 * D.n1 + y0) * D.n2 + z0;
          const float* r01 = src_lin + ((size_t)xs0 * D.n1 + y0 + dy) * D.n2 + z0;
          const float* r11 = src_lin + ((size_t)xs1 * D.n1 + y0 + dy) * D.n2 + z0;
          c000 = r00[0]; c001 = r00[dz]; c100 = r10[0]; c101 = r10[dz];
          c010 = r01[0]; c011 = r01[dz]; c110 = r11[0]; c111 = r11[dz];
        }
        const float c00 = c000 * ax + c100 * bx;
        const float c01 = c001 * ax + c101 * bx;
        const float c10 = c010 * ax + c110 * bx;
        const float c11 = c011 * ax + c111 * bx;
        const float c0 = c00 * ay + c10 * by;
        const float c1 = c01 * ay + c11 * by;
        r = c0 * az + c1 * bz;
      }
      if (E.gamma > 0.f) {
        if (FAST) r = 300.0f * __builtin_amdgcn_exp2f(E.gamma * (__builtin_amdgcn_logf(r) - 8.2288186904958804f));
        else r = 300.0f * powf(r / 300.0f, E.gamma);
      }
      if (E.bias) r = r * (FAST ? __builtin_amdgcn_exp2f(bval[q] * 1.4426950408889634f) : expf(bval[q]));
      v[q] = r;
    }
  }
  const int k = k0 + 4 * k4;
  if (live_ij && k < D.n2) {  // n2 % 4 == 0: the four voxels are all inside or all outside
    const size_t o = ((size_t)i * D.n1 + j) * D.n2 + k;
    if (HAS_LIN) *reinterpret_cast<float4*>(out_lin + o) = make_float4(v[0], v[1], v[2], v[3]);
    if (HAS_NN) {
      if (sizeof(LD) == 1) {
        *reinterpret_cast<uint32_t*>(out_nn + o) = lab4;
      } else {
        *reinterpret_cast<float4*>(out_nn + o) = make_float4((float)(lab4 & 255u), (float)((lab4 >> 8) & 255u),
                                                             (float)((lab4 >> 16) & 255u), (float)(lab4 >> 24));
      }
    }
  }
}


template <typename LT>
int launch_warp(const fsg_deform* d, const int32_t* mm6, const float* src_lin, float* out_lin, const LT* src_nn,
                LT* out_nn, const fsg_epilogue* epi, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm6) return FSG_E_BADARG;
  if ((src_lin == nullptr) != (out_lin == nullptr)) return FSG_E_BADARG;
  if ((src_nn == nullptr) != (out_nn == nullptr)) return FSG_E_BADARG;
  if (!src_lin && !src_nn) return FSG_E_BADARG;
  if ((const void*)src_lin == (const void*)out_lin && src_lin) return FSG_E_BADARG;
  EpiK E;
  rc = fill_epilogue(epi, E);
  if (rc) return rc;
  const int need = (D.field ? 3 * D.f2 : 0) + (E.bias ? E.b2 : 0);
  if (D.rows && D.row_stride < need) return FSG_E_BADARG;
  if (g_warp_variant == 0 &&
      !(g_tuning_flags & (FSG_TUNE_GENERIC_WARP | FSG_TUNE_NO_PATCH | FSG_TUNE_BUFFER_LOADS | FSG_TUNE_NO_LEAN))) {
    // default: the lean body (fsg_warp_lean.hip); FSG_E_ALIGN = configuration outside its domain
    rc = fsg_launch_warp_lean(D, E, mm6, src_lin, out_lin, src_nn, out_nn, (int)sizeof(LT), (int)sizeof(LT),
                              !(g_tuning_flags & FSG_TUNE_PRECISE_MATH), stream);
    if (rc != FSG_E_ALIGN) return rc;
  }
  if (need <= PATCH_ROWCAP && D.n2 >= 2 && !(g_tuning_flags & (FSG_TUNE_GENERIC_WARP | FSG_TUNE_NO_PATCH))) {
    const int ntiles = ((D.n0 + PATCH - 1) / PATCH) * ((D.n1 + PATCH - 1) / PATCH);
    const bool fast = !(g_tuning_flags & FSG_TUNE_PRECISE_MATH);
    const dim3 grid((unsigned)ntiles), block(1024);
    hipStream_t st = fsg_stream(stream);
    const bool buf = (g_tuning_flags & FSG_TUNE_BUFFER_LOADS) != 0;
    if (g_warp_variant > 0) {
      // lockstep tile variants (fsg_warp_set_variant): 1 = 8x8 rows x 16 voxels, 2 = 4x8 rows x 32, 3 = 8x4 rows x 32, 4 = 4x16 rows x 16
#define FSG_LAUNCH_TILE(L, N, F, KZ, WI)                                                                                \
  do {                                                                                                                 \
    constexpr int PI_ = WI, PJ_ = (16 / WI) * (64 / KZ);                                                               \
    const dim3 g((unsigned)(((D.n0 + PI_ - 1) / PI_) * ((D.n1 + PJ_ - 1) / PJ_)));                                     \
    hipLaunchKernelGGL((warp_tile_kernel<LT, L, N, F, KZ, WI>), g, block, 0, st, D, mm6, src_lin, out_lin, src_nn,     \
                       out_nn, E);                                                                                     \
  } while (0)
#define FSG_LAUNCH_TILE_V(L, N, F)                      \
  do {                                                  \
    switch (g_warp_variant) {                           \
      case 1: FSG_LAUNCH_TILE(L, N, F, 16, 8); break;   \
      case 2: FSG_LAUNCH_TILE(L, N, F, 32, 4); break;   \
      case 3: FSG_LAUNCH_TILE(L, N, F, 32, 8); break;   \
      default: FSG_LAUNCH_TILE(L, N, F, 16, 4); break;  \
    }                                                   \
  } while (0)
      if (src_lin && src_nn) { if (fast) FSG_LAUNCH_TILE_V(true, true, true); else FSG_LAUNCH_TILE_V(true, true, false); }
      else if (src_lin)      { if (fast) FSG_LAUNCH_TILE_V(true, false, true); else FSG_LAUNCH_TILE_V(true, false, false); }
      else                   { FSG_LAUNCH_TILE_V(false, true, true); }
#undef FSG_LAUNCH_TILE_V
#undef FSG_LAUNCH_TILE
      FSG_RETURN_LAUNCH();
    }
#define FSG_LAUNCH_PATCH(L, N, F)                                                                                      \
  do {                                                                                                                 \
    if (buf)                                                                                                           \
      hipLaunchKernelGGL((warp_patch_kernel<LT, L, N, F, true>), grid, block, 0, st, D, mm6, src_lin, out_lin, src_nn, \
                         out_nn, E);                                                                                   \
    else                                                                                                               \
      hipLaunchKernelGGL((warp_patch_kernel<LT, L, N, F, false>), grid, block, 0, st, D, mm6, src_lin, out_lin,        \
                         src_nn, out_nn, E);                                                                           \
  } while (0)
    if (src_lin && src_nn) { if (fast) FSG_LAUNCH_PATCH(true, true, true); else FSG_LAUNCH_PATCH(true, true, false); }
    else if (src_lin)      { if (fast) FSG_LAUNCH_PATCH(true, false, true); else FSG_LAUNCH_PATCH(true, false, false); }
    else                   { FSG_LAUNCH_PATCH(false, true, true); }
#undef FSG_LAUNCH_PATCH
    FSG_RETURN_LAUNCH();
  }
  if (need <= ROWCAP && D.n2 >= 2 && !(g_tuning_flags & FSG_TUNE_GENERIC_WARP)) {
#if FSG_WARP_MAP == 2
    const int tiles_j = (D.n1 + 2 * WARP_ROWS_PER_WAVE - 1) / (2 * WARP_ROWS_PER_WAVE);
    const int ntiles = ((D.n0 + 1) / 2) * tiles_j;
#else
    const int tiles_j = (D.n1 + 4 * WARP_ROWS_PER_WAVE - 1) / (4 * WARP_ROWS_PER_WAVE);
    const int ntiles = D.n0 * tiles_j;
#endif
    const bool fast = !(g_tuning_flags & FSG_TUNE_PRECISE_MATH);
    const dim3 grid((unsigned)ntiles), block(256);
    hipStream_t st = fsg_stream(stream);
#define FSG_LAUNCH_WARP(L, N, F) \
  hipLaunchKernelGGL((warp_rows_kernel<LT, L, N, F>), grid, block, 0, st, D, mm6, src_lin, out_lin, src_nn, out_nn, E)
    if (src_lin && src_nn) { if (fast) FSG_LAUNCH_WARP(true, true, true); else FSG_LAUNCH_WARP(true, true, false); }
    else if (src_lin)      { if (fast) FSG_LAUNCH_WARP(true, false, true); else FSG_LAUNCH_WARP(true, false, false); }
    else                   { FSG_LAUNCH_WARP(false, true, true); }
#undef FSG_LAUNCH_WARP
    FSG_RETURN_LAUNCH();
  }
  hipLaunchKernelGGL(warp_kernel<LT>, fsg_grid3(D.n0, D.n1, D.n2), fsg_block3(), 0, fsg_stream(stream), D, mm6,
                     src_lin, out_lin, src_nn, out_nn, E);
  FSG_RETURN_LAUNCH();
}


// nearest-neighbour source as uint8 (label volumes), output uint8 or float32
template <typename LD>
int launch_warp_u8src(const fsg_deform* d, const int32_t* mm6, const float* src_lin, float* out_lin,
                      const uint8_t* src_nn, LD* out_nn, const fsg_epilogue* epi, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm6) return FSG_E_BADARG;
  if ((src_lin == nullptr) != (out_lin == nullptr)) return FSG_E_BADARG;
  if ((src_nn == nullptr) != (out_nn == nullptr)) return FSG_E_BADARG;
  if (!src_lin && !src_nn) return FSG_E_BADARG;
  EpiK E;
  rc = fill_epilogue(epi, E);
  if (rc) return rc;
  if (g_warp_variant == 0 && sizeof(LD) == 4 &&
      !(g_tuning_flags & (FSG_TUNE_GENERIC_WARP | FSG_TUNE_NO_PATCH | FSG_TUNE_BUFFER_LOADS | FSG_TUNE_NO_LEAN | FSG_TUNE_BRICK))) {
    // uint8 labels in, float32 labels out: served by the lean kernel (the uint8 -> uint8 case reaches it through launch_warp)
    rc = fsg_launch_warp_lean(D, E, mm6, src_lin, out_lin, src_nn, out_nn, 1, 4, !(g_tuning_flags & FSG_TUNE_PRECISE_MATH),
                              stream);
    if (rc != FSG_E_ALIGN) return rc;
  }
  const uintptr_t al16 = (uintptr_t)src_lin | (uintptr_t)out_lin | (sizeof(LD) == 4 ? (uintptr_t)out_nn : 0);
  const uintptr_t al4 = (uintptr_t)src_nn | (sizeof(LD) == 1 ? (uintptr_t)out_nn : 0);
  if ((D.n2 & 3) || (al16 & 15) || (al4 & 3) || D.n2 < 4 || !(g_tuning_flags & FSG_TUNE_BRICK)) return FSG_E_ALIGN;
  const int nb = ((D.n0 + BRI - 1) / BRI) * ((D.n1 + BRJ - 1) / BRJ) * ((D.n2 + BRK - 1) / BRK);
  const bool fast = !(g_tuning_flags & FSG_TUNE_PRECISE_MATH);
  const dim3 grid((unsigned)nb), block(256);
  hipStream_t st = fsg_stream(stream);
#define FSG_LAUNCH_BRICK(L, N, F) \
  hipLaunchKernelGGL((warp_brick_kernel<LD, L, N, F>), grid, block, 0, st, D, mm6, src_lin, out_lin, src_nn, out_nn, E)
  if (src_lin && src_nn) { if (fast) FSG_LAUNCH_BRICK(true, true, true); else FSG_LAUNCH_BRICK(true, true, false); }
  else if (src_lin)      { if (fast) FSG_LAUNCH_BRICK(true, false, true); else FSG_LAUNCH_BRICK(true, false, false); }
  else                   { FSG_LAUNCH_BRICK(false, true, true); }
#undef FSG_LAUNCH_BRICK
  FSG_RETURN_LAUNCH();
}

// ---- generic gather with explicit coordinates ---------------------------------------------------
__global__ __launch_bounds__(256) void interp_kernel(const float* __restrict__ src, int sx, int sy, int sz,
                                                     const float* __restrict__ ii, const float* __restrict__ jj,
                                                     const float* __restrict__ kk, size_t npts, int mode,
                                                     float defval, float* __restrict__ dst) {
  FsgDeformK D;
  D.n0 = sx; D.n1 = sy; D.n2 = sz; D.flip = 0;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npts; p += (size_t)gridDim.x * blockDim.x) {
    const float x = ii[p], y = jj[p], z = kk[p];
    if (mode == 1) {
      dst[p] = sample_nearest<float>(src, D, x, y, z);
    } else {
      const float hx = (float)(sx - 1), hy = (float)(sy - 1), hz = (float)(sz - 1);
      const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= hx) && (y <= hy) && (z <= hz);
      dst[p] = ok ? sample_linear(src, D, x, y, z) : defval;
    }
  }
}

__global__ void mm_init_kernel(int32_t* mm, int nmin, int nmax) {
  const int t = threadIdx.x;
  if (t < nmin) mm[t] = fsg_f2key(INFINITY);
  else if (t < nmin + nmax) mm[t] = fsg_f2key(-INFINITY);
}

}  // namespace

extern "C" {

int fsg_set_tuning(int flags) {
  const int prev = g_tuning_flags;
  g_tuning_flags = flags;
  return prev;
}

extern int g_lean_pace, g_lean_ablate;
int fsg_warp_set_variant(int variant) {
  const int prev = g_warp_variant;
  g_lean_ablate = 0;
#ifdef FSG_DIAG  // the ablation kernels (wrong results by construction) exist in a -DFSG_DIAG build only
  if (variant == 8 || variant == 9) { g_warp_variant = 0; g_lean_pace = 1; g_lean_ablate = variant - 7; }
#else
  if (variant == 8 || variant == 9) return FSG_E_BADARG;
#endif
  if (variant >= 0 && variant <= 4) { g_warp_variant = variant; g_lean_pace = -1; }
  if (variant >= 5 && variant <= 7) { g_warp_variant = 0; g_lean_pace = variant == 5 ? 0 : (variant == 6 ? 2 : 1); }
  return prev;
}

int fsg_minmax_init(int32_t* mm, int nmin, int nmax, void* stream) {
  if (!mm || nmin < 0 || nmax < 0 || nmin + nmax <= 0 || nmin + nmax > 64) return FSG_E_BADARG;
  hipLaunchKernelGGL(mm_init_kernel, dim3(1), dim3(64), 0, fsg_stream(stream), mm, nmin, nmax);
  FSG_RETURN_LAUNCH();
}

int fsg_deform_rows_f32(const fsg_deform* d, const fsg_epilogue* epi, float* rows, int row_stride, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  EpiK E;
  rc = fill_epilogue(epi, E);
  if (rc) return rc;
  const int need = (D.field ? 3 * D.f2 : 0) + (E.bias ? E.b2 : 0);
  if (!rows || row_stride < need) return FSG_E_BADARG;
  if (need == 0) return 0;
  if (D.n0 > 65535) return FSG_E_TOOBIG;
  if (rows_block_fits(D, E) && !(g_tuning_flags & FSG_TUNE_SPLIT_HEAD)) {
    const dim3 grid((unsigned)((D.n1 + RB_J - 1) / RB_J), (unsigned)D.n0);
    hipLaunchKernelGGL(deform_rows_block_kernel, grid, dim3(256), 0, fsg_stream(stream), D, E, rows, row_stride);
    FSG_RETURN_LAUNCH();
  }
  const dim3 grid((unsigned)((D.n1 * need + 255) / 256), (unsigned)D.n0);
  hipLaunchKernelGGL(deform_rows_kernel, grid, dim3(256), 0, fsg_stream(stream), D, E, rows, row_stride);
  FSG_RETURN_LAUNCH();
}

int fsg_coords_minmax_f32(const fsg_deform* d, int32_t* mm6, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm6) return FSG_E_BADARG;
  const int rows = D.n0 * D.n1;
  if ((D.field ? 3 * D.f2 : 0) <= ROWCAP && !(g_tuning_flags & FSG_TUNE_GENERIC_WARP)) {
    int grid = (rows + 7) / 8 < 2048 ? (rows + 7) / 8 : 2048;
    const int rpb = (rows + grid - 1) / grid;
    grid = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(coords_minmax_rows_kernel<false>, dim3(grid), dim3(256), 0, fsg_stream(stream), D, mm6, rpb);
    FSG_RETURN_LAUNCH();
  }
  const int grid = rows < 2048 ? rows : 2048;
  hipLaunchKernelGGL(coords_minmax_kernel, dim3(grid), dim3(256), 0, fsg_stream(stream), D, mm6);
  FSG_RETURN_LAUNCH();
}

int fsg_coords_floormin_f32(const fsg_deform* d, int32_t* mm3, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm3) return FSG_E_BADARG;
  hipStream_t st = fsg_stream(stream);
  const int faces = faces_total(D.n0, D.n1, D.n2);
  int g1 = (faces + 255) / 256;
  if (g1 > 1024) g1 = 1024;
  hipLaunchKernelGGL(coords_faces_min_kernel, dim3(g1), dim3(256), 0, st, D, mm3);
  const int rows = D.n0 * D.n1;
  if ((D.field ? 3 * D.f2 : 0) <= ROWCAP) {
    // 512 workgroups: in the usual case every one of them exits on its first instruction, and fewer of them
    // exit sooner; the rare full pass runs at a quarter of the parallelism, which is fine
    int grid = (rows + 7) / 8 < 512 ? (rows + 7) / 8 : 512;
    const int rpb = (rows + grid - 1) / grid;
    grid = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(coords_minmax_rows_kernel<true>, dim3(grid), dim3(256), 0, st, D, mm3, rpb);
  } else {
    // very large coarse grids: exact per-voxel pass (writes the maxima after the three minima too)
    return FSG_E_TOOBIG;
  }
  FSG_RETURN_LAUNCH();
}

int fsg_coords_floormin_rest_f32(const fsg_deform* d, int32_t* mm3, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm3) return FSG_E_BADARG;
  if ((D.field ? 3 * D.f2 : 0) > ROWCAP) return FSG_E_TOOBIG;
  const int rows = D.n0 * D.n1;
  int grid = (rows + 7) / 8 < 512 ? (rows + 7) / 8 : 512;
  const int rpb = (rows + grid - 1) / grid;
  grid = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(coords_minmax_rows_kernel<true>, dim3(grid), dim3(256), 0, fsg_stream(stream), D, mm3, rpb);
  FSG_RETURN_LAUNCH();
}

static int launch_sample_head(const HeadGmm& G, size_t n, const fsg_deform* d, const fsg_epilogue* epi, float* rows,
                              int row_stride, int32_t* mm3, void* stream);

// ---- internal entry points of the look-ahead (fsg_pipeline.cpp; not part of include/fsg_hip.h) ------------------------------
// the conditional floor(min) pass + the next sample's draw job (drawk: a fsg_ride::DrawK on the host)
extern "C" int fsg_internal_floormin_rest_ride(const fsg_deform* d, int32_t* mm3, const void* drawk, unsigned draw_blocks, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm3 || !drawk || draw_blocks == 0 || draw_blocks > 4096) return FSG_E_BADARG;
  if ((D.field ? 3 * D.f2 : 0) > ROWCAP) return FSG_E_TOOBIG;
  const int rows = D.n0 * D.n1;
  int grid = (rows + 7) / 8 < 512 ? (rows + 7) / 8 : 512;
  const int rpb = (rows + grid - 1) / grid;
  grid = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(floormin_ride_kernel, dim3((unsigned)grid + draw_blocks), dim3(256), 0, fsg_stream(stream), D, mm3, rpb, grid,
                     *reinterpret_cast<const fsg_ride::DrawK*>(drawk));
  FSG_RETURN_LAUNCH();
}

int fsg_sample_head_f32(const uint8_t* l0, const uint8_t* l1, const uint8_t* l2, const uint8_t* l3, size_t n,
                        const float* mus, const float* sigmas, int ntab, const float* noise, uint64_t seed,
                        uint64_t stream_id, float* out, const fsg_deform* d, const fsg_epilogue* epi, float* rows,
                        int row_stride, int32_t* mm3, void* stream) {
  if (n == 0 || !l0 || !mus || !sigmas || !out || ntab <= 0 || ntab > 256 || !mm3) return FSG_E_BADARG;
  const uintptr_t al = (uintptr_t)l0 | (uintptr_t)l1 | (uintptr_t)l2 | (uintptr_t)l3;
  if ((al & 3) || ((uintptr_t)out & 15)) return FSG_E_ALIGN;
  HeadGmm G{l0, l1, l2, l3, n, mus, sigmas, ntab, noise, seed, stream_id, out, 0, 0, 0u};
  return launch_sample_head(G, n, d, epi, rows, row_stride, mm3, stream);
}

int fsg_sample_head_codes_f32(const uint16_t* codes, const uint8_t* tuples, int ntuples, int stride, const int32_t sel[4],
                              size_t n, const float* mus, const float* sigmas, int ntab, uint64_t seed, uint64_t stream_id,
                              float* out, const fsg_deform* d, const fsg_epilogue* epi, float* rows, int row_stride,
                              int32_t* mm3, void* stream) {
  if (n == 0 || !codes || !tuples || !sel || !mus || !sigmas || !out || ntab <= 0 || ntab > 256 || !mm3) return FSG_E_BADARG;
  if (ntuples <= 0 || stride <= 0 || stride > 256) return FSG_E_BADARG;
  for (int m = 0; m < 4; ++m)
    if (sel[m] < 0 || sel[m] >= stride) return FSG_E_BADARG;
  if (ntuples > FSG_CODES_MAX || n > ((size_t)1 << 30)) return FSG_E_TOOBIG;
  if ((n & 7) || ((uintptr_t)codes & 15) || ((uintptr_t)out & 15)) return FSG_E_ALIGN;
  HeadGmm G{reinterpret_cast<const uint8_t*>(codes), tuples, nullptr, nullptr, n, mus, sigmas, ntab, nullptr, seed, stream_id, out,
            ntuples, stride, (uint32_t)sel[0] | ((uint32_t)sel[1] << 8) | ((uint32_t)sel[2] << 16) | ((uint32_t)sel[3] << 24)};
  return launch_sample_head(G, n, d, epi, rows, row_stride, mm3, stream);
}

static int launch_sample_head(const HeadGmm& G, size_t n, const fsg_deform* d, const fsg_epilogue* epi, float* rows,
                              int row_stride, int32_t* mm3, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  EpiK E;
  rc = fill_epilogue(epi, E);
  if (rc) return rc;
  const int need = (D.field ? 3 * D.f2 : 0) + (E.bias ? E.b2 : 0);
  if (need > 0 && (!rows || row_stride < need)) return FSG_E_BADARG;
  if ((size_t)D.n0 * D.n1 * D.n2 != n) return FSG_E_BADARG;
  if ((D.field ? 3 * D.f2 : 0) > ROWCAP) return FSG_E_TOOBIG;
  if (!rows_block_fits(D, E)) return FSG_E_TOOBIG;
  const int rows_gx = need > 0 ? (D.n1 + RB_J - 1) / RB_J : 0;
  const long long nrows = (long long)rows_gx * D.n0;
  const int faces = faces_total(D.n0, D.n1, D.n2);
  int nfaces = (faces + 255) / 256;
  if (nfaces > 1024) nfaces = 1024;
  size_t ngmm = ((n + 3) / 4 + 255) / 256;
  if (ngmm > 4096) ngmm = 4096;  // 4 groups per thread: 33.8 us (8 192: 34.8, 16 384: 39.0); codes mode, 2 x 8 voxels per thread: 30.5 us
                                 // (1 024: 34.7, 2 048: 37.0, 8 192: 33.6)
  if (nrows > 1000000) return FSG_E_TOOBIG;
  hipLaunchKernelGGL(sample_head_kernel, dim3((unsigned)(nrows + nfaces + ngmm)), dim3(256), 0, fsg_stream(stream), G, D,
                     E, rows, row_stride, rows_gx > 0 ? rows_gx : 1, (int)nrows, nfaces, mm3);
  FSG_RETURN_LAUNCH();
}

int fsg_coords_f32(const fsg_deform* d, const int32_t* mm6, float* xx, float* yy, float* zz, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm6 || !xx || !yy || !zz) return FSG_E_BADARG;
  hipLaunchKernelGGL(coords_kernel, fsg_grid3(D.n0, D.n1, D.n2), fsg_block3(), 0, fsg_stream(stream), D, mm6, xx,
                     yy, zz);
  FSG_RETURN_LAUNCH();
}

int fsg_warp_f32(const fsg_deform* d, const int32_t* mm6, const float* src_lin, float* out_lin,
                 const float* src_nn, float* out_nn, const fsg_epilogue* epi, void* stream) {
  return launch_warp<float>(d, mm6, src_lin, out_lin, src_nn, out_nn, epi, stream);
}

int fsg_warp_f32_u8(const fsg_deform* d, const int32_t* mm6, const float* src_lin, float* out_lin,
                    const uint8_t* src_nn, uint8_t* out_nn, const fsg_epilogue* epi, void* stream) {
  const int rc = launch_warp_u8src<uint8_t>(d, mm6, src_lin, out_lin, src_nn, out_nn, epi, stream);
  if (rc != FSG_E_ALIGN) return rc;  // brick kernel not selected / not applicable: patch or row kernel
  return launch_warp<uint8_t>(d, mm6, src_lin, out_lin, src_nn, out_nn, epi, stream);
}

int fsg_warp_f32_u8_to_f32(const fsg_deform* d, const int32_t* mm6, const float* src_lin, float* out_lin,
                           const uint8_t* src_nn, float* out_nn, const fsg_epilogue* epi, void* stream) {
  return launch_warp_u8src<float>(d, mm6, src_lin, out_lin, src_nn, out_nn, epi, stream);
}

int fsg_interp3d_f32(const float* src, int sx, int sy, int sz, const float* ii, const float* jj, const float* kk,
                     size_t npts, int mode, float defval, float* dst, void* stream) {
  if (!src || !ii || !jj || !kk || !dst || sx <= 0 || sy <= 0 || sz <= 0) return FSG_E_BADARG;
  if (mode != 0 && mode != 1) return FSG_E_BADARG;
  if ((size_t)sx * sy * sz > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  if (npts == 0) return 0;
  size_t blocks = (npts + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(interp_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), src, sx, sy, sz, ii,
                     jj, kk, npts, mode, defval, dst);
  FSG_RETURN_LAUNCH();
}

}  // extern "C"
