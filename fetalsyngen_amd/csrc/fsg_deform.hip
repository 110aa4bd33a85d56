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

namespace {

struct Margins { float mx, my, mz; };

__device__ __forceinline__ Margins load_margins(const int32_t* mm6) {
  Margins m;
  m.mx = floorf(fsg_key2f(mm6[0]));
  m.my = floorf(fsg_key2f(mm6[1]));
  m.mz = floorf(fsg_key2f(mm6[2]));
  return m;
}

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
    if (a < 3) atomicMin(&mm6[a], fsg_f2key(v));
    else atomicMax(&mm6[a], fsg_f2key(v));
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

struct EpiK {
  float gamma;
  int b0, b1, b2;
  const float* bias;
  const fsg_tap* bx;
  const fsg_tap* by;
  const fsg_tap* bz;
};

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

int fill_epilogue(const fsg_epilogue* e, EpiK& K) {
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
  hipLaunchKernelGGL(warp_kernel<LT>, fsg_grid3(D.n0, D.n1, D.n2), fsg_block3(), 0, fsg_stream(stream), D, mm6,
                     src_lin, out_lin, src_nn, out_nn, E);
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

int fsg_minmax_init(int32_t* mm, int nmin, int nmax, void* stream) {
  if (!mm || nmin < 0 || nmax < 0 || nmin + nmax <= 0 || nmin + nmax > 64) return FSG_E_BADARG;
  hipLaunchKernelGGL(mm_init_kernel, dim3(1), dim3(64), 0, fsg_stream(stream), mm, nmin, nmax);
  FSG_RETURN_LAUNCH();
}

int fsg_coords_minmax_f32(const fsg_deform* d, int32_t* mm6, void* stream) {
  FsgDeformK D;
  int rc = fsg_fill_deform(d, D);
  if (rc) return rc;
  if (!mm6) return FSG_E_BADARG;
  const int rows = D.n0 * D.n1;
  const int grid = rows < 2048 ? rows : 2048;
  hipLaunchKernelGGL(coords_minmax_kernel, dim3(grid), dim3(256), 0, fsg_stream(stream), D, mm6);
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
  return launch_warp<uint8_t>(d, mm6, src_lin, out_lin, src_nn, out_nn, epi, stream);
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
