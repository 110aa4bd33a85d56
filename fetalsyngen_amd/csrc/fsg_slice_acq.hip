// fsg_slice_acq.hip -- slice acquisition (volume -> motion-corrupted slice stacks) and its adjoint
// (slices -> volume), the two operators behind the reference's SR-artifact simulation
// (generator/artifacts/simulate_reco.py: Scanner.scan :386-407, PSFreconstruction :38-54).
//
// Replaces the reference's own CUDA extension (generator/artifacts/svort/slice_acquisition/
// slice_acq_cuda_kernel.cu: forward :17-171, adjoint forward :472-670, equalize :672-693) and, for
// seed-matched comparison with the reference's CPU run, its torch fallback (slice_acq.py:266-546),
// whose arithmetic is different (see include/fsg_hip.h, FSG_SA_*).
//
// MI355X shape of the work: one 256-thread workgroup = a 16x16 pixel tile of ONE slice, so the slice's
// rigid transform is wave-uniform (scalar registers) and the per-tap products r_ij * tap_offset are formed
// once per workgroup into LDS; every lane then walks the PSF raster, and the footprints of neighbouring
// pixels overlap in the volume, so the 2x2x2 gathers of a wave mostly hit lines its neighbours just
// touched (the 216 MiB of a 384^3 volume sit in the 256 MiB Infinity Cache).  The PSF itself lives in LDS
// (uniform reads are broadcasts; the interp_psf branch gathers its 2x2x2 neighbourhood from there).
#include "fsg_common.h"

namespace {

struct SaParams {
  const float* tr;          // (n, 3, 4)
  const float* psf;         // (pd, ph, pw)
  const uint8_t* vmask;     // (D,H,W) or null
  const uint8_t* smask;     // (n,h,w) or null
  const int32_t* sid;       // adjoint only: slice z of the launch reads slices[sid[z]] (null: identity)
  int D, H, W;
  int pd, ph, pw;
  int n, h, w;
  float res;
};

constexpr int SA_TILE = 16;
constexpr int SA_MAX_PSF = 4096;  // elements (16 KB of LDS)
constexpr int SA_MAX_AXIS = 64;

// LDS image of the PSF and of the per-axis products of one slice's rotation with the tap offsets.
struct SaLds {
  float* psf;   // [pd*ph*pw]
  float* tx;    // [3][pw]   r_{a,0} * ix_p
  float* ty;    // [3][ph]   r_{a,1} * iy_p
  float* tz;    // [3][pd]   r_{a,2} * iz_p
};

__device__ __forceinline__ SaLds sa_lds_layout(float* base, const SaParams& P) {
  SaLds L;
  L.psf = base;
  L.tx = L.psf + P.pd * P.ph * P.pw;
  L.ty = L.tx + 3 * P.pw;
  L.tz = L.ty + 3 * P.ph;
  return L;
}

__device__ __forceinline__ void sa_fill_lds(const SaLds& L, const SaParams& P, const float* __restrict__ T) {
  const int tid = threadIdx.y * SA_TILE + threadIdx.x;
  const int np = P.pd * P.ph * P.pw;
  for (int e = tid; e < np; e += SA_TILE * SA_TILE) L.psf[e] = P.psf[e];
  // float * int in the reference == float * (float)int
  for (int e = tid; e < 3 * P.pw; e += SA_TILE * SA_TILE) {
    const int a = e / P.pw, i = e - a * P.pw;
    L.tx[e] = T[a * 4 + 0] * (float)(i - P.pw / 2);
  }
  for (int e = tid; e < 3 * P.ph; e += SA_TILE * SA_TILE) {
    const int a = e / P.ph, i = e - a * P.ph;
    L.ty[e] = T[a * 4 + 1] * (float)(i - P.ph / 2);
  }
  for (int e = tid; e < 3 * P.pd; e += SA_TILE * SA_TILE) {
    const int a = e / P.pd, i = e - a * P.pd;
    L.tz[e] = T[a * 4 + 2] * (float)(i - P.pd / 2);
  }
}

// pixel centre in voxel coordinates (slice_acq_cuda_kernel.cu:46-56): the pixel offset is formed in double
// (the literal `2.` promotes), the rotation in fp32.
__device__ __forceinline__ void sa_centre(const SaParams& P, const float* __restrict__ T, int ix, int iy, float& xc,
                                          float& yc, float& zc) {
  const float _x = (float)(((double)ix - (P.w - 1) / 2.) * (double)P.res + (double)T[3]);
  const float _y = (float)(((double)iy - (P.h - 1) / 2.) * (double)P.res + (double)T[7]);
  const float _z = T[11];
  xc = T[0] * _x + T[1] * _y + T[2] * _z;
  yc = T[4] * _x + T[5] * _y + T[6] * _z;
  zc = T[8] * _x + T[9] * _y + T[10] * _z;
  xc = (float)((double)xc + (P.W - 1) / 2.);
  yc = (float)((double)yc + (P.H - 1) / 2.);
  zc = (float)((double)zc + (P.D - 1) / 2.);
}

// interp_psf branch (:79-101): PSF value at the snapped voxel, trilinear in the PSF grid.  false = outside.
__device__ __forceinline__ bool sa_psf_at(const SaLds& L, const SaParams& P, const float* __restrict__ T, float dx,
                                          float dy, float dz, float& val) {
  const float xp = (float)((double)(T[0] * dx + T[4] * dy + T[8] * dz) + (P.pw - 1) / 2.);
  const float yp = (float)((double)(T[1] * dx + T[5] * dy + T[9] * dz) + (P.ph - 1) / 2.);
  const float zp = (float)((double)(T[2] * dx + T[6] * dy + T[10] * dz) + (P.pd - 1) / 2.);
  if (xp < 0 || yp < 0 || zp < 0 || xp >= (float)(P.pw - 1) || yp >= (float)(P.ph - 1) || zp >= (float)(P.pd - 1))
    return false;
  const float xf = floorf(xp), yf = floorf(yp), zf = floorf(zp);
  const float wx = xp - xf, wy = yp - yf, wz = zp - zf;
  const int sy = P.pw, sz = P.pw * P.ph;
  const float* q = L.psf + (int)zf * sz + (int)yf * sy + (int)xf;
  float v = 0.f;
  v += (1 - wx) * (1 - wy) * (1 - wz) * q[0];
  v += wx * (1 - wy) * (1 - wz) * q[1];
  v += (1 - wx) * wy * (1 - wz) * q[sy];
  v += (1 - wx) * (1 - wy) * wz * q[sz];
  v += wx * wy * (1 - wz) * q[1 + sy];
  v += wx * (1 - wy) * wz * q[1 + sz];
  v += (1 - wx) * wy * wz * q[sy + sz];
  v += wx * wy * wz * q[sy + sz + 1];
  val = v;
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// forward, CUDA-kernel semantics.  NN = interp_psf.
// ---------------------------------------------------------------------------------------------------------
template <bool NN, bool VM>
__global__ __launch_bounds__(SA_TILE* SA_TILE) void sa_forward_kernel(SaParams P, const float* __restrict__ vol,
                                                                      float* __restrict__ slices,
                                                                      float* __restrict__ weights) {
  extern __shared__ float smem[];
  const int in = blockIdx.z;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  const SaLds L = sa_lds_layout(smem, P);
  sa_fill_lds(L, P, T);
  __syncthreads();
  const int ix = blockIdx.x * SA_TILE + threadIdx.x, iy = blockIdx.y * SA_TILE + threadIdx.y;
  if (ix >= P.w || iy >= P.h) return;
  const size_t idx = ((size_t)in * P.h + iy) * P.w + ix;
  if (P.smask && !P.smask[idx]) {
    slices[idx] = 0.f;
    if (weights) weights[idx] = 0.f;
    return;
  }
  float xc, yc, zc;
  sa_centre(P, T, ix, iy, xc, yc, zc);
  const int Sy = P.W, Sz = P.H * P.W;
  const float hx = (float)(P.W - 1), hy = (float)(P.H - 1), hz = (float)(P.D - 1);
  float val = 0.f, weight = 0.f;
  int ip = 0;
  for (int kz = 0; kz < P.pd; ++kz) {
    const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
    for (int ky = 0; ky < P.ph; ++ky) {
      const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
      for (int kx = 0; kx < P.pw; ++kx, ++ip) {
        float pv = L.psf[ip];
        if (pv == 0.f) continue;  // wave-uniform
        const float x = xc + L.tx[kx] + yx + zx;
        const float y = yc + L.tx[P.pw + kx] + yy + zy;
        const float z = zc + L.tx[2 * P.pw + kx] + yz + zz;
        if (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz) continue;
        if (NN) {
          const float xr = roundf(x), yr = roundf(y), zr = roundf(z);
          const int iv = (int)zr * Sz + (int)yr * Sy + (int)xr;
          if (VM && !P.vmask[iv]) continue;
          if (!sa_psf_at(L, P, T, xr - xc, yr - yc, zr - zc, pv)) continue;
          val += pv * vol[iv];
          weight += pv;
        } else {
          const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
          const float wx = x - xf, wy = y - yf, wz = z - zf;
          const int iv = (int)zf * Sz + (int)yf * Sy + (int)xf;
          const float* __restrict__ q = vol + iv;
          float pw_;
#define SA_CORNER(OFF, WEXPR)                     \
  if (!VM || P.vmask[iv + (OFF)]) {               \
    pw_ = (WEXPR) * pv;                           \
    val += pw_ * q[(OFF)];                        \
    weight += pw_;                                \
  }
          SA_CORNER(0, (1 - wx) * (1 - wy) * (1 - wz))
          SA_CORNER(1, wx * (1 - wy) * (1 - wz))
          SA_CORNER(Sy, (1 - wx) * wy * (1 - wz))
          SA_CORNER(Sz, (1 - wx) * (1 - wy) * wz)
          SA_CORNER(1 + Sy, wx * wy * (1 - wz))
          SA_CORNER(1 + Sz, wx * (1 - wy) * wz)
          SA_CORNER(Sy + Sz, (1 - wx) * wy * wz)
          SA_CORNER(Sy + Sz + 1, wx * wy * wz)
#undef SA_CORNER
        }
      }
    }
  }
  const bool good = weight > 0.f;
  slices[idx] = good ? val / weight : 0.f;
  if (weights) weights[idx] = good ? weight : 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// adjoint, CUDA-kernel semantics (:472-670): pass 1 = pixel weight, pass 2 = scatter with fp32 atomics.
// ---------------------------------------------------------------------------------------------------------
template <bool NN, bool VM>
__global__ __launch_bounds__(SA_TILE* SA_TILE) void sa_adjoint_kernel(SaParams P, const float* __restrict__ slices,
                                                                      float* __restrict__ vol,
                                                                      float* __restrict__ vol_weight) {
  extern __shared__ float smem[];
  const int in = blockIdx.z;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  const SaLds L = sa_lds_layout(smem, P);
  sa_fill_lds(L, P, T);
  __syncthreads();
  const int ix = blockIdx.x * SA_TILE + threadIdx.x, iy = blockIdx.y * SA_TILE + threadIdx.y;
  if (ix >= P.w || iy >= P.h) return;
  const size_t idx = ((size_t)(P.sid ? P.sid[in] : in) * P.h + iy) * P.w + ix;
  if (P.smask && !P.smask[idx]) return;
  const float s = slices[idx];
  float xc, yc, zc;
  sa_centre(P, T, ix, iy, xc, yc, zc);
  const int Sy = P.W, Sz = P.H * P.W;
  const float hx = (float)(P.W - 1), hy = (float)(P.H - 1), hz = (float)(P.D - 1);

  float weight = 0.f;
  int ip = 0;
  for (int kz = 0; kz < P.pd; ++kz) {
    const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
    for (int ky = 0; ky < P.ph; ++ky) {
      const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
      for (int kx = 0; kx < P.pw; ++kx, ++ip) {
        float pv = L.psf[ip];
        if (pv == 0.f) continue;
        const float x = xc + L.tx[kx] + yx + zx;
        const float y = yc + L.tx[P.pw + kx] + yy + zy;
        const float z = zc + L.tx[2 * P.pw + kx] + yz + zz;
        if (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz) continue;
        if (NN && !sa_psf_at(L, P, T, roundf(x) - xc, roundf(y) - yc, roundf(z) - zc, pv)) continue;
        weight += pv;
      }
    }
  }
  if (weight < 0.5f) return;  // border

  ip = 0;
  for (int kz = 0; kz < P.pd; ++kz) {
    const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
    for (int ky = 0; ky < P.ph; ++ky) {
      const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
      for (int kx = 0; kx < P.pw; ++kx, ++ip) {
        float pv = L.psf[ip];
        if (pv == 0.f) continue;
        const float x = xc + L.tx[kx] + yx + zx;
        const float y = yc + L.tx[P.pw + kx] + yy + zy;
        const float z = zc + L.tx[2 * P.pw + kx] + yz + zz;
        if (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz) continue;
        if (NN) {
          const float xr = roundf(x), yr = roundf(y), zr = roundf(z);
          if (!sa_psf_at(L, P, T, xr - xc, yr - yc, zr - zc, pv)) continue;
          pv /= weight;
          const int iv = (int)zr * Sz + (int)yr * Sy + (int)xr;
          if (VM && !P.vmask[iv]) continue;
          unsafeAtomicAdd(vol + iv, pv * s);
          if (vol_weight) unsafeAtomicAdd(vol_weight + iv, pv);
        } else {
          const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
          const float wx = x - xf, wy = y - yf, wz = z - zf;
          const int iv = (int)zf * Sz + (int)yf * Sy + (int)xf;
          pv /= weight;
          float pw_;
#define SA_CORNER(OFF, WEXPR)                                            \
  if (!VM || P.vmask[iv + (OFF)]) {                                      \
    pw_ = (WEXPR) * pv;                                                  \
    unsafeAtomicAdd(vol + iv + (OFF), pw_ * s);                          \
    if (vol_weight) unsafeAtomicAdd(vol_weight + iv + (OFF), pw_);       \
  }
          SA_CORNER(0, (1 - wx) * (1 - wy) * (1 - wz))
          SA_CORNER(1, wx * (1 - wy) * (1 - wz))
          SA_CORNER(Sy, (1 - wx) * wy * (1 - wz))
          SA_CORNER(Sz, (1 - wx) * (1 - wy) * wz)
          SA_CORNER(1 + Sy, wx * wy * (1 - wz))
          SA_CORNER(1 + Sz, wx * (1 - wy) * wz)
          SA_CORNER(Sy + Sz, (1 - wx) * wy * wz)
          SA_CORNER(Sy + Sz + 1, wx * wy * wz)
#undef SA_CORNER
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// torch-fallback semantics (slice_acq.py:272-310): position = (shift + R((off - T) + T)) + R(pixel + T),
// strict inside test, round half to even, raw PSF value; one kernel serves forward and adjoint.
// LDS: per-tap rotated offsets (3 floats) next to the PSF.
// ---------------------------------------------------------------------------------------------------------
template <bool ADJ, bool GS = false>
__global__ __launch_bounds__(SA_TILE* SA_TILE) void sa_torch_kernel(SaParams P, const float* __restrict__ vol_in,
                                                                    float* __restrict__ slices_io,
                                                                    float* __restrict__ weights_out,
                                                                    float* __restrict__ vol_out,
                                                                    float* __restrict__ vol_weight) {
  extern __shared__ float smem[];
  const int in = blockIdx.z;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  const int np = P.pd * P.ph * P.pw;
  float* lp = smem;           // psf
  float* lo = smem + np;      // [np][3] shift + rotated tap offset
  const int tid = threadIdx.y * SA_TILE + threadIdx.x;
  const float shx = ((float)P.W - 1.f) / 2.0f, shy = ((float)P.H - 1.f) / 2.0f, shz = ((float)P.D - 1.f) / 2.0f;
  for (int e = tid; e < np; e += SA_TILE * SA_TILE) {
    lp[e] = P.psf[e];
    const int kx = e % P.pw, ky = (e / P.pw) % P.ph, kz = e / (P.pw * P.ph);
    // xyz_masked_untransformed(psf > 0, shape, 1.0) (:266-269): (index - (n-1)/2) * 1.0 in fp32
    const float ox = ((float)kx - ((float)P.pw - 1.f) / 2.f) * 1.0f;
    const float oy = ((float)ky - ((float)P.ph - 1.f) / 2.f) * 1.0f;
    const float oz = ((float)kz - ((float)P.pd - 1.f) / 2.f) * 1.0f;
    const float ax = (ox - T[3]) + T[3], ay = (oy - T[7]) + T[7], az = (oz - T[11]) + T[11];
    lo[3 * e + 0] = shx + (T[0] * ax + T[1] * ay + T[2] * az);
    lo[3 * e + 1] = shy + (T[4] * ax + T[5] * ay + T[6] * az);
    lo[3 * e + 2] = shz + (T[8] * ax + T[9] * ay + T[10] * az);
  }
  __syncthreads();
  const int ix = blockIdx.x * SA_TILE + threadIdx.x, iy = blockIdx.y * SA_TILE + threadIdx.y;
  if (ix >= P.w || iy >= P.h) return;
  const size_t idx = ((size_t)((ADJ && P.sid) ? P.sid[in] : in) * P.h + iy) * P.w + ix;
  const bool live = !P.smask || P.smask[idx];
  if (!live) {
    if (!ADJ) {
      slices_io[idx] = 0.f;
      if (weights_out) weights_out[idx] = 0.f;
    }
    return;
  }
  const float px = (((float)ix - ((float)P.w - 1.f) / 2.f) * P.res) + T[3];
  const float py = (((float)iy - ((float)P.h - 1.f) / 2.f) * P.res) + T[7];
  const float pz = (0.f * P.res) + T[11];
  const float sx = T[0] * px + T[1] * py + T[2] * pz;
  const float sy = T[4] * px + T[5] * py + T[6] * pz;
  const float sz = T[8] * px + T[9] * py + T[10] * pz;
  const float lx = shx * 2, ly = shy * 2, lz = shz * 2;
  const int Sy = P.W, Sz = P.H * P.W;
  if (GS) {
    // 1x1x1 PSF without weights: slice_acquisition_no_psf_torch (slice_acq.py:445-480) = F.grid_sample(trilinear,
    // zeros padding, align_corners=True) at R(pixel + T) / ((n-1)/2)
    const float x = ((sx / shx + 1.f) / 2.f) * ((float)P.W - 1.f);
    const float y = ((sy / shy + 1.f) / 2.f) * ((float)P.H - 1.f);
    const float z = ((sz / shz + 1.f) / 2.f) * ((float)P.D - 1.f);
    const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
    const float wx = x - xf, wy = y - yf, wz = z - zf;
    const int x0 = (int)xf, y0 = (int)yf, z0 = (int)zf;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int cx = x0 + (c & 1), cy = y0 + ((c >> 1) & 1), cz = z0 + (c >> 2);
      if (cx < 0 || cy < 0 || cz < 0 || cx >= P.W || cy >= P.H || cz >= P.D) continue;
      const float wgt = ((c & 1) ? wx : 1.f - wx) * (((c >> 1) & 1) ? wy : 1.f - wy) * ((c >> 2) ? wz : 1.f - wz);
      const int iv = cz * Sz + cy * Sy + cx;
      float v = vol_in[iv];
      if (P.vmask) v = v * (float)P.vmask[iv];
      acc += v * wgt;
    }
    slices_io[idx] = acc;
    return;
  }
  const float s = ADJ ? slices_io[idx] : 0.f;
  float val = 0.f, weight = 0.f;
  for (int e = 0; e < np; ++e) {
    const float pv = lp[e];
    if (!(pv > 0.f)) continue;
    const float x = lo[3 * e] + sx, y = lo[3 * e + 1] + sy, z = lo[3 * e + 2] + sz;
    if (!(x > 0 && y > 0 && z > 0 && x < lx && y < ly && z < lz)) continue;
    const int iv = (int)rintf(z) * Sz + (int)rintf(y) * Sy + (int)rintf(x);
    if (ADJ) {
      unsafeAtomicAdd(vol_out + iv, pv * s);
      if (vol_weight) unsafeAtomicAdd(vol_weight + iv, pv);
    } else {
      float v = vol_in[iv];
      if (P.vmask) v = v * (float)P.vmask[iv];
      val += pv * v;
      weight += pv;
    }
  }
  if (!ADJ) {
    slices_io[idx] = weight > 1e-2f ? val / weight : val;
    if (weights_out) weights_out[idx] = weight;
  }
}

// equalize (:672-693, is_grad = false) / the fallback's `weight > 1e-2` rule (slice_acq.py:539-545)
__global__ __launch_bounds__(256) void sa_equalize_kernel(float* __restrict__ vol, const float* __restrict__ w,
                                                          const uint8_t* __restrict__ vmask, float thr, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    float v = vol[e];
    if (w) {
      const float ww = w[e];
      if (ww > thr) v = v / ww;
    }
    if (vmask) v = v * (float)vmask[e];
    vol[e] = v;
  }
}

int sa_check(const SaParams& P, size_t& lds, bool torch_mode) {
  if (!P.tr || !P.psf) return FSG_E_BADARG;
  if (P.D < 2 || P.H < 2 || P.W < 2 || P.n <= 0 || P.h <= 0 || P.w <= 0 || P.pd <= 0 || P.ph <= 0 || P.pw <= 0)
    return FSG_E_BADARG;
  if ((size_t)P.D * P.H * P.W > (size_t)0x7FFFFFFF || (size_t)P.n * P.h * P.w > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  if (P.pd > SA_MAX_AXIS || P.ph > SA_MAX_AXIS || P.pw > SA_MAX_AXIS || P.pd * P.ph * P.pw > SA_MAX_PSF) return FSG_E_TOOBIG;
  if (P.n > 65535) return FSG_E_TOOBIG;
  const size_t np = (size_t)P.pd * P.ph * P.pw;
  lds = torch_mode ? np * 4 * sizeof(float) : (np + 3 * (size_t)(P.pd + P.ph + P.pw)) * sizeof(float);
  return 0;
}

dim3 sa_grid(const SaParams& P) {
  return dim3((unsigned)((P.w + SA_TILE - 1) / SA_TILE), (unsigned)((P.h + SA_TILE - 1) / SA_TILE), (unsigned)P.n);
}

}  // namespace

extern "C" {

int fsg_slice_acq_forward_f32(const float* transforms, const float* vol, const uint8_t* vol_mask, const float* psf, int pd,
                              int ph, int pw, const uint8_t* slices_mask, float* slices, float* slices_weight, int D, int H,
                              int W, int n, int h, int w, float res_slice, int mode, void* stream) {
  if (!vol || !slices || mode < 0 || mode > 2) return FSG_E_BADARG;
  SaParams P{transforms, psf, vol_mask, slices_mask, nullptr, D, H, W, pd, ph, pw, n, h, w, res_slice};
  size_t lds = 0;
  const int rc = sa_check(P, lds, mode == FSG_SA_TORCH);
  if (rc) return rc;
  const dim3 grid = sa_grid(P), block(SA_TILE, SA_TILE);
  hipStream_t st = fsg_stream(stream);
  if (mode == FSG_SA_TORCH) {
    if (pd * ph * pw == 1 && !slices_weight)
      hipLaunchKernelGGL((sa_torch_kernel<false, true>), grid, block, lds, st, P, vol, slices, slices_weight, (float*)nullptr,
                         (float*)nullptr);
    else
      hipLaunchKernelGGL((sa_torch_kernel<false, false>), grid, block, lds, st, P, vol, slices, slices_weight, (float*)nullptr,
                         (float*)nullptr);
  } else if (mode == FSG_SA_LINEAR) {
    if (vol_mask) hipLaunchKernelGGL((sa_forward_kernel<false, true>), grid, block, lds, st, P, vol, slices, slices_weight);
    else hipLaunchKernelGGL((sa_forward_kernel<false, false>), grid, block, lds, st, P, vol, slices, slices_weight);
  } else {
    if (vol_mask) hipLaunchKernelGGL((sa_forward_kernel<true, true>), grid, block, lds, st, P, vol, slices, slices_weight);
    else hipLaunchKernelGGL((sa_forward_kernel<true, false>), grid, block, lds, st, P, vol, slices, slices_weight);
  }
  FSG_RETURN_LAUNCH();
}

int fsg_slice_acq_adjoint_f32(const float* transforms, const float* psf, int pd, int ph, int pw, const float* slices,
                              const uint8_t* slices_mask, const int32_t* slice_ids, const uint8_t* vol_mask, float* vol,
                              float* vol_weight, int D, int H, int W, int n, int h, int w, float res_slice, int mode,
                              void* stream) {
  if (!slices || !vol || mode < 0 || mode > 2) return FSG_E_BADARG;
  SaParams P{transforms, psf, vol_mask, slices_mask, slice_ids, D, H, W, pd, ph, pw, n, h, w, res_slice};
  size_t lds = 0;
  const int rc = sa_check(P, lds, mode == FSG_SA_TORCH);
  if (rc) return rc;
  hipStream_t st = fsg_stream(stream);
  const size_t bytes = (size_t)D * H * W * sizeof(float);
  hipError_t e = hipMemsetAsync(vol, 0, bytes, st);
  if (e != hipSuccess) return (int)e;
  if (vol_weight) {
    e = hipMemsetAsync(vol_weight, 0, bytes, st);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid = sa_grid(P), block(SA_TILE, SA_TILE);
  if (mode == FSG_SA_TORCH) {
    // the fallback multiplies by vol_mask at the very end (fsg_equalize_f32), not per contribution
    hipLaunchKernelGGL((sa_torch_kernel<true, false>), grid, block, lds, st, P, (const float*)nullptr, const_cast<float*>(slices),
                       (float*)nullptr, vol, vol_weight);
  } else if (mode == FSG_SA_LINEAR) {
    if (vol_mask) hipLaunchKernelGGL((sa_adjoint_kernel<false, true>), grid, block, lds, st, P, slices, vol, vol_weight);
    else hipLaunchKernelGGL((sa_adjoint_kernel<false, false>), grid, block, lds, st, P, slices, vol, vol_weight);
  } else {
    if (vol_mask) hipLaunchKernelGGL((sa_adjoint_kernel<true, true>), grid, block, lds, st, P, slices, vol, vol_weight);
    else hipLaunchKernelGGL((sa_adjoint_kernel<true, false>), grid, block, lds, st, P, slices, vol, vol_weight);
  }
  FSG_RETURN_LAUNCH();
}

int fsg_equalize_f32(float* vol, const float* vol_weight, const uint8_t* vol_mask, float threshold, size_t n, void* stream) {
  if (!vol || n == 0 || (!vol_weight && !vol_mask)) return FSG_E_BADARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sa_equalize_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), vol, vol_weight, vol_mask,
                     threshold, n);
  FSG_RETURN_LAUNCH();
}

}  // extern "C"
