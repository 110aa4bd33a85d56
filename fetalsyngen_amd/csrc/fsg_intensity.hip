// fsg_intensity.hip -- K1 GMM intensity draw, device RNG, K5/K8 stand-alone pointwise stages,
// per-label statistics.
//
// Replaces `ImageFromSeeds.sample_intensities` (generator/intensity/rand_gmm.py:146-149: two table
// gathers with int64 indices, a 16.7M-element randn, a multiply-add and a boolean-mask write = 6
// full-volume passes) with one pass: 1 B/voxel label read + 4 B/voxel store (+4 B/voxel when the noise
// is injected from the host RNG tape).  Also RandGamma / RandNoise as stand-alone calls
// (generator/augmentation/synthseg.py:274, :230-233).
#include "fsg_common.h"

namespace {

template <typename LT>
__global__ __launch_bounds__(256) void gmm_kernel(const LT* __restrict__ labels, size_t n,
                                                  const float* __restrict__ mus, const float* __restrict__ sigmas,
                                                  int ntab, const float* __restrict__ noise, uint64_t seed,
                                                  uint64_t stream_id, float* __restrict__ out) {
  __shared__ float s_mu[256], s_sg[256];
  for (int t = threadIdx.x; t < 256; t += blockDim.x) {
    s_mu[t] = t < ntab ? mus[t] : 0.f;
    s_sg[t] = t < ntab ? sigmas[t] : 0.f;
  }
  __syncthreads();
  const size_t nblk = (n + 3) >> 2;  // groups of 4 consecutive voxels = one Philox block
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < nblk; g += (size_t)gridDim.x * blockDim.x) {
    const size_t e = g << 2;
    float z[4];
    if (noise) {
#pragma unroll
      for (int q = 0; q < 4; ++q) z[q] = (e + q < n) ? noise[e + q] : 0.f;
    } else {
      const float4 r = fsg_randn4(seed, stream_id, (uint64_t)g);
      z[0] = r.x; z[1] = r.y; z[2] = r.z; z[3] = r.w;
    }
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int l = (e + q < n) ? (int)labels[e + q] : 0;
      l = min(max(l, 0), 255);
      const float t = s_mu[l] + s_sg[l] * z[q];
      v[q] = t < 0.f ? 0.f : t;
    }
    if (e + 3 < n && ((uintptr_t)(out + e) & 15) == 0) {
      *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e + q < n) out[e + q] = v[q];
    }
  }
}

// K1 with the seed map given as its (up to) four per-meta-label volumes (rand_gmm.py:90-99 sums them):
// the byte-wise sum is formed on the fly from 32-bit words (4 voxels), supports are disjoint so no byte
// overflows into its neighbour.  Saves three full-volume add launches and the summed volume.
__global__ __launch_bounds__(256) void gmm_x4_kernel(const uint8_t* __restrict__ l0, const uint8_t* __restrict__ l1,
                                                     const uint8_t* __restrict__ l2, const uint8_t* __restrict__ l3,
                                                     size_t n, const float* __restrict__ mus,
                                                     const float* __restrict__ sigmas, int ntab,
                                                     const float* __restrict__ noise, uint64_t seed,
                                                     uint64_t stream_id, float* __restrict__ out,
                                                     int32_t* __restrict__ mm, int nmin, int nmax) {
  __shared__ float s_mu[256], s_sg[256];
  for (int t = threadIdx.x; t < 256; t += blockDim.x) {
    s_mu[t] = t < ntab ? mus[t] : 0.f;
    s_sg[t] = t < ntab ? sigmas[t] : 0.f;
  }
  // first kernel of a sample: also resets the sample's min/max keys (saves a launch)
  if (mm && blockIdx.x == 0 && (int)threadIdx.x < nmin + nmax)
    mm[threadIdx.x] = (int)threadIdx.x < nmin ? fsg_f2key(INFINITY) : fsg_f2key(-INFINITY);
  __syncthreads();
  fsg_gmm_x4_loop(l0, l1, l2, l3, n, s_mu, s_sg, noise, seed, stream_id, out, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, size_t n, uint64_t seed,
                                                    uint64_t stream_id) {
  const size_t nblk = (n + 3) >> 2;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < nblk; g += (size_t)gridDim.x * blockDim.x) {
    const float4 r = fsg_randn4(seed, stream_id, (uint64_t)g);
    const size_t e = g << 2;
    const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (e + q < n) out[e + q] = v[q];
  }
}

__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x, size_t n,
                                                        const float* __restrict__ noise, uint64_t seed,
                                                        uint64_t stream_id, float std, float* __restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float z = noise ? noise[e] : fsg_randn1(seed, stream_id, (uint64_t)e);
    const float v = x[e] + std * z;
    out[e] = v < 0.f ? 0.f : v;
  }
}

__global__ __launch_bounds__(256) void gamma_kernel(const float* __restrict__ x, size_t n, float gamma,
                                                    float* __restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    out[e] = 300.0f * powf(x[e] / 300.0f, gamma);
}

__global__ __launch_bounds__(256) void bias_kernel(const float* __restrict__ x, int nx, int ny, int nz,
                                                   const float* __restrict__ bias, int b1, int b2,
                                                   const fsg_tap* __restrict__ bx, const fsg_tap* __restrict__ by,
                                                   const fsg_tap* __restrict__ bz, float* __restrict__ out) {
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (k >= nz || j >= ny) return;
  const size_t o = ((size_t)i * ny + j) * nz + k;
  const float b = fsg_tab_interp<1>(bias, b1, b2, 0, bx[i], by[j], bz[k]);
  out[o] = x[o] * expf(b);
}

// per-label count / sum / sumsq.  Label volumes are piecewise constant, so a wave usually sees 1-3
// distinct labels: peel them off one at a time (readfirstlane + ballot), reduce the matching lanes
// with wave shuffles, and let one lane add into the block's LDS table; one global atomic per
// (block, label present).
__global__ __launch_bounds__(256) void label_stats_kernel(const uint8_t* __restrict__ labels,
                                                          const float* __restrict__ values, size_t n, int nlabels,
                                                          unsigned long long* __restrict__ count,
                                                          double* __restrict__ sum, double* __restrict__ sumsq) {
  __shared__ float s_sum[256], s_sq[256];
  __shared__ unsigned int s_cnt[256];
  for (int t = threadIdx.x; t < 256; t += blockDim.x) { s_sum[t] = 0.f; s_sq[t] = 0.f; s_cnt[t] = 0u; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t per_block = 256 * 16;  // bounded chunk per block keeps the fp32 partial sums small
  const size_t base = (size_t)blockIdx.x * per_block;
  for (int it = 0; it < 16; ++it) {
    const size_t e = base + (size_t)it * 256 + threadIdx.x;
    const bool valid = e < n;
    const int l = valid ? (int)labels[e] : -1;
    const float v = valid ? values[e] : 0.f;
    unsigned long long todo = __ballot(valid && l < nlabels);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int cur = __shfl(l, leader, FSG_WAVE);
      const bool mine = valid && l == cur;
      const unsigned long long mask = __ballot(mine);
      float a = mine ? v : 0.f, b = mine ? v * v : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, FSG_WAVE); b += __shfl_xor(b, o, FSG_WAVE); }
      if (lane == leader) {
        atomicAdd(&s_sum[cur], a);
        atomicAdd(&s_sq[cur], b);
        atomicAdd(&s_cnt[cur], (unsigned int)__popcll(mask));
      }
      todo &= ~mask;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nlabels; t += blockDim.x) {
    if (s_cnt[t]) {
      atomicAdd(&count[t], (unsigned long long)s_cnt[t]);
      atomicAdd(&sum[t], (double)s_sum[t]);
      atomicAdd(&sumsq[t], (double)s_sq[t]);
    }
  }
}

#ifndef FSG_GMM_GRID
#define FSG_GMM_GRID 8192
#endif

inline unsigned grid_for(size_t items, unsigned cap = 4096) {
  size_t b = (items + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

template <typename LT>
int launch_gmm(const LT* labels, size_t n, const float* mus, const float* sigmas, int ntab, const float* noise,
               uint64_t seed, uint64_t stream_id, float* out, void* stream) {
  if (n == 0) return 0;  // empty volume: nothing to do, pointers may be null
  if (!labels || !mus || !sigmas || !out || ntab <= 0 || ntab > 256) return FSG_E_BADARG;
  hipLaunchKernelGGL(gmm_kernel<LT>, dim3(grid_for((n + 3) / 4, 8192)), dim3(256), 0, fsg_stream(stream), labels, n,
                     mus, sigmas, ntab, noise, seed, stream_id, out);
  FSG_RETURN_LAUNCH();
}

}  // namespace

extern "C" {

int fsg_randn_f32(float* out, size_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  if (!out) return FSG_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4, 8192)), dim3(256), 0, fsg_stream(stream), out, n, seed,
                     stream_id);
  FSG_RETURN_LAUNCH();
}

int fsg_gmm_sample_u8(const uint8_t* labels, size_t n, const float* mus, const float* sigmas, int ntab,
                      const float* noise, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
  return launch_gmm<uint8_t>(labels, n, mus, sigmas, ntab, noise, seed, stream_id, out, stream);
}

int fsg_gmm_sample_u8x4(const uint8_t* l0, const uint8_t* l1, const uint8_t* l2, const uint8_t* l3, size_t n,
                        const float* mus, const float* sigmas, int ntab, const float* noise, uint64_t seed,
                        uint64_t stream_id, float* out, void* stream) {
  return fsg_gmm_sample_u8x4_mm(l0, l1, l2, l3, n, mus, sigmas, ntab, noise, seed, stream_id, out, nullptr, 0, 0,
                                stream);
}

int fsg_gmm_sample_u8x4_mm(const uint8_t* l0, const uint8_t* l1, const uint8_t* l2, const uint8_t* l3, size_t n,
                           const float* mus, const float* sigmas, int ntab, const float* noise, uint64_t seed,
                           uint64_t stream_id, float* out, int32_t* mm, int nmin, int nmax, void* stream) {
  if (mm && (nmin < 0 || nmax < 0 || nmin + nmax > 64)) return FSG_E_BADARG;
  if (n == 0) return mm ? fsg_minmax_init(mm, nmin, nmax, stream) : 0;
  if (!l0 || !mus || !sigmas || !out || ntab <= 0 || ntab > 256) return FSG_E_BADARG;
  const uintptr_t al = (uintptr_t)l0 | (uintptr_t)l1 | (uintptr_t)l2 | (uintptr_t)l3;
  if ((al & 3) || ((uintptr_t)out & 15)) return FSG_E_ALIGN;
  hipLaunchKernelGGL(gmm_x4_kernel, dim3(grid_for((n + 3) / 4, FSG_GMM_GRID)), dim3(256), 0, fsg_stream(stream), l0, l1, l2, l3,
                     n, mus, sigmas, ntab, noise, seed, stream_id, out, mm, nmin, nmax);
  FSG_RETURN_LAUNCH();
}

int fsg_gmm_sample_i64(const int64_t* labels, size_t n, const float* mus, const float* sigmas, int ntab,
                       const float* noise, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
  return launch_gmm<int64_t>(labels, n, mus, sigmas, ntab, noise, seed, stream_id, out, stream);
}

int fsg_label_stats_u8(const uint8_t* labels, const float* values, size_t n, int nlabels, unsigned long long* count,
                       double* sum, double* sumsq, void* stream) {
  if (!labels || !values || !count || !sum || !sumsq || nlabels <= 0 || nlabels > 256) return FSG_E_BADARG;
  if (n == 0) return 0;
  const size_t blocks = (n + 4095) / 4096;
  if (blocks > 0x7FFFFFFF) return FSG_E_TOOBIG;
  hipLaunchKernelGGL(label_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), labels, values, n,
                     nlabels, count, sum, sumsq);
  FSG_RETURN_LAUNCH();
}

int fsg_gamma_f32(const float* x, size_t n, float gamma, float* out, void* stream) {
  if (!x || !out) return FSG_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(gamma_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, fsg_stream(stream), x, n, gamma, out);
  FSG_RETURN_LAUNCH();
}

int fsg_bias_mul_f32(const float* x, int nx, int ny, int nz, const float* bias, int b0, int b1, int b2,
                     const fsg_tap* bx, const fsg_tap* by, const fsg_tap* bz, float* out, void* stream) {
  if (!x || !bias || !bx || !by || !bz || !out) return FSG_E_BADARG;
  if (nx <= 0 || ny <= 0 || nz <= 0 || b0 <= 0 || b1 <= 0 || b2 <= 0) return FSG_E_BADARG;
  hipLaunchKernelGGL(bias_kernel, fsg_grid3(nx, ny, nz), fsg_block3(), 0, fsg_stream(stream), x, nx, ny, nz, bias, b1,
                     b2, bx, by, bz, out);
  FSG_RETURN_LAUNCH();
}

int fsg_add_noise_f32(const float* x, size_t n, const float* noise, uint64_t seed, uint64_t stream_id,
                      float noise_std, float* out, void* stream) {
  if (!x || !out) return FSG_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, fsg_stream(stream), x, n, noise, seed,
                     stream_id, noise_std, out);
  FSG_RETURN_LAUNCH();
}

}  // extern "C"
