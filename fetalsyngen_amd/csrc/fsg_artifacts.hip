// fsg_artifacts.hip -- the volumetric pieces of the SR-artifact stages:
//   * mixture-of-Gaussians weight volume        generator/artifacts/utils.py:125-160 `mog_3d_tensor`
//   * fractal Perlin noise weight volume        generator/artifacts/utils.py:224-388
//   * spatially weighted blends                 simulate_reco.py:704, augmentation/artifacts.py:125, :322-337
//   * slice-stack corruptions of the scanner    simulate_reco.py:236-298 (Rician noise, signal voids), :409 (sums)
//   * binary morphology for the boundary stage  generator/artifacts/utils.py:163-210, augmentation/artifacts.py:484-499
//
// All of it is streaming work over a (D,H,W) volume: one pass, coalesced along the fastest axis, everything that
// depends on one index only (per-axis Gaussian terms, Perlin lattice coordinates) hoisted into small per-axis
// tables that stay in L1/L2.  The reference builds full (D,H,W) coordinate grids and one full-volume temporary
// per Gaussian / per lattice corner.
#include "fsg_common.h"

namespace {

// ---- mixture of Gaussians ----------------------------------------------------------------------------------
// tab[(g*3 + a) * L + i] = ((i - c[g][a]) / s[g][a])^2, a = 0: last axis (x), 1: middle (y), 2: first (z);
// L = max(D,H,W).  c/s are given in (x0,y0,z0) order, exactly as mog_3d_tensor unpacks them (:151-156).
__global__ __launch_bounds__(256) void mog_tables_kernel(const float* __restrict__ centers, const float* __restrict__ sigmas,
                                                         int k, int D, int H, int W, int L, float* __restrict__ tab) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= k * 3 * L) return;
  const int i = e % L, ga = e / L, a = ga % 3;
  const int n = a == 0 ? W : (a == 1 ? H : D);
  if (i >= n) return;
  const float t = ((float)i - centers[ga]) / sigmas[ga];
  tab[e] = t * t;
}

// out = clamp(sum_g exp(-(tx + ty + tz) / 2), 0, 1); one thread = 4 consecutive x
template <bool FAST>
__global__ __launch_bounds__(256) void mog_sum_kernel(const float* __restrict__ tab, int k, int D, int H, int W, int L,
                                                      float* __restrict__ out) {
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y, z = blockIdx.z;
  if (x4 >= W) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < k; ++g) {
    const float* tx = tab + (size_t)(g * 3 + 0) * L;
    const float ty = tab[(size_t)(g * 3 + 1) * L + y];
    const float tz = tab[(size_t)(g * 3 + 2) * L + z];
    // a blob whose (y,z) distance alone puts exp(-d/2) below the smallest normal fp32 (e^-88) adds nothing to this row:
    // uniform skip (y, z are per-workgroup), which is what makes 200 narrow blobs cost as much as the few that reach a row
    if (ty + tz > 176.f) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (x4 + q < W) {
        const float d = tx[x4 + q] + ty + tz;
        acc[q] += FAST ? __builtin_amdgcn_exp2f(d * -0.72134752044448170368f) : expf(-d / 2);  // v_exp_f32
      }
    }
  }
  float* o = out + ((size_t)z * H + y) * W + x4;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (x4 + q < W) o[q] = fminf(fmaxf(acc[q], 0.f), 1.f);
}

// ---- fractal Perlin noise ----------------------------------------------------------------------------------
struct PerlinOct {
  const float* grad;  // (r0+1, r1+1, r2+1, 3)
  const float* lin;   // linspace(0, r_a, n_a) for a = 0,1,2 back to back (n0 + n1 + n2 floats)
  int r[3];
  float amp;
};
struct PerlinArgs {
  PerlinOct o[8];
  int noct;
  int n0, n1, n2;
};

__device__ __forceinline__ float perlin_fade(float t) { return t * t * t * (t * (t * 6 - 15) + 10); }

// one octave at (i,j,k) (utils.py:255-327): lattice cell = floor(grid), corner index clamped to r, dot of the
// corner gradient with the offset to the corner, quintic fade, lerp along axis 0, then 1, then 2.
__device__ __forceinline__ float perlin_octave(const PerlinOct& o, int n0, int n1, int i, int j, int k) {
  const float g0 = o.lin[i], g1 = o.lin[n0 + j], g2 = o.lin[n0 + n1 + k];
  const float f0 = floorf(g0), f1 = floorf(g1), f2 = floorf(g2);
  const float l0 = g0 - f0, l1 = g1 - f1, l2 = g2 - f2;
  const int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
  const int a0 = min(c0, o.r[0]), a1 = min(c0 + 1, o.r[0]);
  const int b0 = min(c1, o.r[1]), b1 = min(c1 + 1, o.r[1]);
  const int d0 = min(c2, o.r[2]), d1 = min(c2 + 1, o.r[2]);
  const int s1 = (o.r[2] + 1) * 3, s0 = (o.r[1] + 1) * s1;
  auto dot = [&](int ia, int ib, int id, float ox, float oy, float oz) {
    const float* g = o.grad + ia * s0 + ib * s1 + id * 3;
    return g[0] * (l0 - ox) + g[1] * (l1 - oy) + g[2] * (l2 - oz);
  };
  const float n000 = dot(a0, b0, d0, 0.f, 0.f, 0.f), n100 = dot(a1, b0, d0, 1.f, 0.f, 0.f);
  const float n010 = dot(a0, b1, d0, 0.f, 1.f, 0.f), n110 = dot(a1, b1, d0, 1.f, 1.f, 0.f);
  const float n001 = dot(a0, b0, d1, 0.f, 0.f, 1.f), n101 = dot(a1, b0, d1, 1.f, 0.f, 1.f);
  const float n011 = dot(a0, b1, d1, 0.f, 1.f, 1.f), n111 = dot(a1, b1, d1, 1.f, 1.f, 1.f);
  const float t0 = perlin_fade(l0), t1 = perlin_fade(l1), t2 = perlin_fade(l2);
  const float n00 = n000 * (1 - t0) + t0 * n100;
  const float n10 = n010 * (1 - t0) + t0 * n110;
  const float n01 = n001 * (1 - t0) + t0 * n101;
  const float n11 = n011 * (1 - t0) + t0 * n111;
  const float m0 = n00 * (1 - t1) + t1 * n10;
  const float m1 = n01 * (1 - t1) + t1 * n11;
  return m0 * (1 - t2) + t2 * m1;
}

// raw fractal sum (utils.py:375-384) + its global min / max as ordered keys
__global__ __launch_bounds__(256) void perlin_kernel(PerlinArgs A, float* __restrict__ out, int32_t* __restrict__ mm) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y, i = blockIdx.z;
  float v = 0.f;
  const bool live = k < A.n2;
  if (live) {
    for (int q = 0; q < A.noct; ++q) v += A.o[q].amp * perlin_octave(A.o[q], A.n0, A.n1, i, j, k);
    out[((size_t)i * A.n1 + j) * A.n2 + k] = v;
  }
  float lo = fsg_wave_min(live ? v : INFINITY), hi = fsg_wave_max(live ? v : -INFINITY);
  __shared__ float red[2][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
    fsg_atomic_min_key(&mm[0], lo);
    fsg_atomic_max_key(&mm[1], hi);
  }
}

// ---- weighted blend ----------------------------------------------------------------------------------------
struct BlendArgs {
  const float* a;
  const float* b;
  const float* w;
  const int32_t* w_mm;   // W_PERLIN: min/max keys of w
  float increase;
  const float* seg;      // optional: w *= (seg > 0)
  int w_mode;            // 0 plain weight volume, 1 raw Perlin noise normalised on the fly (utils.py:386-387)
  int b_mode;            // 0 plain, 1 structured noise: b' = clamp(a + std * b / max|b|, 0, 2 max a) (artifacts.py:322-327)
  const int32_t* b_mm;   // b_mode 1: min/max keys of b
  const int32_t* a_mm;   // b_mode 1: min/max keys of a
  float std;
  float* out;            // (1 - w) a + w b
  float* w_out;          // optional: the weight actually used
  size_t n;
};

__global__ __launch_bounds__(256) void blend_kernel(BlendArgs B) {
  float mn = 0.f, den = 1.f, bscale = 1.f, amax2 = 0.f;
  if (B.w_mode == 1) {
    mn = fsg_key2f(B.w_mm[0]);
    den = fsg_key2f(B.w_mm[1]) - mn;
  }
  if (B.b_mode == 1) {
    bscale = fmaxf(fabsf(fsg_key2f(B.b_mm[0])), fabsf(fsg_key2f(B.b_mm[1])));
    amax2 = fsg_key2f(B.a_mm[1]) * 2;
  }
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < B.n; e += (size_t)gridDim.x * blockDim.x) {
    float w = B.w[e];
    if (B.w_mode == 1) w = fminf(fmaxf((w + B.increase - mn) / den, 0.f), 1.f);
    if (B.seg) w = (B.seg[e] > 0.f ? 1.f : 0.f) * w;
    if (B.w_out) B.w_out[e] = w;
    if (B.out) {
      const float a = B.a[e];
      float b = B.b[e];
      if (B.b_mode == 1) b = fminf(fmaxf(a + B.std * (b / bscale), 0.f), amax2);
      B.out[e] = (1 - w) * a + w * b;
    }
  }
}

// ---- scanner corruptions of a slice stack ------------------------------------------------------------------
// Rician noise on the pixels above the threshold (simulate_reco.py:247-255); z from Philox (2 per pixel) or from
// dense fields n1, n2 (host-tape mode: the reference's compacted draws scattered back by the caller).
__global__ __launch_bounds__(256) void slice_noise_kernel(float* __restrict__ s, size_t n, float thr, float sigma,
                                                          const float* __restrict__ n1, const float* __restrict__ n2,
                                                          uint64_t seed, uint64_t stream_id) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float v = s[e];
    if (!(v > thr)) continue;
    float z1, z2;
    if (n1) {
      z1 = n1[e];
      z2 = n2[e];
    } else {
      const float4 z = fsg_randn4(seed, stream_id, e >> 1);
      z1 = (e & 1) ? z.z : z.x;
      z2 = (e & 1) ? z.w : z.y;
    }
    const float p = v + z1 * sigma, q = z2 * sigma;
    s[e] = sqrtf(p * p + q * q);
  }
}

// signal voids (simulate_reco.py:258-298): slice sid[t] *= 1 - A exp(sx x'^2 + sy y'^2), (x', y') the pixel
// position rotated about a random centre.  par[t] = {yc, xc, cos, sin, A, sx, sy}; ylin/xlin = the linspace axes.
__global__ __launch_bounds__(256) void slice_void_kernel(float* __restrict__ s, int h, int w, const int32_t* __restrict__ sid,
                                                         const float* __restrict__ par, const float* __restrict__ ylin,
                                                         const float* __restrict__ xlin) {
  const int t = blockIdx.y;
  const float* p = par + t * 7;
  float* sl = s + (size_t)sid[t] * h * w;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < h * w; e += gridDim.x * blockDim.x) {
    const int iy = e / w, ix = e - iy * w;
    const float y = ylin[iy] - p[0], x = xlin[ix] - p[1];
    const float xr = p[2] * x - p[3] * y, yr = p[3] * x + p[2] * y;
    const float m = 1 - p[4] * expf(p[5] * (xr * xr) + p[6] * (yr * yr));
    sl[e] *= m;
  }
}

// per-slice sums (simulate_reco.py:409): one workgroup per slice, fixed reduction tree (deterministic)
__global__ __launch_bounds__(1024) void slice_sums_kernel(const float* __restrict__ s, size_t hw, float* __restrict__ out) {
  const float* sl = s + (size_t)blockIdx.x * hw;
  double acc = 0.0;
  for (size_t e = threadIdx.x; e < hw; e += blockDim.x) acc += (double)sl[e];
  __shared__ double red[1024];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)red[0];
}


// ---- k-th set voxel (torch.where(mask)[...][randperm] of the reference, without materialising the index lists) --
constexpr int NZ_BLOCK = 4096;  // voxels per count bucket

template <typename T>
__device__ __forceinline__ bool nz_pred(T v, int mode, float value) {
  const float f = (float)v;
  return mode == 0 ? f > value : (mode == 1 ? f == value : f != value);
}

template <typename T>
__global__ __launch_bounds__(256) void nonzero_count_kernel(const T* __restrict__ v, size_t n, int mode, float value,
                                                            int32_t* __restrict__ counts) {
  const size_t base = (size_t)blockIdx.x * NZ_BLOCK;
  int c = 0;
  for (int e = threadIdx.x; e < NZ_BLOCK; e += 256)
    if (base + e < n && nz_pred(v[base + e], mode, value)) ++c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, FSG_WAVE);
  __shared__ int red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one wave per request: flat index of the rank[q]-th set voxel (raster order) of bucket blk[q]
template <typename T>
__global__ __launch_bounds__(64) void nonzero_select_kernel(const T* __restrict__ v, size_t n, int mode, float value,
                                                            const int32_t* __restrict__ blk, const int32_t* __restrict__ rank,
                                                            long long* __restrict__ out) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const size_t base = (size_t)blk[q] * NZ_BLOCK;
  int r = rank[q];
  for (int it = 0; it < NZ_BLOCK / 64; ++it) {
    const size_t e = base + (size_t)it * 64 + lane;
    const bool p = e < n && nz_pred(v[e], mode, value);
    const unsigned long long b = __ballot(p);
    const int cnt = __popcll(b);
    if (r < cnt) {
      const int before = __popcll(b & ((1ull << lane) - 1ull));
      if (p && before == r) out[q] = (long long)e;
      return;
    }
    r -= cnt;
  }
  if (lane == 0) out[q] = -1;
}


// values[e] for the voxels of bucket b satisfying the predicate, written in raster order from offsets[b]
__global__ __launch_bounds__(64) void compact_kernel(const float* __restrict__ values, const float* __restrict__ pv, size_t n,
                                                     int mode, float value, const long long* __restrict__ offsets,
                                                     float* __restrict__ out) {
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * NZ_BLOCK;
  long long o = offsets[blockIdx.x];
  for (int it = 0; it < NZ_BLOCK / 64; ++it) {
    const size_t e = base + (size_t)it * 64 + lane;
    const bool p = e < n && nz_pred(pv[e], mode, value);
    const unsigned long long b = __ballot(p);
    if (p) out[o + __popcll(b & ((1ull << lane) - 1ull))] = values[e];
    o += __popcll(b);
  }
}

// small element-wise helpers: 0: a + b, 1: a > value, 2: a == value, 3: a * b, 4: a * (b > value), 5: max(a, b),
// 6: (a - b) > value, 7: a <= value
__global__ __launch_bounds__(256) void ewise_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, int op,
                                                    float value, float* __restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float x = a[e];
    float r;
    switch (op) {
      case 0: r = x + b[e]; break;
      case 1: r = x > value ? 1.f : 0.f; break;
      case 2: r = x == value ? 1.f : 0.f; break;
      case 3: r = x * b[e]; break;
      case 4: r = x * (b[e] > value ? 1.f : 0.f); break;
      case 5: r = fmaxf(x, b[e]); break;
      case 6: r = (x - b[e]) > value ? 1.f : 0.f; break;
      default: r = x <= value ? 1.f : 0.f; break;
    }
    out[e] = r;
  }
}


// ---- distance passes: ball / octahedron dilation without the (2r+1)^3 convolution ---------------------------------
// dst[v] = min over |t| <= r (inside the volume) of src'[v + t e_axis] + cost(t); cost = t^2 (squared Euclidean) or |t|
// (city block).  first: src is a mask (set -> 0, unset -> BIG).  Three passes (one per axis) give the exact squared
// Euclidean / city-block distance to the mask wherever it is <= r^2 / r: thresholding it is the zero-padded conv3d with
// skimage's ball(r) > 0 (artifacts.py:484-499), respectively r repeated ball(1) dilations (artifacts.py:587-589).
constexpr float DIST_BIG = 1e9f;

__global__ __launch_bounds__(256) void dist_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, int n0, int n1,
                                                        int n2, int axis, int r, int metric, int first) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y, i = blockIdx.z;
  if (k >= n2) return;
  const size_t v = ((size_t)i * n1 + j) * n2 + k;
  const int pos = axis == 0 ? i : (axis == 1 ? j : k);
  const int len = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
  const long long stride = axis == 0 ? (long long)n1 * n2 : (axis == 1 ? n2 : 1);
  const int lo = max(-r, -pos), hi = min(r, len - 1 - pos);
  float best = DIST_BIG;
  for (int t = lo; t <= hi; ++t) {
    float s = src[(long long)v + t * stride];
    if (first) s = s > 0.f ? 0.f : DIST_BIG;
    const float c = metric == 0 ? (float)(t * t) : (float)abs(t);
    best = fminf(best, s + c);
  }
  dst[v] = best;
}

// SimulatedBoundaries, fuzzy branch (artifacts.py:565-602) fused: the reference stacks n_dilate successive dilations of
// `mask`, picks one per voxel through a one-hot of the rounded probability map and multiplies by mask_modif.
// stack[k] = {city-block distance to mask <= max(k-1, 0)}, so with k = clamp(rint(p * n_dilate - 1), 0):
//   out = image * mask_modif * (dist <= max(k - 1, 0)),  p = mog on the voxels mask_modif added to mask, 0 elsewhere.
__global__ __launch_bounds__(256) void boundary_mask_kernel(const float* __restrict__ image, const float* __restrict__ mask,
                                                            const float* __restrict__ mask_modif, const float* __restrict__ mog,
                                                            const float* __restrict__ dist, int n_dilate, size_t n,
                                                            float* __restrict__ out, float* __restrict__ mask_out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float mm = mask_modif[e];
    const bool surf = (mm - mask[e]) > 0.f;
    const float p = surf ? mog[e] : 0.f;
    int k = (int)rintf(p * (float)n_dilate - 1.f);
    if (k < 0) k = 0;
    const float thr = (float)(k > 1 ? k - 1 : 0);
    const float m = mm * (dist[e] <= thr ? 1.f : 0.f);
    if (out) out[e] = image[e] * m;
    if (mask_out) mask_out[e] = m;
  }
}

// out[e] = a[e] * (u_e < p), u from Philox(seed, stream_id): random thinning of a shell (device-RNG stand-in for
// `diff[randperm(n)[: int(0.9 n)]] = 0`, artifacts.py:515-518)
__global__ __launch_bounds__(256) void bernoulli_kernel(const float* __restrict__ a, size_t n, float p, uint64_t seed,
                                                        uint64_t stream_id, float* __restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float x = a[e];
    float r = 0.f;
    if (x != 0.f) {
      const uint64_t blk = e >> 2;
      const uint4 q = fsg_philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32),
                                        (uint32_t)seed, (uint32_t)(seed >> 32));
      const uint32_t w = (e & 3) == 0 ? q.x : ((e & 3) == 1 ? q.y : ((e & 3) == 2 ? q.z : q.w));
      r = ((float)(w >> 8) * 5.9604644775390625e-08f) < p ? x : 0.f;
    }
    out[e] = r;
  }
}

__global__ __launch_bounds__(256) void scatter_const_kernel(float* __restrict__ out, size_t n, const long long* __restrict__ idx,
                                                            int k, float value) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < k && idx[q] >= 0 && (size_t)idx[q] < n) out[idx[q]] = value;
}

}  // namespace

extern "C" {

int fsg_mog3d_f32(const float* centers, const float* sigmas, int k, int D, int H, int W, float* tables, float* out,
                  void* stream) {
  if (!centers || !sigmas || !tables || !out || k <= 0 || D <= 0 || H <= 0 || W <= 0) return FSG_E_BADARG;
  if ((size_t)D * H * W > (size_t)0x7FFFFFFF || H > 65535 || D > 65535) return FSG_E_TOOBIG;
  const int L = D > H ? (D > W ? D : W) : (H > W ? H : W);
  hipStream_t st = fsg_stream(stream);
  const int ne = k * 3 * L;
  hipLaunchKernelGGL(mog_tables_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, centers, sigmas, k, D, H, W, L,
                     tables);
  const int tx = W >= 1024 ? 256 : 64;
  const dim3 grid((unsigned)((W + 4 * tx - 1) / (4 * tx)), (unsigned)H, (unsigned)D);
  if (g_tuning_flags & FSG_TUNE_PRECISE_MATH)
    hipLaunchKernelGGL(mog_sum_kernel<false>, grid, dim3(tx), 0, st, (const float*)tables, k, D, H, W, L, out);
  else
    hipLaunchKernelGGL(mog_sum_kernel<true>, grid, dim3(tx), 0, st, (const float*)tables, k, D, H, W, L, out);
  FSG_RETURN_LAUNCH();
}

int fsg_perlin_fractal_f32(const float* const* grads, const float* const* lins, const int32_t* res, const float* amps,
                           int noct, int n0, int n1, int n2, float* out, int32_t* mm, void* stream) {
  if (!grads || !lins || !res || !amps || !out || !mm || noct <= 0 || noct > 8 || n0 <= 0 || n1 <= 0 || n2 <= 0)
    return FSG_E_BADARG;
  if ((size_t)n0 * n1 * n2 > (size_t)0x7FFFFFFF || n0 > 65535 || n1 > 65535) return FSG_E_TOOBIG;
  PerlinArgs A;
  A.noct = noct; A.n0 = n0; A.n1 = n1; A.n2 = n2;
  for (int q = 0; q < noct; ++q) {
    if (!grads[q] || !lins[q] || res[3 * q] <= 0 || res[3 * q + 1] <= 0 || res[3 * q + 2] <= 0) return FSG_E_BADARG;
    A.o[q].grad = grads[q]; A.o[q].lin = lins[q]; A.o[q].amp = amps[q];
    for (int a = 0; a < 3; ++a) A.o[q].r[a] = res[3 * q + a];
  }
  const int tx = n2 >= 256 ? 256 : (n2 > 128 ? 256 : (n2 > 64 ? 128 : 64));
  hipLaunchKernelGGL(perlin_kernel, dim3((unsigned)((n2 + tx - 1) / tx), (unsigned)n1, (unsigned)n0), dim3(tx), 0,
                     fsg_stream(stream), A, out, mm);
  FSG_RETURN_LAUNCH();
}

int fsg_blend_f32(const float* a, const float* b, const float* w, size_t n, int w_mode, const int32_t* w_mm, float increase,
                  const float* seg, int b_mode, const int32_t* b_mm, const int32_t* a_mm, float std, float* out,
                  float* w_out, void* stream) {
  if (!w || n == 0 || (!out && !w_out) || w_mode < 0 || w_mode > 1 || b_mode < 0 || b_mode > 1) return FSG_E_BADARG;
  if (out && (!a || !b)) return FSG_E_BADARG;
  if (w_mode == 1 && !w_mm) return FSG_E_BADARG;
  if (b_mode == 1 && (!b_mm || !a_mm)) return FSG_E_BADARG;
  BlendArgs B{a, b, w, w_mm, increase, seg, w_mode, b_mode, b_mm, a_mm, std, out, w_out, n};
  size_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(blend_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), B);
  FSG_RETURN_LAUNCH();
}

int fsg_slice_noise_f32(float* slices, size_t n, float threshold, float sigma, const float* noise1, const float* noise2,
                        uint64_t seed, uint64_t stream_id, void* stream) {
  if (!slices || n == 0 || ((noise1 == nullptr) != (noise2 == nullptr))) return FSG_E_BADARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(slice_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), slices, n, threshold, sigma,
                     noise1, noise2, seed, stream_id);
  FSG_RETURN_LAUNCH();
}

int fsg_slice_void_f32(float* slices, int h, int w, const int32_t* slice_ids, const float* params, int nvoid,
                       const float* ylin, const float* xlin, void* stream) {
  if (!slices || !slice_ids || !params || !ylin || !xlin || h <= 0 || w <= 0 || nvoid <= 0 || nvoid > 65535)
    return FSG_E_BADARG;
  int bx = (h * w + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(slice_void_kernel, dim3((unsigned)bx, (unsigned)nvoid), dim3(256), 0, fsg_stream(stream), slices, h, w,
                     slice_ids, params, ylin, xlin);
  FSG_RETURN_LAUNCH();
}

int fsg_slice_sums_f32(const float* slices, int n, size_t hw, float* sums, void* stream) {
  if (!slices || !sums || n <= 0 || hw == 0) return FSG_E_BADARG;
  hipLaunchKernelGGL(slice_sums_kernel, dim3((unsigned)n), dim3(1024), 0, fsg_stream(stream), slices, hw, sums);
  FSG_RETURN_LAUNCH();
}

int fsg_nonzero_count_f32(const float* v, size_t n, int mode, float value, int32_t* counts, void* stream) {
  if (!v || !counts || n == 0 || mode < 0 || mode > 2) return FSG_E_BADARG;
  hipLaunchKernelGGL(nonzero_count_kernel<float>, dim3((unsigned)((n + NZ_BLOCK - 1) / NZ_BLOCK)), dim3(256), 0,
                     fsg_stream(stream), v, n, mode, value, counts);
  FSG_RETURN_LAUNCH();
}
int fsg_nonzero_count_u8(const uint8_t* v, size_t n, int mode, float value, int32_t* counts, void* stream) {
  if (!v || !counts || n == 0 || mode < 0 || mode > 2) return FSG_E_BADARG;
  hipLaunchKernelGGL(nonzero_count_kernel<uint8_t>, dim3((unsigned)((n + NZ_BLOCK - 1) / NZ_BLOCK)), dim3(256), 0,
                     fsg_stream(stream), v, n, mode, value, counts);
  FSG_RETURN_LAUNCH();
}
int fsg_nonzero_select_f32(const float* v, size_t n, int mode, float value, const int32_t* bucket, const int32_t* rank,
                           int nreq, long long* out, void* stream) {
  if (!v || !bucket || !rank || !out || n == 0 || nreq <= 0 || mode < 0 || mode > 2) return FSG_E_BADARG;
  hipLaunchKernelGGL(nonzero_select_kernel<float>, dim3((unsigned)nreq), dim3(64), 0, fsg_stream(stream), v, n, mode, value,
                     bucket, rank, out);
  FSG_RETURN_LAUNCH();
}
int fsg_nonzero_select_u8(const uint8_t* v, size_t n, int mode, float value, const int32_t* bucket, const int32_t* rank,
                          int nreq, long long* out, void* stream) {
  if (!v || !bucket || !rank || !out || n == 0 || nreq <= 0 || mode < 0 || mode > 2) return FSG_E_BADARG;
  hipLaunchKernelGGL(nonzero_select_kernel<uint8_t>, dim3((unsigned)nreq), dim3(64), 0, fsg_stream(stream), v, n, mode, value,
                     bucket, rank, out);
  FSG_RETURN_LAUNCH();
}

int fsg_compact_f32(const float* values, const float* pred, size_t n, int mode, float value, const long long* offsets,
                    float* out, void* stream) {
  if (!values || !pred || !offsets || !out || n == 0 || mode < 0 || mode > 2) return FSG_E_BADARG;
  hipLaunchKernelGGL(compact_kernel, dim3((unsigned)((n + NZ_BLOCK - 1) / NZ_BLOCK)), dim3(64), 0, fsg_stream(stream), values,
                     pred, n, mode, value, offsets, out);
  FSG_RETURN_LAUNCH();
}

int fsg_ewise_f32(const float* a, const float* b, size_t n, int op, float value, float* out, void* stream) {
  if (!a || !out || n == 0 || op < 0 || op > 7) return FSG_E_BADARG;
  if ((op == 0 || (op >= 3 && op <= 6)) && !b) return FSG_E_BADARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(ewise_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), a, b, n, op, value, out);
  FSG_RETURN_LAUNCH();
}

int fsg_dist_pass_f32(const float* src, float* dst, int n0, int n1, int n2, int axis, int radius, int metric, int first,
                      void* stream) {
  if (!src || !dst || src == dst || n0 <= 0 || n1 <= 0 || n2 <= 0 || axis < 0 || axis > 2 || radius < 0 || radius > 1024 ||
      metric < 0 || metric > 1)
    return FSG_E_BADARG;
  if ((size_t)n0 * n1 * n2 > (size_t)0x7FFFFFFF || n0 > 65535 || n1 > 65535) return FSG_E_TOOBIG;
  const int tx = n2 > 128 ? 256 : (n2 > 64 ? 128 : 64);
  hipLaunchKernelGGL(dist_pass_kernel, dim3((unsigned)((n2 + tx - 1) / tx), (unsigned)n1, (unsigned)n0), dim3(tx), 0,
                     fsg_stream(stream), src, dst, n0, n1, n2, axis, radius, metric, first);
  FSG_RETURN_LAUNCH();
}

int fsg_boundary_mask_f32(const float* image, const float* mask, const float* mask_modif, const float* mog, const float* dist,
                          int n_dilate, size_t n, float* out, float* mask_out, void* stream) {
  if (!mask || !mask_modif || !mog || !dist || n == 0 || n_dilate <= 0 || (!out && !mask_out) || (out && !image))
    return FSG_E_BADARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(boundary_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), image, mask, mask_modif,
                     mog, dist, n_dilate, n, out, mask_out);
  FSG_RETURN_LAUNCH();
}

int fsg_bernoulli_keep_f32(const float* a, size_t n, float p, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
  if (!a || !out || n == 0) return FSG_E_BADARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(bernoulli_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), a, n, p, seed, stream_id, out);
  FSG_RETURN_LAUNCH();
}

int fsg_scatter_const_f32(float* out, size_t n, const long long* idx, int k, float value, void* stream) {
  if (!out || !idx || k <= 0 || n == 0) return FSG_E_BADARG;
  hipLaunchKernelGGL(scatter_const_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, fsg_stream(stream), out, n, idx, k,
                     value);
  FSG_RETURN_LAUNCH();
}

}  // extern "C"
