// fsg_reduce.hip -- K9/K10 stand-alone: global min/max and the three scalings of the path.
//
// Replaces `output_resized / torch.max(output_resized)` (generator/augmentation/synthseg.py:112), monai's
// ScaleIntensity(0,1) as used at data/datasets.py:40,:311, and the 0..255 image normalisation of
// generator/model.py:138 when these are called outside the fused zoom-back kernel.
#include "fsg_common.h"

namespace {

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, size_t n, int32_t* __restrict__ mm) {
  float lo = INFINITY, hi = -INFINITY;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((((uintptr_t)x) & 15) == 0) {
    const size_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (size_t g = e; g < n4; g += stride) {
      const float4 v = x4[g];
      lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
      hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (size_t g = (n4 << 2) + e; g < n; g += stride) { lo = fminf(lo, x[g]); hi = fmaxf(hi, x[g]); }
  } else {
    for (size_t g = e; g < n; g += stride) { lo = fminf(lo, x[g]); hi = fmaxf(hi, x[g]); }
  }
  __shared__ float red[2][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lo = fsg_wave_min(lo);
  hi = fsg_wave_max(hi);
  if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
    fsg_atomic_min_key(&mm[0], lo);
    fsg_atomic_max_key(&mm[1], hi);
  }
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, size_t n, const int32_t* __restrict__ mm,
                                                    int mode, float* __restrict__ out) {
  const float mn = fsg_key2f(mm[0]), mx = fsg_key2f(mm[1]);
  const float den = mx - mn;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float v = x[e];
    float r;
    if (mode == 0) r = v / mx;
    else if (mode == 1) r = (mn == mx) ? v * 0.0f : (v - mn) / den;
    else r = (v - mn) / den * 255.0f;
    out[e] = r;
  }
}

// byte copy by the compute queue: `src` may be pinned host memory (device-visible under unified addressing).  Used for
// the per-sample parameter arena (<= 64 KB): a copy *kernel* in the launch stream instead of a memcpy command keeps the
// whole sample in one queue.
__global__ __launch_bounds__(256) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (size_t)gridDim.x * blockDim.x) dst[e] = src[e];
}

// fp32 -> fp16 (round to nearest even), 8 elements per thread where alignment allows: the optional half-precision image
// of the output side (a [0,1] image keeps 11 significant bits)
__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ x, size_t n, _Float16* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t e0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const size_t n8 = n >> 3;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    h8* o8 = reinterpret_cast<h8*>(out);
    for (size_t g = e0; g < n8; g += stride) {
      const float4 a = x4[2 * g], b = x4[2 * g + 1];
      h8 h = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
      o8[g] = h;
    }
    for (size_t g = (n8 << 3) + e0; g < n; g += stride) out[g] = (_Float16)x[g];
  } else {
    for (size_t g = e0; g < n; g += stride) out[g] = (_Float16)x[g];
  }
}

}  // namespace

extern "C" {

int fsg_cast_f32_to_f16(const float* x, size_t n, void* out_f16, void* stream) {
  if (n == 0) return 0;
  if (!x || !out_f16) return FSG_E_BADARG;
  size_t blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), x, n, (_Float16*)out_f16);
  FSG_RETURN_LAUNCH();
}

int fsg_copy_bytes(void* dst, const void* src, size_t nbytes, void* stream) {
  if (!dst || !src || nbytes == 0 || (nbytes & 15) || (((uintptr_t)dst | (uintptr_t)src) & 15)) return FSG_E_BADARG;
  const size_t n16 = nbytes >> 4;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), (uint4*)dst, (const uint4*)src, n16);
  FSG_RETURN_LAUNCH();
}

int fsg_reduce_minmax_f32(const float* x, size_t n, int32_t* mm, void* stream) {
  if (!x || !mm || n == 0) return FSG_E_BADARG;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), x, n, mm);
  FSG_RETURN_LAUNCH();
}

int fsg_scale_f32(const float* x, size_t n, const int32_t* mm, int mode, float* out, void* stream) {
  if (!x || !mm || !out || n == 0 || mode < 0 || mode > 2) return FSG_E_BADARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), x, n, mm, mode, out);
  FSG_RETURN_LAUNCH();
}

float fsg_key_to_float(int32_t key) { return fsg_key2f(key); }

int fsg_abi_version(void) { return FSG_ABI_VERSION; }

const char* fsg_error_string(int code) {
  if (code == 0) return "success";
  if (code == FSG_E_BADARG) return "fsg: bad argument (null pointer, non-positive size or bad enum)";
  if (code == FSG_E_TOOBIG) return "fsg: size exceeds kernel index range";
  if (code == FSG_E_ALIGN) return "fsg: shape/alignment not supported by the fast path";
  if (code == FSG_E_NOTABLE) return "fsg: keyed mode needs a tap table that was not registered (fsg_keyed_set_table)";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "fsg: unknown error";
}

}  // extern "C"
