// Microbenchmark: cost of a wave64 gather instruction on MI355X as a function of how many cache lines the
// 64 lanes touch, the access width and the occupancy.  hipcc --offload-arch=gfx950 -O3 gather_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));

template <int W>  // W = floats per lane per load (1, 2, 4)
__global__ __launch_bounds__(256) void gather(const float* __restrict__ src, size_t nfloats, int lstride, int iters,
                                              float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  size_t base = (wave * 9973u * 64u) % (nfloats - (size_t)64 * lstride * 8 - 64);
  base &= ~(size_t)3;
  float acc = 0.f;
#pragma unroll 4
  for (int it = 0; it < iters; ++it) {
    const size_t a = base + (size_t)lane * lstride + (size_t)((it * 7919) & 0xFFFF) * 4;
    const size_t aa = a % (nfloats - 8);
    if (W == 1) acc += src[aa];
    else if (W == 2) { float2_u v = *reinterpret_cast<const float2_u*>(src + aa); acc += v.x + v.y; }
    else { float4 v = *reinterpret_cast<const float4*>(src + (aa & ~(size_t)3)); acc += v.x + v.y + v.z + v.w; }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const size_t nfloats = (size_t)16 << 20;  // 64 MiB source (L3 resident like the 256^3 volume)
  float *src, *out;
  hipMalloc(&src, nfloats * 4);
  hipMalloc(&out, (size_t)8192 * 256 * 4);
  hipMemset(src, 0, nfloats * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 256;
  for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
    for (int w : {1, 2, 4}) {
      for (int ls : {1, 2, 4, 8, 16, 33, 65, 257}) {
        auto run = [&]() {
          if (w == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, src, nfloats, ls, iters, out);
          else if (w == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, src, nfloats, ls, iters, out);
          else hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 0, 0, src, nfloats, ls, iters, out);
        };
        run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) run();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 5;
        const double winstr_per_cu = (double)blocks * 4 * iters / 256.0;
        const double cyc = us * 2400.0 / winstr_per_cu;  // cycles per wave-instruction per CU at 2.4 GHz
        const int lines = (ls * 64 * 4 + 127) / 128 + 1;
        printf("waves/CU=%2d width=%dB lane_stride=%3d floats (~%3d lines): %8.1f us  %6.1f cyc/wave-instr/CU  %7.1f GB/s useful\n",
               blocks * 4 / 256, w * 4, ls, lines > 64 ? 64 : lines, us, cyc, (double)blocks * 256 * iters * w * 4 / us / 1e3);
      }
    }
  }
  return 0;
}
