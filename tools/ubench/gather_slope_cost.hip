// Microbenchmark: TA/TCP cost of the fused warp's gather shapes when the lanes of a wave walk a SLANTED line through the source
// (rotation): 2 output rows x 32 lanes, lane l samples column c0 + l*step and row r0 + floor(l*step*slope); rows are 1 KB apart.
// Compares, per 64 output voxels: dword gather (labels), 8-byte gather per voxel (today's image corners), 16-byte gather per
// TWO voxels (a lane owning two consecutive outputs).  Data L1/L2 resident (16 KB window per workgroup, read-only).
//   hipcc --offload-arch=gfx950 -O3 gather_slope_cost.hip -o gather_slope_cost && ./gather_slope_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int BYTES, int STEP>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ buf, int iters, float slope, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* w = buf + (size_t)blockIdx.x * 4096;  // 16 rows x 256 floats
  const int l = lane & 31, rj = lane >> 5;
  const int drow = (int)floorf((float)(l * STEP) * slope);
  float acc = 0.f;
  int sel = 0;
  for (int it = 0; it < iters; ++it) {
    const int r0 = (it * 3 + wave + sel) & 15, c0 = (it * 7 + wave * 5) & 63;
    const int row = (r0 + rj + drow) & 15, col = c0 + l * STEP;  // col + 3 <= 63 + 62 + 3 < 256
    const float* p = w + row * 256 + col;
    if (BYTES == 4) acc += *p;
    if (BYTES == 8) { f2u v = *reinterpret_cast<const f2u*>(p); acc += v.x + v.y; }
    if (BYTES == 16) { f4u v = *reinterpret_cast<const f4u*>(p); acc += v.x + v.w; }
    sel = (acc == 1234.5f) ? 1 : 0;
  }
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main() {
  const int blocks = 512;
  float *buf, *out;
  hipMalloc(&buf, (size_t)blocks * 4096 * 4);
  hipMalloc(&out, (size_t)blocks * 1024 * 4);
  hipMemset(buf, 0, (size_t)blocks * 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2048;
  const float slopes[] = {0.f, 0.1f, 0.2f, 0.34f};
  for (float slope : slopes) {
    for (int shape = 0; shape < 4; ++shape) {
      auto run = [&]() {
        switch (shape) {
          case 0: hipLaunchKernelGGL((k<4, 1>), dim3(blocks), dim3(1024), 0, 0, buf, iters, slope, out); break;
          case 1: hipLaunchKernelGGL((k<8, 1>), dim3(blocks), dim3(1024), 0, 0, buf, iters, slope, out); break;
          case 2: hipLaunchKernelGGL((k<16, 2>), dim3(blocks), dim3(1024), 0, 0, buf, iters, slope, out); break;
          case 3: hipLaunchKernelGGL((k<8, 2>), dim3(blocks), dim3(1024), 0, 0, buf, iters, slope, out); break;
        }
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
      const double winstr_per_cu = (double)blocks * 16 * iters / 256.0;
      const double cyc = us * 2400.0 / winstr_per_cu;
      const char* names[] = {"4 B per lane, 1 voxel/lane", "8 B per lane, 1 voxel/lane", "16 B per lane, 2 voxels/lane", "8 B per lane, lane step 2"};
      const double vox = (shape == 2) ? 128.0 : 64.0;
      printf("slope %.2f  %-30s %6.1f cyc/wave-instr/CU  = %6.1f cyc per 64 voxels\n", slope, names[shape], cyc, cyc * 64.0 / vox);
    }
  }
  return 0;
}
