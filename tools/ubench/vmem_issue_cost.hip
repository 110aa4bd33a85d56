// Microbenchmark: what one vector-memory wave-instruction costs the CU's L1 path (TA/TCP) on MI355X when its data is
// L1/L2 resident -- the regime of the fused warp kernel, whose TCP_TOTAL_CACHE_ACCESSES (222 per 64 voxels) track its
// run time.  Per access shape: cycles per wave-instruction per CU at 32 waves/CU.
//   hipcc --offload-arch=gfx950 -O3 vmem_issue_cost.hip -o vmem_issue_cost && ./vmem_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* __restrict__ buf, int iters, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 4 KB private window per wave (16 waves -> 64 KB per workgroup: L2 resident, partly L1)
  float* w = buf + ((size_t)blockIdx.x * 16 + wave) * 1024;
  unsigned char* wb = reinterpret_cast<unsigned char*>(w);
  float acc = 0.f;
  int sel = 0;
  for (int it = 0; it < iters; ++it) {
    const int o = ((it * 5 + sel) & 7) * 64;  // float offset of this iteration's 256-B slab inside the window
    if (MODE == 0) acc += w[o + lane];
    if (MODE == 1) { f2u v = *reinterpret_cast<const f2u*>(w + o + lane); acc += v.x + v.y; }
    if (MODE == 2) { f2u v = *reinterpret_cast<const f2u*>(w + ((o + lane * 2) & 1022)); acc += v.x + v.y; }
    if (MODE == 3) { f4a v = *reinterpret_cast<const f4a*>(w + ((o + lane) & ~3)); acc += v.x + v.w; }
    if (MODE == 4) { f4a v = *reinterpret_cast<const f4a*>(w + ((o + lane * 4) & 1020)); acc += v.x + v.w; }
    if (MODE == 5) w[o + lane] = acc + (float)it;
    if (MODE == 6) { f4a v = {acc, acc, acc, (float)it}; *reinterpret_cast<f4a*>(w + ((o + lane * 4) & 1020)) = v; }
    if (MODE == 7) { if (lane < 16) { f4a v = {acc, acc, acc, (float)it}; *reinterpret_cast<f4a*>(w + o + lane * 4) = v; } }
    if (MODE == 8) acc += (float)wb[o * 4 + lane];
    if (MODE == 9) acc += w[((o + (lane & 31) + (lane >> 5) * 512)) & 1023];
    if (MODE == 10) { f2u v = *reinterpret_cast<const f2u*>(w + ((o + (lane & 31) + (lane >> 5) * 512) & 1021)); acc += v.x + v.y; }
    if (MODE == 11) { if (lane < 16) acc += w[o + lane]; }
    if (MODE == 12) { f2u v = *reinterpret_cast<const f2u*>(w + ((o + (lane & 15) + (lane >> 4) * 256) & 1021)); acc += v.x + v.y; }
    sel = (acc == 1234.5f) ? 1 : 0;  // data dependence: the load cannot be hoisted or merged
  }
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main() {
  const int blocks = 512;  // 2 workgroups of 16 waves per CU
  float *buf, *out;
  hipMalloc(&buf, (size_t)blocks * 16 * 1024 * 4);
  hipMalloc(&out, (size_t)blocks * 1024 * 4);
  hipMemset(buf, 0, (size_t)blocks * 16 * 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2048;
  const char* names[] = {"dword load, lanes consecutive (256 B)", "b64 load, lane stride 4 B (overlapping, odd lanes misaligned)",
                         "b64 load, lane stride 8 B (aligned, 512 B)", "b128 load at addr & ~15, lane stride 4 B (4 lanes share 16 B)",
                         "b128 load, lane stride 16 B (1 KB)", "dword store, lanes consecutive", "b128 store, 64 lanes (1 KB)",
                         "b128 store, 16 active lanes (256 B)", "byte load, lanes consecutive (64 B)",
                         "dword load, 2 rows x 32 lanes", "b64 load stride 4 B, 2 rows x 32 lanes", "dword load, 16 active lanes",
                         "b64 load stride 4 B, 4 rows x 16 lanes"};
  for (int mode = 0; mode <= 12; ++mode) {
    auto run = [&]() {
      switch (mode) {
#define C(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(1024), 0, 0, buf, iters, out); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
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
    printf("mode %2d %-62s %8.1f us  %6.1f cyc/wave-instr/CU (at 2.4 GHz)\n", mode, names[mode], us, us * 2400.0 / winstr_per_cu);
  }
  return 0;
}
