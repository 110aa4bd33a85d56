// Microbenchmark: issue cost of the vector ALU instructions the Philox + Box-Muller draw is made of, per SIMD, at full
// occupancy (8 waves per SIMD, 4 independent chains per wave).  cycles per wave-instruction at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[4];
  float f[4];
  for (int c = 0; c < 4; ++c) { a[c] = seed + threadIdx.x * 4 + c; f[c] = 1.0f + (float)(threadIdx.x + c) * 1e-3f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (OP == 0) { const uint64_t p = (uint64_t)0xD2511F53u * a[c]; a[c] = (uint32_t)(p >> 32) ^ (uint32_t)p; }          // v_mad_u64_u32 + xor
      if (OP == 1) { a[c] = __umulhi(a[c], 0xD2511F53u); }                                                               // v_mul_hi_u32
      if (OP == 2) { a[c] = a[c] * 0xD2511F53u; }                                                                        // v_mul_lo_u32
      if (OP == 3) { a[c] = a[c] ^ (a[c] >> 7) ^ seed; }                                                                 // shifts / xors
      if (OP == 4) { f[c] = __builtin_amdgcn_logf(f[c]) + 2.0f; }
      if (OP == 5) { f[c] = __builtin_amdgcn_sinf(f[c]) + 1.5f; }
      if (OP == 6) { f[c] = __builtin_amdgcn_sqrtf(f[c]) + 1.0f; }
      if (OP == 7) { f[c] = f[c] * 1.0001f + 0.5f; }                                                                     // mul + add
      if (OP == 8) { a[c] = (uint32_t)__mul24((int)a[c], 0x511F53) + seed; }                                             // v_mul_u32_u24
    }
  }
  uint32_t r = 0;
  for (int c = 0; c < 4; ++c) r ^= a[c] ^ __builtin_bit_cast(uint32_t, f[c]);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4096;
  const char* names[] = {"v_mad_u64_u32 (+ v_xor)", "v_mul_hi_u32", "v_mul_lo_u32", "shift + 2 xor", "v_log_f32 (+ add)", "v_sin_f32 (+ add)",
                         "v_sqrt_f32 (+ add)", "v_mul_f32 + v_add_f32", "v_mul_u32_u24 (+ add)"};
  const int extra[] = {1, 0, 0, 2, 1, 1, 1, 1, 1};  // full-rate companions per op, subtracted at 4 cycles each
  for (int op = 0; op < 9; ++op) {
    auto run = [&]() {
      switch (op) {
#define C(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8)
#undef C
      }
    };
    run();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) run();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 3;
    const double groups_per_simd = 8.0 * iters * 4;  // (op + companions) groups issued per SIMD
    const double cyc = us * 2400.0 / groups_per_simd;
    printf("%-28s %7.2f cycles per group  (-> %5.1f for the op alone if companions cost 4 each)\n", names[op], cyc, cyc - 4.0 * extra[op]);
  }
  return 0;
}
