// Microbenchmark: streaming rates of the shapes the GMM draw is made of, 256^3 (16.7 M voxels), no arithmetic to speak of.
//   A  read 4 x 16.7 MB (uint32 per lane), write nothing (one float per workgroup)
//   B  read 4 x 16.7 MB (uint4 per lane), write nothing
//   C  write 67 MB (float4 per lane), read nothing
//   D  read 4 x 16.7 MB (uint32 per lane) + write 67 MB (float4 per lane)      <- the GMM draw without its arithmetic
//   E  read 1 x 67 MB (float4 per lane) + write 67 MB (float4 per lane)        <- a copy
//   F  D + the mu / sigma tables in LDS (filled per workgroup, barrier) and the per-voxel table reads
//   G  D + one Philox4x32-10 block and Box-Muller per group (no tables)
//   H  F + G = the GMM draw
//   hipcc --offload-arch=gfx950 -O3 stream_shapes.hip -o stream_shapes && ./stream_shapes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../fetalsyngen_amd/csrc/fsg_common.h"

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* l0, const uint8_t* l1, const uint8_t* l2, const uint8_t* l3, const float4* src,
                                         float4* out, size_t n, float* sink) {
  const size_t ngrp = n >> 2, stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  __shared__ float s_mu[256], s_sg[256];
  if (MODE == 5 || MODE == 7) {
    s_mu[threadIdx.x] = threadIdx.x < 50 ? reinterpret_cast<const float*>(src)[threadIdx.x] + 30.f : 0.f;
    s_sg[threadIdx.x] = threadIdx.x < 50 ? reinterpret_cast<const float*>(src)[256 + threadIdx.x] + 5.f : 0.f;
    __syncthreads();
  }
  if (MODE == 1) {
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < (n >> 4); b += stride) {
      const uint4 a = *reinterpret_cast<const uint4*>(l0 + (b << 4)), c = *reinterpret_cast<const uint4*>(l1 + (b << 4));
      const uint4 d = *reinterpret_cast<const uint4*>(l2 + (b << 4)), e = *reinterpret_cast<const uint4*>(l3 + (b << 4));
      acc += (float)(a.x + c.y + d.z + e.w);
    }
  } else {
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < ngrp; g += stride) {
      uint32_t w = 0;
      if (MODE == 0 || MODE == 3) {
        w = *reinterpret_cast<const uint32_t*>(l0 + (g << 2)) + *reinterpret_cast<const uint32_t*>(l1 + (g << 2)) +
            *reinterpret_cast<const uint32_t*>(l2 + (g << 2)) + *reinterpret_cast<const uint32_t*>(l3 + (g << 2));
      }
      if (MODE == 0) acc += (float)w;
      if (MODE == 2) out[g] = make_float4((float)g, 1.f, 2.f, 3.f);
      if (MODE == 3) out[g] = make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24));
      if (MODE == 4) out[g] = src[g];
      if (MODE >= 5) {
        w = *reinterpret_cast<const uint32_t*>(l0 + (g << 2)) + *reinterpret_cast<const uint32_t*>(l1 + (g << 2)) +
            *reinterpret_cast<const uint32_t*>(l2 + (g << 2)) + *reinterpret_cast<const uint32_t*>(l3 + (g << 2));
        float z[4] = {0.5f, 0.25f, 0.125f, 1.f};
        if (MODE == 6 || MODE == 7) { const float4 r = fsg_randn4(7, 1, (uint64_t)g); z[0] = r.x; z[1] = r.y; z[2] = r.z; z[3] = r.w; }
        float v[4];
        for (int q = 0; q < 4; ++q) {
          const int l = (int)((w >> (8 * q)) & 255u);
          const float t = (MODE == 6) ? (float)l + z[q] : s_mu[l] + s_sg[l] * z[q];
          v[q] = t < 0.f ? 0.f : t;
        }
        out[g] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (MODE <= 1 && acc == 1234.5f) sink[blockIdx.x] = acc;
}

int main() {
  const size_t n = (size_t)256 * 256 * 256;
  uint8_t* l[4];
  for (int i = 0; i < 4; ++i) { hipMalloc(&l[i], n); hipMemset(l[i], i + 1, n); }  // labels 1+2+3+4 = 10 everywhere
  float4 *src, *out;
  float* sink;
  hipMalloc(&src, n * 4); hipMalloc(&out, n * 4); hipMalloc(&sink, 1 << 20);
  hipMemset(src, 0, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"A read 4 x 16.7 MB, u32 per lane", "B read 4 x 16.7 MB, uint4 per lane", "C write 67 MB", "D read 4 x 16.7 MB + write 67 MB",
                         "E copy 67 MB -> 67 MB", "F D + LDS tables", "G D + Philox/Box-Muller", "H D + tables + Philox (the draw)"};
  const double mb[] = {67.1, 67.1, 67.1, 134.2, 134.2, 134.2, 134.2, 134.2};
  for (int grid : {2048, 8192}) {
    for (int mode = 0; mode < 8; ++mode) {
      auto run = [&]() {
        switch (mode) {
#define C(M) case M: hipLaunchKernelGGL(k<M>, dim3(grid), dim3(256), 0, 0, l[0], l[1], l[2], l[3], src, out, n, sink); break;
          C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
#undef C
        }
      };
      for (int r = 0; r < 3; ++r) run();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 20; ++r) run();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / 20;
      printf("grid %5d  %-36s %6.1f us  %5.2f TB/s\n", grid, names[mode], us, mb[mode] / us);
    }
  }
  return 0;
}
