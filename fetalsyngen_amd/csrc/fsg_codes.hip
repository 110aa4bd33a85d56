// The seed volumes of a subject as one uint16 code volume (include/fsg_hip.h: fsg_seed_codes_build, fsg_sample_head_codes_f32).
//
// Reference: ImageFromSeeds.load_seeds (intensity/rand_gmm.py:91-99) adds up one label volume per meta label for every sample;
// the volumes a sample can select from are fixed per subject.  A voxel's "column" -- the byte every seed volume holds there --
// takes few distinct values per subject (its meta label and its sub-cluster under every sub-cluster count), so the column's
// index is a 2-byte code and a sample's label a table look-up.  This file builds codes + distinct columns in one pass: an exact
// hash set of columns in global memory (open addressing; a slot is claimed with one atomicCAS, published with its code, and
// compared word by word on a hit -- the hash only picks the first slot), sized for at most FSG_CODES_MAX columns.
#include "fsg_common.h"

namespace {

constexpr int SC_SLOTS = 8192;      // power of two, 4 x FSG_CODES_MAX
constexpr int SC_WORDS = 16;        // a column padded to 64 bytes (nparts <= 64)
constexpr int SC_MAXPARTS = 64;

struct CodesK {
  const uint8_t* part[SC_MAXPARTS];
  int nparts, nwords, stride, cap;
  uint32_t n;
  uint16_t* codes;
  uint8_t* tuples;     // [cap][stride], zero-initialised by the host side
  int32_t* state;      // [SC_SLOTS] 0 empty, 1 being written, 2 ready
  int32_t* slot_code;  // [SC_SLOTS]
  uint32_t* slot_col;  // [SC_SLOTS][SC_WORDS]
  int32_t* count;      // distinct columns so far (may run past cap: then the result is unusable and the caller is told)
};

// Global protocol for one column (exact; see the file header).  Returns its code, or -1 when the table is filling up.
__device__ __forceinline__ int sc_global_code(const CodesK& P, const uint32_t* col, uint32_t h) {
  uint32_t slot = h & (SC_SLOTS - 1);
  // every lane completes whatever it starts inside ONE trip of this loop (the lane that claims a slot writes and publishes it
  // before the trip ends), so lanes of one wave that wait for each other's slot cannot dead-lock
  for (int guard = 0; guard < 64 * SC_SLOTS; ++guard) {
    const int st = __hip_atomic_load(&P.state[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (st == 2) {
      bool same = true;
#pragma unroll
      for (int w = 0; w < SC_WORDS; ++w)
        if (w < P.nwords)
          same = same && __hip_atomic_load(&P.slot_col[(size_t)slot * SC_WORDS + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == col[w];
      if (same) return __hip_atomic_load(&P.slot_code[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      slot = (slot + 1) & (SC_SLOTS - 1);
    } else if (st == 0) {
      if (__hip_atomic_load(P.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > SC_SLOTS / 2) return -1;  // filling up: give up
      if (atomicCAS(&P.state[slot], 0, 1) == 0) {
        const int c = atomicAdd(P.count, 1);
#pragma unroll
        for (int w = 0; w < SC_WORDS; ++w)
          if (w < P.nwords) __hip_atomic_store(&P.slot_col[(size_t)slot * SC_WORDS + w], col[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&P.slot_code[slot], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c < P.cap)
#pragma unroll
          for (int j = 0; j < SC_MAXPARTS; ++j)
            if (j < P.nparts) P.tuples[(size_t)c * P.stride + j] = (uint8_t)(col[j >> 2] >> (8 * (j & 3)));
        __hip_atomic_store(&P.state[slot], 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return c;
      }
    }  // st == 1: another lane is writing this slot -- look again
  }
  atomicAdd(P.count, 2 * SC_SLOTS);  // (unreachable in a table at most half full; if it ever is, the host sees count > cap and keeps the volumes)
  return -1;
}

// A workgroup keeps the columns it has resolved in LDS (a subject has tens of distinct columns, a workgroup meets a handful): the
// global hash set then sees ~one look-up per (workgroup, column) instead of one per voxel (6.6 ms -> well under 1 ms at 256^3).
// LDS entries are claimed with a CAS on their code word (-1 empty, -2 being written, >= 0 the code) and published with a
// workgroup fence; a reader that meets anything but a published entry just takes the global path.
constexpr int SC_LOCAL = 256;

// resolve one column to its code: the workgroup's LDS cache first, the global hash set on a miss
__device__ __forceinline__ int sc_resolve(const CodesK& P, const uint32_t* col, uint32_t (*l_col)[SC_WORDS], int* l_code) {
  uint32_t h = 2166136261u;
#pragma unroll
  for (int w = 0; w < SC_WORDS; ++w) {
    if (w < P.nwords) {
      h = (h ^ col[w]) * 16777619u;
      h ^= h >> 15;
    }
  }
  uint32_t ls = (h >> 13) & (SC_LOCAL - 1);
  int free_slot = -1;
  for (int probe = 0; probe < 4; ++probe) {
    const int lc = __hip_atomic_load(&l_code[ls], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lc >= 0) {
      bool same = true;
#pragma unroll
      for (int w = 0; w < SC_WORDS; ++w)
        if (w < P.nwords) same = same && l_col[ls][w] == col[w];
      if (same) return lc;
    } else if (lc == -1 && free_slot < 0) {
      free_slot = (int)ls;
    }
    ls = (ls + 1) & (SC_LOCAL - 1);
  }
  const int code = sc_global_code(P, col, h);
  if (code >= 0 && free_slot >= 0 && atomicCAS(&l_code[free_slot], -1, -2) == -1) {
#pragma unroll
    for (int w = 0; w < SC_WORDS; ++w)
      if (w < P.nwords) l_col[free_slot][w] = col[w];
    __hip_atomic_store(&l_code[free_slot], code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  return code;
}

// Four consecutive voxels per lane: one 4-byte load per seed volume (a wave's request is 256 bytes instead of 64), four
// columns resolved one after the other, one 8-byte store of their codes.  n % 4 == 0 and 4-byte aligned volumes (the launcher
// falls back to the one-voxel form otherwise).
template <int VPL>
__global__ __launch_bounds__(256) void seed_codes_kernel(const CodesK P) {
  __shared__ uint32_t l_col[SC_LOCAL][SC_WORDS];
  __shared__ int l_code[SC_LOCAL];
  for (int e = threadIdx.x; e < SC_LOCAL; e += 256) l_code[e] = -1;
  __syncthreads();
  const uint32_t ng = P.n / VPL;
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < ng; g += gridDim.x * 256u) {
    uint32_t col[VPL][SC_WORDS];
#pragma unroll
    for (int q = 0; q < VPL; ++q)
#pragma unroll
      for (int w = 0; w < SC_WORDS; ++w) col[q][w] = 0u;
#pragma unroll
    for (int j = 0; j < SC_MAXPARTS; ++j) {
      if (j < P.nparts) {  // uniform
        uint32_t word;
        if (VPL == 4) word = *reinterpret_cast<const uint32_t*>(P.part[j] + (size_t)g * 4u);
        else word = P.part[j][g];
#pragma unroll
        for (int q = 0; q < VPL; ++q) col[q][j >> 2] |= ((word >> (8 * q)) & 255u) << (8 * (j & 3));
      }
    }
    uint32_t packed[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      const int code = sc_resolve(P, col[q], l_col, l_code);
      const uint32_t c16 = (uint32_t)(code < 0 || code >= P.cap ? 0 : code);
      packed[q >> 1] |= c16 << (16 * (q & 1));
    }
    if (VPL == 4) *reinterpret_cast<uint2*>(P.codes + (size_t)g * 4u) = make_uint2(packed[0], packed[1]);
    else P.codes[g] = (uint16_t)packed[0];
  }
}

}  // namespace

extern "C" {

size_t fsg_seed_codes_work_bytes(void) { return (size_t)SC_SLOTS * (2 * sizeof(int32_t) + SC_WORDS * sizeof(uint32_t)) + 64; }

int fsg_seed_codes_build(const uint8_t* const* parts, int nparts, size_t n, int stride, uint16_t* codes, uint8_t* tuples, int cap,
                         void* work, size_t work_bytes, int32_t* count_dev, void* stream) {
  if (!parts || nparts <= 0 || nparts > SC_MAXPARTS || n == 0 || !codes || !tuples || !work || !count_dev) return FSG_E_BADARG;
  if (stride <= nparts || stride > 256 || cap <= 0 || cap > FSG_CODES_MAX) return FSG_E_BADARG;
  if (n > ((size_t)1 << 30)) return FSG_E_TOOBIG;
  if (work_bytes < fsg_seed_codes_work_bytes() || ((uintptr_t)work & 15)) return FSG_E_BADARG;
  for (int j = 0; j < nparts; ++j)
    if (!parts[j]) return FSG_E_BADARG;
  hipStream_t st = fsg_stream(stream);
  hipError_t e = hipMemsetAsync(work, 0, fsg_seed_codes_work_bytes(), st);
  if (e == hipSuccess) e = hipMemsetAsync(tuples, 0, (size_t)cap * stride, st);
  if (e == hipSuccess) e = hipMemsetAsync(count_dev, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  CodesK P;
  for (int j = 0; j < SC_MAXPARTS; ++j) P.part[j] = j < nparts ? parts[j] : nullptr;
  P.nparts = nparts; P.nwords = (nparts + 3) / 4; P.stride = stride; P.cap = cap;
  P.n = (uint32_t)n;
  P.codes = codes; P.tuples = tuples;
  char* w = (char*)work;
  P.state = (int32_t*)w;
  P.slot_code = (int32_t*)(w + (size_t)SC_SLOTS * sizeof(int32_t));
  P.slot_col = (uint32_t*)(w + (size_t)SC_SLOTS * 2 * sizeof(int32_t));
  P.count = count_dev;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  bool quad = (n & 3) == 0 && (((uintptr_t)codes) & 7) == 0;
  for (int j = 0; j < nparts; ++j) quad = quad && (((uintptr_t)parts[j]) & 3) == 0;
  if (quad) {
    blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(seed_codes_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, P);
  } else {
    hipLaunchKernelGGL(seed_codes_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, P);
  }
  FSG_RETURN_LAUNCH();
}

}  // extern "C"
