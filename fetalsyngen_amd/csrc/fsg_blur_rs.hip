// fsg_blur_rs.hip -- K6 + K7 (+ K8) fused per axis: Gaussian blur and axis-aligned down-sampling as ONE operator per axis.
//
// Reference (paths relative to /root/reference/fetalsyngen/): RandResample.__call__ = gaussian_blur_3d (utils/generation.py:
// 84-110: conv3d along x, then y, then z, zero padded, no border renormalisation) followed by fast_3D_interp_torch on an
// axis-aligned grid (generator/augmentation/synthseg.py:84-104; lerp chain x, then y, then z, strict > 0 validity:
// utils/generation.py:227-278), then RandNoise at the low resolution (synthseg.py:230-233).
// Both steps are separable linear operators, so  L_z L_y L_x B_z B_y B_x  =  (L_z B_z)(L_y B_y)(L_x B_x)  in exact arithmetic:
// operators of different axes commute.  Per axis, output j is
//        out[j] = w_lo[j] * B[lo[j]] + w_hi[j] * B[hi[j]],     B[i] = sum_t k[t] * in[i + t - R]   (zeros outside),
// so the blurred row B[i] of an axis is only ever needed by the one or two outputs that reference it: it can live in
// registers for the few instructions between its last tap and the lerp, and the blurred full-resolution volume never exists:
//        unfused (r02):  blur x 8N + blur y,z 8N + K7 (4N + 4M)            = 21 B/voxel at mu = 0.3
//        here:           x pass 4N + 4N (m/n)  +  y,z pass 4N (m/n) + 4M   = 10.5 B/voxel
// Rounding differs from the reference's order by a few ulp of the 0..255 values (the blur's own tolerance is atol 1e-3:
// conv3d's summation order is unspecified); quirks kept: zero-padded un-renormalised borders, an output whose position is
// outside (0, n-1] is 0 (lo < 0 in the table), noise added after, negatives clamped.
//
// Per axis the arithmetic is the reference's own: B[i] accumulated over ascending taps (fmaf, as fsg_blur.hip), then
// w_lo * B[lo] + w_hi * B[hi] with separate multiplies and an add (fsg_mix, as fsg_zoom.hip); only the order ACROSS axes differs.
//
//   blur_rs_x_kernel<R>  : axis 0.  Thread = one float4 column and a chunk of TL = 16 input rows: the body of blur_strided_v4
//                          (TL + 1 blurred rows from TL + 1 + 2R coalesced row loads, register sliding window, everything
//                          unrolled), then the outputs whose lower neighbour lies in the chunk are emitted straight from those
//                          registers -- the row index is wave-uniform, so "register lo - l0" is a uniform switch (scalar
//                          branches), not a gather.  The first output of a chunk is found by a ballot over the tap table.
//   blur_rs_yz_kernel<R> : axes 1 + 2 of one x-plane and 16 input y rows: the same body along y (4 + 1 blurred rows per wave)
//                          emitting its output rows into LDS, then per row the z blur on aligned ds_read_b128 windows into a
//                          second LDS buffer, then the z lerp + noise + clamp over the tile's contiguous output range (one
//                          Philox block per aligned quad of the flat output index, exactly K7's indexing), 16-byte stores.
#include "fsg_common.h"

namespace {

constexpr int RS_MAXR = 8;

__device__ __forceinline__ void rs_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int RS_KCAP = 2 * RS_MAXR + 1;
struct TapsK { float w[RS_KCAP + 3]; };


// First output j (of m) whose lower neighbour lo[j] >= l0, for a table whose lo is non-decreasing in j (every plain
// resampling table: positions delta + j * n / m).  Outputs with lo < 0 ("outside": value 0) sort first.  Wave-uniform
// result; `guess` may be anything, a good one makes it one ballot.
__device__ __forceinline__ int rs_first_output(const fsg_tap* __restrict__ tab, int m, int l0, int guess) {
  const int lane = threadIdx.x & 63;
  int jg = min(max(guess, 0), m);
  while (jg > 0 && __builtin_amdgcn_readfirstlane(tab[jg - 1].lo) >= l0) jg = max(jg - 48, 0);  // guess too high
  for (;;) {
    const int j = jg + lane;
    const bool hit = j >= m || tab[j].lo >= l0;
    const unsigned long long b = __ballot(hit);
    if (b) return jg + (int)__builtin_ctzll(b);
    jg += 64;
  }
}

__device__ __forceinline__ float4 rs_mix4(float wl, const float4& a, float wh, const float4& b) {
  return make_float4(fsg_mix(wl, a.x, wh, b.x), fsg_mix(wl, a.y, wh, b.y), fsg_mix(wl, a.z, wh, b.z), fsg_mix(wl, a.w, wh, b.w));
}

// The taps of the (at most 64) outputs that follow `jb`, one per lane; output q's fields are read back with v_readlane.
struct LaneTaps { int lo, hi; float wl, wh; };
__device__ __forceinline__ LaneTaps rs_lane_taps(const fsg_tap* __restrict__ tab, int m, int jb) {
  const int j = jb + (int)(threadIdx.x & 63);
  LaneTaps t{0x7FFFFFFF, 0x7FFFFFFF, 0.f, 0.f};
  if (j < m) {
    const int4 v = *reinterpret_cast<const int4*>(tab + j);
    t.lo = v.x; t.hi = v.y; t.wl = __builtin_bit_cast(float, v.z); t.wh = __builtin_bit_cast(float, v.w);
  }
  return t;
}
// One look-up for a chunk [l0, l1): the candidates jg .. jg + 63 are loaded once (one tap per lane, rs_chunk_guess +
// rs_lane_taps: issued early, consumed after the blur); jb = first output with lo >= l0, je = first with lo >= l1 (or m), and
// the chunk's taps are lanes (jb - jg) .. of the same registers.  Falls back to the searching routine when the guess window
// misses (never for plain down-sampling tables: a chunk of <= 16 rows holds <= 17 outputs and the guess is within a few
// outputs of the truth).
struct ChunkTaps { LaneTaps t; int jb, je, shift; };
__device__ __forceinline__ int rs_chunk_guess(int m, int l0, float f) { return min(max((int)(((float)l0 + 0.5f) * f - 0.5f) - 6, 0), m); }
__device__ __forceinline__ ChunkTaps rs_chunk_resolve(const fsg_tap* __restrict__ tab, int m, int l0, int l1, int jg, const LaneTaps& t) {
  const int lane = threadIdx.x & 63;
  ChunkTaps c;
  c.t = t;
  const unsigned long long b0 = __ballot(c.t.lo >= l0 || jg + lane >= m), b1 = __ballot(c.t.lo >= l1 || jg + lane >= m);
  const bool ok = b0 && b1 && (jg == 0 || !(b0 & 1ull));  // lane 0 already >= l0 with jg > 0: the guess may be too high
  if (ok) {
    c.jb = jg + (int)__builtin_ctzll(b0);
    c.je = jg + (int)__builtin_ctzll(b1);
    c.shift = c.jb - jg;
  } else {
    c.jb = rs_first_output(tab, m, l0, jg);
    c.je = rs_first_output(tab, m, l1, c.jb);
    c.t = rs_lane_taps(tab, m, c.jb);
    c.shift = 0;
  }
  return c;
}
__device__ __forceinline__ int rs_rl(int v, int q) { return __builtin_amdgcn_readlane(v, q); }
__device__ __forceinline__ float rs_rl(float v, int q) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), q));
}

// A row of the blur's input through a raw buffer descriptor: a row outside the axis (negative offset -> wraps past the
// descriptor's size, or beyond its end) comes back as zeros from the range check -- the blur's zero padding without a clamped
// address and four selects per row (descriptor sizes < 2^31 bytes: fsg_blur_resample_supported).
// (the builtin's result is cast as a whole: element access on it directly compiles to a ONE-dword load, hipcc 7.2)
typedef float rs_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rs_row_load(__amdgpu_buffer_rsrc_t r, unsigned byte_offset) {
  const rs_f4 v = __builtin_bit_cast(rs_f4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---- axis 0 ------------------------------------------------------------------------------------------------------------
constexpr int RSX_TL = 16;

template <int R>
__global__ __launch_bounds__(256) void blur_rs_x_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n, int m,
                                                        int inner4, const fsg_tap* __restrict__ tab, TapsK K) {
  constexpr int TL = RSX_TL, NB = TL + 1;  // NB blurred rows l0 .. l0 + TL: the last one only ever serves as an upper neighbour
  const int tx = threadIdx.x, ty = __builtin_amdgcn_readfirstlane((int)threadIdx.y);  // (a wave = one row of the block: uniform, and the compiler should know)
  const int c = blockIdx.x * 64 + tx;
  const int l0 = (blockIdx.y * 4 + ty) * TL;
  if (l0 >= n) return;  // whole wave
  const bool live = c < inner4;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)n * (unsigned)inner4 * 16u, 0x00020000);
  const unsigned colb = (unsigned)(live ? c : 0) * 16u, rowb = (unsigned)inner4 * 16u;
  const float fr = (float)m / (float)n;
  const int jg = rs_chunk_guess(m, l0, fr);
  const LaneTaps lt0 = rs_lane_taps(tab, m, jg);  // requested now, looked at after the blur
  float4 acc[NB];
#pragma unroll
  for (int o = 0; o < NB; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < NB + 2 * R; ++t) {
    const int l = l0 + t - R;
    // unconditional load (a predicated one hides the number of loads in flight from the compiler, which then waits for all
    // of them before every use); rows outside the axis read as zeros (rs_row_load)
    const float4 v = rs_row_load(rsrc, (unsigned)l * rowb + colb);
#pragma unroll
    for (int o = 0; o < NB; ++o) {
      const int tap = t - o;
      if (tap >= 0 && tap <= 2 * R) {
        const float w = K.w[tap];
        acc[o].x = fmaf(w, v.x, acc[o].x);
        acc[o].y = fmaf(w, v.y, acc[o].y);
        acc[o].z = fmaf(w, v.z, acc[o].z);
        acc[o].w = fmaf(w, v.w, acc[o].w);
      }
    }
  }
  // Outputs whose lower neighbour is in [l0, l0 + TL); chunk 0 also owns the "outside" outputs (lo < 0 -> 0).  The tables
  // are down-sampling tables (m <= n: positions at least one input row apart), so an input row is the lower neighbour of AT
  // MOST ONE output: a static walk over the chunk's rows with a uniform "does an output sit here" test keeps every register
  // index a compile-time constant (a switch over the row index is turned into a dynamic index by the optimiser and the
  // accumulators land in scratch memory).
  const ChunkTaps ct = rs_chunk_resolve(tab, m, l0, l0 + TL, jg, lt0);
  const LaneTaps lt = ct.t;
  const int jb = ct.jb - ct.shift;  // output of lane 0
  float4* d = dst + c;
  int q = ct.shift;
  while (q < 63 && rs_rl(lt.lo, q) < 0) {  // outside outputs (only ever at the start of the axis)
    if (live) d[(size_t)(jb + q) * inner4] = make_float4(0.f, 0.f, 0.f, 0.f);
    ++q;
  }
#pragma unroll
  for (int o = 0; o < TL; ++o) {
    if (rs_rl(lt.lo, q) == l0 + o) {
      const float wl = rs_rl(lt.wl, q), wh = rs_rl(lt.wh, q);
      const bool same = rs_rl(lt.hi, q) == l0 + o;
      const float4 out = rs_mix4(wl, acc[o], wh, same ? acc[o] : acc[o + 1]);
      if (live) d[(size_t)(jb + q) * inner4] = out;
      q = min(q + 1, 63);
    }
  }
}

// ---- axes 1 + 2 ----------------------------------------------------------------------------------------------------------
// rows per wave / waves per workgroup: 8 / 4 measured best at 256^3 (4 / 4: latency chains too short a work item, 44 -> 28 us
// at m = 250 without noise going to 8 / 4; 16 / 2: 197-254 VGPRs, 31 -> 41 us at m = 171)
#ifndef FSG_RSY_TL
#define FSG_RSY_TL 8
#endif
#ifndef FSG_RSY_NW
#define FSG_RSY_NW 4
#endif
#ifndef FSG_RSY_BATCH
#define FSG_RSY_BATCH 4
#endif
constexpr int RSY_BATCH = FSG_RSY_BATCH;
constexpr int RSY_TL = FSG_RSY_TL, RSY_NW = FSG_RSY_NW, RSY_IN = RSY_NW * RSY_TL, RSY_WROWS = RSY_TL + 1;  // input y rows per wave / workgroup; output rows a wave can emit (m <= n, + an "outside" one)

struct NoiseK {
  int mode;  // 0 none, 1 pointer, 2 Philox
  const float* noise;
  uint64_t seed, stream_id;
  float std;
};

// RSY_NW INDEPENDENT waves per workgroup, no barrier: wave w owns RSY_TL input rows of its x-plane (yin0 + 8w .. + 7) and everything
// downstream of them -- the output rows whose lower neighbour they are, their z blur (in place in the wave's own LDS rows)
// and their z resampling + noise.  Versions with workgroup-wide phases (y for the tile, barrier, z blur for the tile,
// barrier, lerp for the tile) and with 4 rows per wave ran 2-3x slower: every phase is short and ends in a wait, so a wave
// must carry enough rows through its own chain of waits.  The z taps of a lane's outputs stay in registers (the same for
// every row): NQ quads of consecutive outputs per lane, m2 <= 256 NQ.
// SAME: both axes use one tap set (the isotropic case the generator draws): Kz is not read, which halves the kernel's scalar
// register footprint (two tap sets spill 40-77 SGPRs at R >= 4).
template <int R, bool SAME, int NQ>
__global__ __launch_bounds__(64 * RSY_NW) void blur_rs_yz_kernel(const float4* __restrict__ src, float* __restrict__ dst, int ny, int nz,
                                                         int m1, int m2, const fsg_tap* __restrict__ taby,
                                                         const fsg_tap* __restrict__ tabz, TapsK Ky, TapsK Kz, NoiseK NZ) {
  constexpr int RP = (R + 3) & ~3, TL = RSY_TL, NB = TL + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int inner4 = nz >> 2, pitch = nz + 2 * RP;
  const int tx = threadIdx.x, tyc = __builtin_amdgcn_readfirstlane((int)threadIdx.y);  // wave-uniform: row offsets become scalar
  float* rows = lds + (size_t)tyc * RSY_WROWS * pitch;            // this wave's rows, zero z halos
  const int tiles_y = (ny + RSY_IN - 1) / RSY_IN;
  const int tile = xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int bx = tile / tiles_y;
  const int yin0 = (tile - bx * tiles_y) * RSY_IN;
  const size_t plane4 = (size_t)bx * ny * inner4;
  const int l0 = yin0 + tyc * TL;
  if (l0 >= ny) return;  // whole wave; there is no barrier in this kernel
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + plane4), 0, (unsigned)ny * (unsigned)nz * 4u, 0x00020000);
  // requested now, looked at later: the y taps of the chunk's outputs, the z taps of this lane's outputs
  const int jg = rs_chunk_guess(m1, l0, (float)m1 / (float)ny);
  const LaneTaps lt0 = rs_lane_taps(taby, m1, jg);
  int zlo[NQ][4], zhi[NQ][4];
  float zwl[NQ][4], zwh[NQ][4];
#pragma unroll
  for (int i = 0; i < NQ; ++i)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = (tx + 64 * i) * 4 + u;
      const int4 v = *reinterpret_cast<const int4*>(tabz + min(k, m2 - 1));
      const bool ok = k < m2 && v.x >= 0;
      zlo[i][u] = ok ? v.x : 0; zhi[i][u] = ok ? v.y : 0;
      zwl[i][u] = ok ? __builtin_bit_cast(float, v.z) : 0.f;  // an "outside" output: 0 * row[0] + 0 * row[0] = 0
      zwh[i][u] = ok ? __builtin_bit_cast(float, v.w) : 0.f;
    }
  for (int e = tx; e < RSY_WROWS * (2 * RP / 4); e += 64) {  // zero z halos of the wave's rows
    const int r = e / (2 * RP / 4), h = e - r * (2 * RP / 4);
    float* row = rows + (size_t)r * pitch;
    reinterpret_cast<float4*>(h < RP / 4 ? row : row + RP + nz)[h < RP / 4 ? h : h - RP / 4] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- y: blur in registers, resampled rows into the wave's LDS rows ----
  // (every lane runs every round of this loop -- column clamped, stores guarded: the table look-up below is a wave-level
  // routine of ballots and lane reads, and a lane that sat out would also miss jb / nrows)
  int jb = 0, nrows = 0;
  for (int cb = 0; cb < inner4; cb += 64) {
    const bool live = cb + tx < inner4;
    const int c = min(cb + tx, inner4 - 1);
    const unsigned colb = (unsigned)c * 16u, rowb = (unsigned)inner4 * 16u;
    float4 acc[NB];
#pragma unroll
    for (int o = 0; o < NB; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    // The NT row loads go out in batches of RSY_BATCH, two batches ahead of the arithmetic, with a scheduling barrier between
    // the batches: left to itself the scheduler hoists all NT loads (and their NT uniform 64-bit row offsets) to the top --
    // 4 NT vector + 2 NT scalar registers live at once, which is what set this kernel's occupancy (141-244 VGPRs, 16-130
    // spilled SGPRs at R >= 3).  Same loads, same arithmetic in the same order.
    constexpr int NT = NB + 2 * R;
    float4 rowv[NT];
    int l0s = l0;  // (opaque copy: the row offsets are computed next to their loads instead of NT scalar pairs ahead of the loop)
    asm volatile("" : "+s"(l0s));
#pragma unroll
    for (int t = 0; t < NT && t < 2 * RSY_BATCH; ++t) rowv[t] = rs_row_load(rsrc, (unsigned)(l0s + t - R) * rowb + colb);
#pragma unroll
    for (int b0 = 0; b0 < NT; b0 += RSY_BATCH) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = b0 + 2 * RSY_BATCH; t < NT && t < b0 + 3 * RSY_BATCH; ++t)
        rowv[t] = rs_row_load(rsrc, (unsigned)(l0s + t - R) * rowb + colb);  // unconditional; zeros outside the axis
#pragma unroll
      for (int t = b0; t < NT && t < b0 + RSY_BATCH; ++t) {
        const float4 v = rowv[t];
#pragma unroll
        for (int o = 0; o < NB; ++o) {
          const int tap = t - o;
          if (tap >= 0 && tap <= 2 * R) {
            const float w = Ky.w[tap];
            acc[o].x = fmaf(w, v.x, acc[o].x);
            acc[o].y = fmaf(w, v.y, acc[o].y);
            acc[o].z = fmaf(w, v.z, acc[o].z);
            acc[o].w = fmaf(w, v.w, acc[o].w);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const ChunkTaps ct = rs_chunk_resolve(taby, m1, l0, l0 + TL, jg, lt0);
    const LaneTaps lt = ct.t;
    const int jl0 = ct.jb - ct.shift;                 // output of lane 0 of the taps
    jb = ct.jb;
    nrows = min(ct.je - ct.jb, RSY_WROWS);
    int q = ct.shift;  // static walk over the chunk's rows, as in blur_rs_x_kernel
    while (q < 63 && rs_rl(lt.lo, q) < 0) {
      if (live && jl0 + q - jb < RSY_WROWS) reinterpret_cast<float4*>(rows + (size_t)(jl0 + q - jb) * pitch + RP)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      ++q;
    }
#pragma unroll
    for (int o = 0; o < TL; ++o) {
      if (rs_rl(lt.lo, q) == l0 + o) {
        const float wl = rs_rl(lt.wl, q), wh = rs_rl(lt.wh, q);
        const bool same = rs_rl(lt.hi, q) == l0 + o;
        const float4 out = rs_mix4(wl, acc[o], wh, same ? acc[o] : acc[o + 1]);
        if (live && jl0 + q - jb < RSY_WROWS) reinterpret_cast<float4*>(rows + (size_t)(jl0 + q - jb) * pitch + RP)[c] = out;
        q = min(q + 1, 63);
      }
    }
  }
  rs_wave_sync();
  // ---- per output row: z blur in place (every lane reads its windows, then the row is overwritten; taps ascending, fmaf),
  //      then z resampling + noise + clamp of the lane's quads (one Philox block per aligned quad of the FLAT output index:
  //      a row that does not start on a multiple of 4 takes its normals from two neighbouring blocks) ----
  // Two rows per round: their blur windows are read before ONE wave sync and written after it, and their two Philox +
  // Box-Muller chains (each a long run of dependent instructions) interleave in the same basic block.
  for (int r0 = 0; r0 < nrows; r0 += 2) {
    float o[2][2][4];  // [row of the pair][float4 of the lane: nz <= 512][component]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float* row = rows + (size_t)min(r0 + p, nrows - 1) * pitch;
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        const int qq = min(tx + 64 * qi, inner4 - 1);
        float win[4 + 2 * RP];
#pragma unroll
        for (int u = 0; u < (4 + 2 * RP) / 4; ++u) {
          const float4 t = reinterpret_cast<const float4*>(row)[qq + u];
          win[4 * u] = t.x; win[4 * u + 1] = t.y; win[4 * u + 2] = t.z; win[4 * u + 3] = t.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[p][qi][e] = 0.f;
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) {
          const float w = SAME ? Ky.w[t] : Kz.w[t];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[p][qi][e] = fmaf(w, win[RP - R + e + t], o[p][qi][e]);
        }
        if (qi == 0 && inner4 <= 64) break;  // uniform: one float4 per lane and row
      }
    }
    rs_wave_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (r0 + p < nrows) {
        float* row = rows + (size_t)(r0 + p) * pitch;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
          if (tx + 64 * qi < inner4) reinterpret_cast<float4*>(row + RP)[tx + 64 * qi] = make_float4(o[p][qi][0], o[p][qi][1], o[p][qi][2], o[p][qi][3]);
      }
    }
    rs_wave_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (r0 + p >= nrows) break;  // uniform
      const float* rd = rows + (size_t)(r0 + p) * pitch + RP;
      const size_t rbase = ((size_t)bx * m1 + jb + r0 + p) * m2;  // flat index of the row's first output
      const int sh = (int)(rbase & 3);                           // wave-uniform misalignment against the Philox quads
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int k0 = (tx + 64 * i) * 4;
        if (i > 0 && 256 * i >= m2) break;  // uniform
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fsg_mix(zwl[i][u], rd[zlo[i][u]], zwh[i][u], rd[zhi[i][u]]);
        const size_t e0 = rbase + (size_t)k0;
        if (NZ.mode == 2) {
          // outputs e0 .. e0 + 3 = elements sh .. 3 of block e0 >> 2 and 0 .. sh - 1 of the next one
          // (the next block is the NEXT LANE's own block -- its outputs start four elements further -- so it comes over by a
          // lane shift instead of a second Philox + Box-Muller evaluation; only the last lane of the row may need its own)
          const float4 za = fsg_randn4<true>(NZ.seed, NZ.stream_id, (uint64_t)(e0 >> 2));
          float z[4] = {za.x, za.y, za.z, za.w};
          if (sh) {
            float4 zb = make_float4(__shfl_down(za.x, 1, FSG_WAVE), __shfl_down(za.y, 1, FSG_WAVE), __shfl_down(za.z, 1, FSG_WAVE),
                                    __shfl_down(za.w, 1, FSG_WAVE));
            // lane 63's successor is not in this wave-instruction: needed only if one of its VALID outputs reaches into it
            const int nvalid63 = min(m2 - (63 + 64 * i) * 4, 4);  // uniform
            if (sh + nvalid63 - 1 >= 4) {
              if (tx == 63) zb = fsg_randn4<true>(NZ.seed, NZ.stream_id, (uint64_t)(e0 >> 2) + 1);
            }
            const float zz[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
            if (sh == 1) { z[0] = zz[1]; z[1] = zz[2]; z[2] = zz[3]; z[3] = zz[4]; }
            else if (sh == 2) { z[0] = zz[2]; z[1] = zz[3]; z[2] = zz[4]; z[3] = zz[5]; }
            else { z[0] = zz[3]; z[1] = zz[4]; z[2] = zz[5]; z[3] = zz[6]; }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] += NZ.std * z[u];
        } else if (NZ.mode == 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < m2) v[u] += NZ.std * NZ.noise[e0 + u];
        }
        if (NZ.mode != 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = v[u] < 0.f ? 0.f : v[u];
        }
        if (k0 + 3 < m2) {  // 16-byte store, 4-byte aligned (rows start anywhere)
          typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
          *reinterpret_cast<f4u*>(dst + e0) = f4u{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < m2) dst[e0 + u] = v[u];
        }
      }
    }
  }
}

size_t rs_yz_lds(int nz, int m2, int R) {
  const int RP = (R + 3) & ~3;
  (void)m2;
  return ((size_t)RSY_NW * RSY_WROWS * (nz + 2 * RP)) * sizeof(float);
}

template <int R>
int launch_rs_yz(const float* src, float* dst, int m0, int ny, int nz, int m1, int m2, const fsg_tap* ty, const fsg_tap* tz,
                 const TapsK& Ky, const TapsK& Kz, const NoiseK& NZ, hipStream_t st) {
  const size_t lds = rs_yz_lds(nz, m2, R);
  if (lds > 64000) return FSG_E_ALIGN;
  const unsigned tiles_y = (unsigned)((ny + RSY_IN - 1) / RSY_IN);
  bool same = true;
  for (int t = 0; t < RS_KCAP + 3; ++t) same = same && Ky.w[t] == Kz.w[t];
#define RS_YZ(S, Q)                                                                                            \
  hipLaunchKernelGGL((blur_rs_yz_kernel<R, S, Q>), dim3(tiles_y * (unsigned)m0), dim3(64, RSY_NW), lds, st,        \
                     reinterpret_cast<const float4*>(src), dst, ny, nz, m1, m2, ty, tz, Ky, S ? Ky : Kz, NZ)
  if (m2 <= 256) { if (same) RS_YZ(true, 1); else RS_YZ(false, 1); }
  else           { if (same) RS_YZ(true, 2); else RS_YZ(false, 2); }
#undef RS_YZ
  FSG_RETURN_LAUNCH();
}

template <int R>
int launch_rs_x(const float* src, float* dst, int n, int m, int inner4, const fsg_tap* tab, const TapsK& K, hipStream_t st) {
  const int chunks = (n + RSX_TL - 1) / RSX_TL;
  dim3 grid((unsigned)((inner4 + 63) / 64), (unsigned)((chunks + 3) / 4)), block(64, 4);
  hipLaunchKernelGGL(blur_rs_x_kernel<R>, grid, block, 0, st, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n, m, inner4, tab, K);
  FSG_RETURN_LAUNCH();
}

// taps centred in a set of radius R (>= their own radius): zero weights outside -- fmaf(0, v, acc) == acc, so the padded
// blur is bit-identical to the unpadded one
int rs_fill_taps(const float* taps_host, int ntaps, int R, TapsK& K) {
  if (!taps_host || ntaps < 3 || (ntaps & 1) == 0 || ntaps > RS_KCAP || (ntaps >> 1) > R || R > RS_MAXR) return FSG_E_ALIGN;
  const int pad = R - (ntaps >> 1);
  for (int t = 0; t < RS_KCAP + 3; ++t) K.w[t] = (t >= pad && t < pad + ntaps) ? taps_host[t - pad] : 0.f;
  return 0;
}

}  // namespace

extern "C" {

int fsg_blur_resample_supported(int n0, int n1, int n2, int m0, int m1, int m2, int ntaps_x, int ntaps_y, int ntaps_z) {
  if (g_tuning_flags & FSG_TUNE_NO_BLUR_RS) return 0;  // A/B switch: every caller then takes the unfused sequence
  if (n0 <= 0 || n1 <= 0 || n2 <= 0 || m0 <= 0 || m1 <= 0 || m2 <= 0) return 0;
  if (m0 > n0 || m1 > n1 || m2 > n2) return 0;  // down-sampling tables (lo non-decreasing, at most one output per input row + 1)
  for (int nt : {ntaps_x, ntaps_y, ntaps_z})
    if (nt < 3 || (nt & 1) == 0 || nt > RS_KCAP) return 0;
  if ((n2 & 3) || n2 > 512 || ((long long)n1 * n2 & 3)) return 0;
  if ((size_t)n0 * n1 * n2 > ((size_t)1 << 29)) return 0;  // rows are addressed by 32-bit byte offsets in a buffer descriptor < 2^31 bytes
  const int Ryz = (ntaps_y > ntaps_z ? ntaps_y : ntaps_z) >> 1;
  if (rs_yz_lds(n2, m2, Ryz) > 64000) return 0;
  return 1;
}

int fsg_blur_resample_x_f32(const float* src, int n0, int n1, int n2, const fsg_tap* tx, int m0, const float* taps_host,
                            int ntaps, float* dst, void* stream) {
  if (!src || !dst || src == dst || !tx) return FSG_E_BADARG;
  if (n0 <= 0 || n1 <= 0 || n2 <= 0 || m0 <= 0) return FSG_E_BADARG;
  if (m0 > n0) return FSG_E_ALIGN;
  const int R = ntaps >> 1;
  TapsK K;
  int rc = rs_fill_taps(taps_host, ntaps, R, K);
  if (rc) return rc;
  const long long inner = (long long)n1 * n2;
  if ((inner & 3) || ((((uintptr_t)src) | ((uintptr_t)dst)) & 15)) return FSG_E_ALIGN;
  if ((size_t)n0 * inner > ((size_t)1 << 29)) return FSG_E_TOOBIG;  // 32-bit byte offsets in a descriptor < 2^31 bytes
  const int inner4 = (int)(inner >> 2);
  hipStream_t st = fsg_stream(stream);
  switch (R) {
    case 1: return launch_rs_x<1>(src, dst, n0, m0, inner4, tx, K, st);
    case 2: return launch_rs_x<2>(src, dst, n0, m0, inner4, tx, K, st);
    case 3: return launch_rs_x<3>(src, dst, n0, m0, inner4, tx, K, st);
    case 4: return launch_rs_x<4>(src, dst, n0, m0, inner4, tx, K, st);
    case 5: return launch_rs_x<5>(src, dst, n0, m0, inner4, tx, K, st);
    case 6: return launch_rs_x<6>(src, dst, n0, m0, inner4, tx, K, st);
    case 7: return launch_rs_x<7>(src, dst, n0, m0, inner4, tx, K, st);
    case 8: return launch_rs_x<8>(src, dst, n0, m0, inner4, tx, K, st);
    default: return FSG_E_ALIGN;
  }
}

int fsg_blur_resample_yz_noise_f32(const float* src, int m0, int n1, int n2, const fsg_tap* ty, const fsg_tap* tz, int m1,
                                   int m2, const float* taps_y_host, int ntaps_y, const float* taps_z_host, int ntaps_z,
                                   int noise_mode, const float* noise, uint64_t seed, uint64_t stream_id, float noise_std,
                                   float* dst, void* stream) {
  if (!src || !dst || src == dst || !ty || !tz) return FSG_E_BADARG;
  if (m0 <= 0 || n1 <= 0 || n2 <= 0 || m1 <= 0 || m2 <= 0 || noise_mode < 0 || noise_mode > 2) return FSG_E_BADARG;
  if (noise_mode == 1 && !noise) return FSG_E_BADARG;
  if (m1 > n1 || m2 > n2) return FSG_E_ALIGN;
  const int R = (ntaps_y > ntaps_z ? ntaps_y : ntaps_z) >> 1;  // one radius for both axes: the narrower tap set is zero-padded
  TapsK Ky, Kz;
  int rc = rs_fill_taps(taps_y_host, ntaps_y, R, Ky);
  if (rc) return rc;
  rc = rs_fill_taps(taps_z_host, ntaps_z, R, Kz);
  if (rc) return rc;
  if ((n2 & 3) || n2 > 512 || (((uintptr_t)src) & 15)) return FSG_E_ALIGN;
  if ((size_t)m0 * n1 * n2 > (size_t)0x7FFFFFFF || (size_t)m0 * m1 * m2 > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  if ((size_t)n1 * n2 > ((size_t)1 << 29)) return FSG_E_TOOBIG;  // a plane is one buffer descriptor (< 2^31 bytes)
  NoiseK NZ{noise_mode, noise, seed, stream_id, noise_std};
  hipStream_t st = fsg_stream(stream);
  switch (R) {
    case 1: return launch_rs_yz<1>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 2: return launch_rs_yz<2>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 3: return launch_rs_yz<3>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 4: return launch_rs_yz<4>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 5: return launch_rs_yz<5>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 6: return launch_rs_yz<6>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 7: return launch_rs_yz<7>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    case 8: return launch_rs_yz<8>(src, dst, m0, n1, n2, m1, m2, ty, tz, Ky, Kz, NZ, st);
    default: return FSG_E_ALIGN;
  }
}

}  // extern "C"
