// fsg_warp_lean.hip -- K3+K4(+K5): the fused warp with a lean per-voxel instruction stream.
//
// Same contract and the same fp32 operation order as warp_patch_kernel / warp_rows_kernel of fsg_deform.hip
// (reference: affine_nonrigid.py:327-366 positions, utils/generation.py:204-288 samplers, synthseg.py:274 and
// :178-182 epilogue) -- outputs are bit-identical (tests/test_hip_parity.py::test_warp_work_shapes_are_bit_identical).
//
// What is different is the cost per voxel.  The patch kernel issues ~230 vector instructions per 64 voxels and its
// time follows that count (profiles/r02_warp_*.txt): it is bound by instruction issue, not by bytes.  This body
// spends ~40 % fewer:
//   * the per-row coarse displacement values sit in LDS as one float4 (dx, dy, dz, -) per coarse z index: two
//     ds_read_b128 per voxel instead of six ds_read_b32 with six address computations;
//   * sampling positions are clamped with v_med3_f32 (one instruction per axis instead of two compare/select
//     pairs; a coordinate of -0.0 may come out as +0.0, which no consumer below distinguishes);
//   * raw buffer loads / stores: a 32-bit byte offset per access, no 64-bit address arithmetic, and an
//     out-of-range offset returns 0 instead of faulting -- so no index is clamped: indices are in range by
//     construction, and where the reference clamps the upper neighbour onto the base (coordinate exactly on the
//     last plane / row) that neighbour's weight is exactly 0, so the finite value (or the 0 of an out-of-range
//     read) found there contributes +-0;
//   * the axis-0 flip is folded into a signed plane stride and a base offset (no per-voxel select);
//   * index products on v_mul_i32_i24 / v_mad_i32_i24;
//   * work shape: the 16 waves of a workgroup sweep 8 x 4 adjacent rows, 32 voxels of each per lockstep step
//     (fsg_warp_set_variant 3 of the patch kernel family, the fastest of the shapes measured).
// Domain (else FSG_E_ALIGN and the caller falls back): per-row coarse values precomputed (fsg_deform_rows_f32),
// coarse grids of at most 32 entries along z, shape[2] <= 512, shape[1]*shape[2] < 2^22, fewer than 2^30 voxels.
#include "fsg_common.h"

int g_lean_ablate = 0;  // 1 / 2: the diagnostic kernels (ABL), never set by the product
int g_lean_pace = -1;  // lockstep barrier: 1 = every step, 2 = every second step, 0 = none; -1 = chosen per launch (fsg_warp_set_variant 5..7 force one)

namespace {

constexpr int LEAN_F2CAP = 32;   // coarse displacement entries along z
constexpr int LEAN_B2CAP = 32;   // coarse bias entries along z
constexpr int LEAN_TZCAP = 512;  // z extent whose taps are staged in LDS

typedef float f2v __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ T lean_load_label(__amdgpu_buffer_rsrc_t r, unsigned elem);
template <>
__device__ __forceinline__ float lean_load_label<float>(__amdgpu_buffer_rsrc_t r, unsigned elem) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, elem * 4u, 0, 0));
}
template <>
__device__ __forceinline__ uint8_t lean_load_label<uint8_t>(__amdgpu_buffer_rsrc_t r, unsigned elem) {
  return __builtin_amdgcn_raw_buffer_load_b8(r, elem, 0, 0);
}
__device__ __forceinline__ void lean_store_label(__amdgpu_buffer_rsrc_t r, unsigned elem, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, elem * 4u, 0, 0);
}
__device__ __forceinline__ void lean_store_label(__amdgpu_buffer_rsrc_t r, unsigned elem, uint8_t v) {
  __builtin_amdgcn_raw_buffer_store_b8(v, r, elem, 0, 0);
}

// ABL (diagnostic instantiations only, tools/kernel_bench.py --variant 8 / 9): 1 = every vector-memory access predicated off at run time (arithmetic and
// LDS traffic only), 2 = trivial sampling positions (memory traffic and blends only).  0 in every product launch.
template <typename ST, typename DT, bool HAS_LIN, bool HAS_NN, bool FAST, int KZ, int WI, int ABL = 0>
__global__ __launch_bounds__(1024, 8) void warp_lean_kernel(FsgDeformK D, const int32_t* __restrict__ mm6,
                                                         const float* __restrict__ src_lin, float* __restrict__ out_lin,
                                                         const ST* __restrict__ src_nn, DT* __restrict__ out_nn, EpiK E,
                                                         int pace) {
  constexpr int RJ = 64 / KZ;  // rows (along j) per wave
  constexpr int WJ = 16 / WI;  // waves along j
  constexpr int PI = WI, PJ = WJ * RJ, NROW = PI * PJ;
  __shared__ float4 s_f[NROW][LEAN_F2CAP];  // x/y-interpolated coarse displacement of every row: (dx, dy, dz, -) per coarse z
  __shared__ float s_b[NROW][LEAN_B2CAP];   // x/y-interpolated coarse bias of every row
  __shared__ int4 s_tz[LEAN_TZCAP], s_bz[LEAN_TZCAP];  // z taps of the displacement / bias grids
  const bool has_field = D.field != nullptr, has_bias = E.bias != nullptr;
  for (int t = threadIdx.x; t < D.n2; t += 1024) {
    if (has_field) s_tz[t] = *reinterpret_cast<const int4*>(D.tz + t);
    if (has_bias) s_bz[t] = *reinterpret_cast<const int4*>(E.bz + t);
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tiles_j = (D.n1 + PJ - 1) / PJ;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int wi = wave / WJ, wj = wave - wi * WJ;
  const int i_raw = (tile / tiles_j) * PI + wi;
  const int j_wave = (tile % tiles_j) * PJ + wj * RJ;
  const int rj = lane / KZ, kz = lane - rj * KZ;
  const int j_raw = j_wave + rj;
  const bool live_row = i_raw < D.n0 && j_raw < D.n1;
  const int i = min(i_raw, D.n0 - 1), j = min(j_raw, D.n1 - 1);
  const Margins m = load_margins(mm6);
  const int nf = has_field ? 3 * D.f2 : 0;
  // stage the wave's rows from the precomputed per-row values (channel-major there, interleaved here)
#pragma unroll
  for (int r = 0; r < RJ; ++r) {
    const int jr = min(j_wave + r, D.n1 - 1);
    const float* __restrict__ g = D.rows + ((size_t)i * D.n1 + jr) * D.row_stride;
    float* dst = reinterpret_cast<float*>(&s_f[wave * RJ + r][0]);
    if (has_field && lane < D.f2) {
      dst[lane * 4 + 0] = g[lane];
      dst[lane * 4 + 1] = g[D.f2 + lane];
      dst[lane * 4 + 2] = g[2 * D.f2 + lane];
    }
    if (has_bias && lane < E.b2) s_b[wave * RJ + r][lane] = g[nf + lane];
  }
  __syncthreads();

  const int rowid = wave * RJ + rj;
  const float4* __restrict__ sf = s_f[rowid];
  const float* __restrict__ sb = s_b[rowid];
  const unsigned nvox = (unsigned)D.n0 * (unsigned)D.n1 * (unsigned)D.n2;
  const __amdgpu_buffer_rsrc_t r_lin = __builtin_amdgcn_make_buffer_rsrc((void*)src_lin, 0, HAS_LIN ? nvox * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_olin = __builtin_amdgcn_make_buffer_rsrc((void*)out_lin, 0, HAS_LIN ? nvox * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_nn =
      __builtin_amdgcn_make_buffer_rsrc((void*)src_nn, 0, HAS_NN ? nvox * (unsigned)sizeof(ST) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_onn =
      __builtin_amdgcn_make_buffer_rsrc((void*)out_nn, 0, HAS_NN ? nvox * (unsigned)sizeof(DT) : 0u, 0x00020000);
  // element offset of source voxel (x, y, z) = base + x * sxs + y * sy + z, the flip folded into base / sxs
  const int sy = D.n2;
  const int sx = D.n1 * D.n2;
  const int sxs = D.flip ? -sx : sx;
  const int base = D.flip ? (D.n0 - 1) * sx : 0;
  const int dxb = sxs * 4, oyb = sy * 4;  // byte strides towards the upper x / y neighbours
  const float hx = (float)(D.n0 - 1), hy = (float)(D.n1 - 1), hz = (float)(D.n2 - 1);
  const float pxb = (float)i - D.cen[0], pyb = (float)j - D.cen[1];
  const unsigned orow = (unsigned)(((size_t)i * D.n1 + j) * D.n2);  // first output element of this lane's row

  // One step of one lane, split in two so that the gathers of step s+1 are in flight while step s is blended:
  //   issue(kb, P)  -- sampling position of voxel (i, j, kb + kz), the label gather and the four 8-byte gathers
  //   finish(P)     -- blend, epilogue, stores
  // The arithmetic is written on register PAIRS (f2v) wherever two results share their operation sequence -- (y, z) of
  // the affine map, the two z-neighbours of every blend -- so that it lands on v_pk_mul_f32 / v_pk_add_f32 with the pairs
  // where the loads put them (no v_mov / v_pk_mov shuffles); each half is the reference's own operation order.
  struct Pend {
    f2v p00, p10, p01, p11;
    float bx;
    f2v byz;
    bool ok;
    ST lab;
  };
  const f2v A_x = {D.A[3], D.A[6]}, A_y = {D.A[4], D.A[7]}, A_z = {D.A[5], D.A[8]}, c2_yz = {D.c2[1], D.c2[2]};
  const f2v m_yz = {m.my, m.mz}, pb_xy = {pxb, pyb};
  const float hz1 = hz - 1.f;
  auto issue = [&](int kb, Pend& P) {
    const int k = min(kb + kz, D.n2 - 1);
    f2v pxy = pb_xy;
    float pz = (float)k - D.cen[2];
    if (has_field && ABL != 2) {
      const int4 c = s_tz[k];
      const float wl = __builtin_bit_cast(float, c.z), wh = __builtin_bit_cast(float, c.w);
      const float4 a = sf[c.x], b = sf[c.y];
      pxy = pxy + (f2v{a.x, a.y} * wl + f2v{b.x, b.y} * wh);
      float tz = wl * a.z;
      asm("" : "+v"(tz));  // keeps this product scalar: a packed form would first move (a.z, b.z) into a register pair
      pz = pz + (tz + wh * b.z);
    }
    float x = D.A[0] * pxy.x + D.A[1] * pxy.y + D.A[2] * pz + D.c2[0];
    f2v yz = A_x * pxy.x + A_y * pxy.y + A_z * pz + c2_yz;
    if (ABL == 2) { x = (float)i + 0.25f; yz.x = (float)j + 0.25f; yz.y = (float)k * 0.98f + 0.3f; }
    x = __builtin_amdgcn_fmed3f(x, 0.f, hx) - m.mx;
    yz = f2v{__builtin_amdgcn_fmed3f(yz.x, 0.f, hy), __builtin_amdgcn_fmed3f(yz.y, 0.f, hz)} - m_yz;
    const bool mem = ABL != 1 || x == 1.2345e30f;  // ABL 1: never true, decided per lane at run time
    P.lab = 0;
    if (HAS_NN) {
      const int xi = (int)rintf(x), yi = (int)rintf(yz.x), zi = (int)rintf(yz.y);  // round half to even; in range by construction
      const unsigned e = (unsigned)(__mul24(xi, sxs) + (__mul24(yi, sy) + (zi + base)));
      if (ABL != 1) P.lab = lean_load_label<ST>(r_nn, e);
      else if (mem) P.lab = lean_load_label<ST>(r_nn, e);
      else P.lab = (ST)(e & 7u);
    }
    if (HAS_LIN) {
      P.ok = (x > 0.f) && (yz.x > 0.f) && (yz.y > 0.f);
      // z exactly on the last column: the pair (z0, z0 + 1) would leave the row, so the pair one to the left is read with
      // weights (0, 1) -- the reference's (1, 0) on a neighbour clamped onto the column itself gives the same value
      const float fx = floorf(x);
      const f2v fyz = {floorf(yz.x), __builtin_amdgcn_fmed3f(floorf(yz.y), -1.f, hz1)};
      P.bx = x - fx;
      P.byz = yz - fyz;
      // unconditional: a voxel that samples outside the volume is clamped onto a face, its (unused) reads are in range and
      // mostly hit the lines its neighbours fetch; a predicated gather would make the number of loads in flight unknown to
      // the compiler, which then waits for ALL of them (vmcnt(0)) -- including the next step's -- before every blend
      const unsigned o = (unsigned)(__mul24((int)fx, sxs) + (__mul24((int)fyz.x, sy) + ((int)fyz.y + base))) * 4u;
      if (ABL != 1 || mem) {
        P.p00 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(r_lin, o, 0, 0));
        P.p10 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(r_lin, o + (unsigned)dxb, 0, 0));
        P.p01 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(r_lin, o + (unsigned)oyb, 0, 0));
        P.p11 = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(r_lin, o + (unsigned)(dxb + oyb), 0, 0));
      } else {
        const float t = __builtin_bit_cast(float, (o & 0xFFFFu) | 0x3F800000u);
        P.p00 = f2v{t, t + 1.f}; P.p10 = f2v{t + 2.f, t}; P.p01 = f2v{t, t + 3.f}; P.p11 = f2v{t + 1.f, t};
      }
    }
  };
  auto finish = [&](int kb, Pend& P) {
    const int kk = kb + kz;
    const bool live = live_row && kk < D.n2;
    const unsigned oelem = orow + (unsigned)kk;
    if (HAS_LIN) {
      const float bx = P.bx, ax = 1.f - bx;
      const f2v byz = P.byz, ayz = 1.f - byz;
      const f2v cx0 = P.p00 * ax + P.p10 * bx;      // (c00, c01): the two z-neighbours at y0
      const f2v cx1 = P.p01 * ax + P.p11 * bx;      // (c10, c11): at y0 + 1
      const f2v cy = cx0 * ayz.x + cx1 * byz.x;     // (c0, c1)
      float c0z = cy.x * ayz.y;
      asm("" : "+v"(c0z));  // scalar on purpose, as above: (az, bz) straddles the (ay, az) / (by, bz) pairs
      float v = P.ok ? (c0z + cy.y * byz.y) : 0.f;
      if (E.gamma > 0.f) {
        // 300*(v/300)^g.  FAST: 300 * 2^(g*(log2 v - log2 300)) on v_log_f32 / v_exp_f32; else OCML powf and an IEEE division
        if (FAST) v = 300.0f * __builtin_amdgcn_exp2f(E.gamma * (__builtin_amdgcn_logf(v) - 8.2288186904958804f));
        else v = 300.0f * powf(v / 300.0f, E.gamma);
      }
      if (has_bias) {
        const int4 cb = s_bz[min(kk, D.n2 - 1)];
        const float bval = fsg_mix(__builtin_bit_cast(float, cb.z), sb[cb.x], __builtin_bit_cast(float, cb.w), sb[cb.y]);
        v = v * (FAST ? __builtin_amdgcn_exp2f(bval * 1.4426950408889634f) : expf(bval));
      }
      if (live && (ABL != 1 || v == 1.2345e30f))
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_olin, oelem * 4u, 0, 0);
    }
    if (HAS_NN && live && (ABL != 1 || P.bx == 1.2345e30f)) lean_store_label(r_onn, oelem, (DT)P.lab);
  };

  // the barrier only paces the 16 waves (same z slab -> the brick's source block is what L1 holds); nothing is
  // exchanged through memory inside the loop, so no fence: loads of the next step stay in flight across it
  Pend A, B;
  issue(0, A);
  for (int kb = 0; kb < D.n2; kb += 2 * KZ) {
    const bool more1 = kb + KZ < D.n2;
    if (more1) issue(kb + KZ, B);
    finish(kb, A);
    if (pace == 1) __builtin_amdgcn_s_barrier();
    if (more1) {
      if (kb + 2 * KZ < D.n2) issue(kb + 2 * KZ, A);
      finish(kb + KZ, B);
      if (pace) __builtin_amdgcn_s_barrier();
    }
  }
}

template <typename ST, typename DT>
int launch_lean(const FsgDeformK& D, const EpiK& E, const int32_t* mm6, const float* src_lin, float* out_lin,
                const ST* src_nn, DT* out_nn, bool fast, hipStream_t st) {
  constexpr int KZ = 32, WI = 8;
  constexpr int PI = WI, PJ = (16 / WI) * (64 / KZ);
  const dim3 grid((unsigned)(((D.n0 + PI - 1) / PI) * ((D.n1 + PJ - 1) / PJ))), block(1024);
  // Pacing.  The barrier keeps the 16 waves on one z slab so that L1 holds the source block they share.  With a small
  // slope of the source position along the output row (|dx/dk|, |dy/dk| -- the affine's third column) a slab's block is
  // small and stays resident across two steps, and a barrier every second step is 4-5 us faster at 256^3; beyond
  // ~16 degrees the every-step barrier wins (gpurun_out r4d, profiles/r02_f_warp_pace.txt).
  const float slope = fmaxf(fabsf(D.A[2]), fabsf(D.A[5]));
  const int pace = g_lean_pace >= 0 ? g_lean_pace : (slope < 0.27f ? 2 : 1);
#define FSG_LEAN(L, N, F) \
  hipLaunchKernelGGL((warp_lean_kernel<ST, DT, L, N, F, KZ, WI>), grid, block, 0, st, D, mm6, src_lin, out_lin, src_nn, out_nn, E, \
                     pace)
#ifdef FSG_DIAG  // ablation instantiations (results are wrong; tools/kernel_bench.py --variant 8 / 9 on a -DFSG_DIAG build)
  if (src_lin && src_nn && fast && g_lean_ablate && sizeof(ST) == 4 && sizeof(DT) == 4) {
    if (g_lean_ablate == 1)
      hipLaunchKernelGGL((warp_lean_kernel<ST, DT, true, true, true, KZ, WI, 1>), grid, block, 0, st, D, mm6, src_lin, out_lin,
                         src_nn, out_nn, E, pace);
    else
      hipLaunchKernelGGL((warp_lean_kernel<ST, DT, true, true, true, KZ, WI, 2>), grid, block, 0, st, D, mm6, src_lin, out_lin,
                         src_nn, out_nn, E, pace);
    FSG_RETURN_LAUNCH();
  }
#endif
  if (src_lin && src_nn) { if (fast) FSG_LEAN(true, true, true); else FSG_LEAN(true, true, false); }
  else if (src_lin)      { if (fast) FSG_LEAN(true, false, true); else FSG_LEAN(true, false, false); }
  else                   { FSG_LEAN(false, true, true); }
#undef FSG_LEAN
  FSG_RETURN_LAUNCH();
}

}  // namespace

int fsg_launch_warp_lean(const FsgDeformK& D, const EpiK& E, const int32_t* mm6, const float* src_lin, float* out_lin,
                         const void* src_nn, void* out_nn, int label_in_bytes, int label_out_bytes, bool fast,
                         void* stream) {
  const bool has_field = D.field != nullptr, has_bias = E.bias != nullptr;
  const int need = (has_field ? 3 * D.f2 : 0) + (has_bias ? E.b2 : 0);
  if (need > 0 && (!D.rows || D.row_stride < need)) return FSG_E_ALIGN;
  if ((has_field && D.f2 > LEAN_F2CAP) || (has_bias && E.b2 > LEAN_B2CAP)) return FSG_E_ALIGN;
  if (D.n2 < 2 || D.n2 > LEAN_TZCAP) return FSG_E_ALIGN;
  const long long plane = (long long)D.n1 * D.n2, nvox = plane * D.n0;
  if (plane >= (1ll << 22) || nvox >= (1ll << 30) || D.n0 >= (1 << 22)) return FSG_E_ALIGN;
  hipStream_t st = fsg_stream(stream);
  if (!src_nn) return launch_lean<float, float>(D, E, mm6, src_lin, out_lin, nullptr, nullptr, fast, st);
  if (label_in_bytes == 4 && label_out_bytes == 4)
    return launch_lean<float, float>(D, E, mm6, src_lin, out_lin, (const float*)src_nn, (float*)out_nn, fast, st);
  if (label_in_bytes == 1 && label_out_bytes == 1)
    return launch_lean<uint8_t, uint8_t>(D, E, mm6, src_lin, out_lin, (const uint8_t*)src_nn, (uint8_t*)out_nn, fast, st);
  if (label_in_bytes == 1 && label_out_bytes == 4)
    return launch_lean<uint8_t, float>(D, E, mm6, src_lin, out_lin, (const uint8_t*)src_nn, (float*)out_nn, fast, st);
  return FSG_E_BADARG;
}
