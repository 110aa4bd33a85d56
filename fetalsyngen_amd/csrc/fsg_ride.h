// Device jobs that more than one launch carries: the keyed draw job (its own kernel in fsg_keyed.hip; beside the floor(min)
// pass in fsg_deform.hip when the caller named the next sample: fsg_sample_plan::ride_draw) and the code-volume GMM job of the
// head kernel.  (The GMM job of the NEXT sample beside the zoom-back launches was built and measured in r03 -- K9b 27 -> 52-61 us,
// K9a 18 -> 45-55 us for a job that costs ~21 us in the head kernel: the two slow each other down -- and removed again.)
#pragma once
#include "fsg_common.h"

#define FSG_MM_NSLOTS 64      // slots of the sharded K9 keys (generator/model.py: MM_NSLOTS; fsg_zoom3d_minmax_sharded_f32: 2..64)

namespace fsg_ride {

// order-preserving keys of +inf / -inf (fsg_f2key): what fsg_minmax_init writes
constexpr int32_t KEY_POS_INF = 0x7F800000, KEY_NEG_INF = (int32_t)0x807FFFFF;

struct DrawK {
  uint64_t key;
  int32_t* mm8;
  int32_t* slots;
  float* mus;
  float* sigmas;
  float* bias;
  float* field;
  int nlabels, nseed, tie, nbias, nfield;
  float bias_std, field_std;
  uint8_t seed_labels[256];
  uint8_t gen_classes[256];
};

__device__ __forceinline__ float keyed_uniform(uint64_t key, uint64_t stream, uint32_t e) {
  const uint32_t blk = e >> 2;
  const uint4 r = fsg_philox4x32_10(blk, 0u, (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)key, (uint32_t)(key >> 32));
  const uint32_t w = (e & 3) == 0 ? r.x : ((e & 3) == 1 ? r.y : ((e & 3) == 2 ? r.z : r.w));
  return (float)(w >> 8) * 5.9604644775390625e-08f;  // [0, 1), 24 bits: torch.rand's float32 grid
}

// Workgroup 0: min/max keys, GMM tables (rand_gmm.py:120-145).  Workgroups >= 1: one Philox block (4 normals) per thread of
// the bias grid (stream 4, synthseg.py:172-176) and the coarse displacement grid (stream 3, affine_nonrigid.py:318).
__device__ __forceinline__ void keyed_draw_body(const DrawK& P, const int blk) {
  const int tid = threadIdx.x;
  if (blk == 0) {
    if (tid < 8) P.mm8[tid] = tid < 4 ? KEY_POS_INF : KEY_NEG_INF;
    for (int q = tid; q < FSG_MM_NSLOTS * FSG_MM_SLOT_STRIDE; q += 256) {
      const int f = q % FSG_MM_SLOT_STRIDE;
      P.slots[q] = f == 0 ? KEY_POS_INF : (f == 1 ? KEY_NEG_INF : 0);
    }
    __shared__ float s_mu[256];
    float sg = 0.f;
    if (tid < P.nlabels) {
      s_mu[tid] = 25.f + 200.f * keyed_uniform(P.key, 5, (uint32_t)tid);
      sg = 5.f + 20.f * keyed_uniform(P.key, 5, (uint32_t)(P.nlabels + tid));
    }
    __syncthreads();
    float tied = 0.f;
    if (P.tie && tid < P.nseed) {  // the right-hand side is read in full before anything is written (numpy semantics)
      tied = s_mu[P.gen_classes[tid]] + 25.f * fsg_randn1(P.key, 6, (uint64_t)tid);
      tied = fminf(fmaxf(tied, 0.f), 225.f);
    }
    __syncthreads();
    if (P.tie && tid < P.nseed) s_mu[P.seed_labels[tid]] = tied;
    __syncthreads();
    if (tid < P.nlabels) {
      P.mus[tid] = s_mu[tid];
      P.sigmas[tid] = sg;
    }
    return;
  }
  const int t = (blk - 1) * 256 + tid;
  const int nb4 = (P.nbias + 3) >> 2, nf4 = (P.nfield + 3) >> 2;
  if (t < nb4) {
    const float4 z = fsg_randn4(P.key, 4, (uint64_t)t);
    const float v[4] = {z.x, z.y, z.z, z.w};
    for (int q = 0; q < 4; ++q)
      if (4 * t + q < P.nbias) P.bias[4 * t + q] = P.bias_std * v[q];
  } else if (t - nb4 < nf4) {
    const int u = t - nb4;
    const float4 z = fsg_randn4(P.key, 3, (uint64_t)u);
    const float v[4] = {z.x, z.y, z.z, z.w};
    for (int q = 0; q < 4; ++q)
      if (4 * u + q < P.nfield) P.field[4 * u + q] = P.field_std * v[q];
  }
}


// ---- the GMM draw from a subject's code volume (fsg_sample_head_codes_f32), as a job of `nblk` workgroups ---------------------
struct GmmCodesK {
  const uint8_t* codes;   // uint16 codes, 16-byte aligned
  const uint8_t* tuples;  // [ntuples][stride]
  int ntuples, stride;
  uint32_t sel;           // the four selected bytes of a row, 8 bits each
  size_t n;               // voxels, a multiple of 8
  const float* mus;
  const float* sigmas;
  int ntab;
  uint64_t seed, stream_id;
  float* out;
};

// code_ms: FSG_CODES_MAX float2 of LDS.  Contains one __syncthreads(): every thread of the workgroup calls it.
__device__ __forceinline__ void gmm_codes_job(const GmmCodesK& G, float2* code_ms, const unsigned blk, const unsigned nblk) {
  // (mu, sigma) of every code of the subject under this sample's selection, then 2 + 4 bytes per voxel
  const uint8_t* __restrict__ tup = G.tuples;
  for (int c = threadIdx.x; c < G.ntuples; c += blockDim.x) {
    const uint8_t* row = tup + (size_t)c * G.stride;
    const int lab = (row[G.sel & 255u] + row[(G.sel >> 8) & 255u] + row[(G.sel >> 16) & 255u] + row[G.sel >> 24]) & 255;
    code_ms[c] = lab < G.ntab ? make_float2(G.mus[lab], G.sigmas[lab]) : make_float2(0.f, 0.f);
  }
  __syncthreads();
  // ONE 16-byte load of eight codes per lane and trip, then a Philox block, four table look-ups and a 16-byte store per group
  // of four.  It is the width of a wave's read request that this kernel follows, not its bytes: uint8 codes (subjects with
  // <= 256 columns) in 4 / 8 / 16-byte loads ran at 35.1 / 33.0 / 38.3 us (the last with one trip per thread), uint16 codes in
  // 8 / 16-byte loads at 31.6 / 30.8 (event intervals of the bench); the one-byte form was dropped.
  const uint32_t step = nblk * blockDim.x;
  const uint8_t* __restrict__ codes = G.codes;
  auto draw4 = [&](uint32_t g, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    const float4 r = fsg_randn4<true>(G.seed, G.stream_id, (uint64_t)g);
    const float2 m0 = code_ms[c0], m1 = code_ms[c1], m2 = code_ms[c2], m3 = code_ms[c3];
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(G.out) + (size_t)g * 16u) =
        make_float4(fmaxf(m0.x + m0.y * r.x, 0.f), fmaxf(m1.x + m1.y * r.y, 0.f), fmaxf(m2.x + m2.y * r.z, 0.f),
                    fmaxf(m3.x + m3.y * r.w, 0.f));
  };
  const uint32_t nt = (uint32_t)(G.n >> 3);
  for (uint32_t t = blk * blockDim.x + threadIdx.x; t < nt; t += step) {
    const uint4 w = *reinterpret_cast<const uint4*>(codes + (size_t)t * 16u);
    draw4(2 * t, w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16);
    draw4(2 * t + 1, w.z & 0xFFFFu, w.z >> 16, w.w & 0xFFFFu, w.w >> 16);
  }
}

}  // namespace fsg_ride
