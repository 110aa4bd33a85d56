// fsg_keyed.hip -- keyed mode: every per-sample draw from Philox4x32-10 under the sample's 64-bit key.
//
// What the reference does per sample on the host (paths relative to /root/reference/fetalsyngen/):
//   generator/intensity/rand_gmm.py:82-85,:120-148     randint x4, rand(50) x2, randn(41)
//   generator/deformation/affine_nonrigid.py:140-145    gate, flip
//                                          :248-263    rand(3) x3 -> make_affine_matrix (utils/generation.py:39-71)
//                                          :271-290    torch.rand(3, float64) -> rotation centre
//                                          :303-318    nonlin scale / std, randn(s,s,s,3)
//   generator/augmentation/synthseg.py:263-268         gamma gate, exp(gamma_std * randn)
//                                     :157-176         bias gate, scale, std, randn(b,b,b)
//                                     :63-84           resample gate, spacing, std factor, sizes
//                                     :218-223         noise gate, std
// Here: the same quantities, the same arithmetic from draw to parameter (association order kept), the draws themselves from
// Philox4x32-10 keyed by the sample.  Scalars: host, slot s of the sample = counter (s, 0, 0, 0).  Small tensors: ONE device
// launch writing the sample's parameter block (streams 3..6 of the key).  Large fields: in the consuming kernels (streams 1, 2),
// exactly as "device" mode does.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <new>
#include "fsg_common.h"
#include "fsg_ride.h"

namespace {

constexpr int KT_CAP = 1024;  // largest grid size a table can be registered for

struct KeyedCtx {
  fsg_keyed_config cfg;
  const fsg_tap* tab[4][3][KT_CAP + 1];
  int64_t block_bytes;
  int max_field[3], max_bias[3];
};

// ---- host Philox4x32-10 (same constants and round function as fsg_philox4x32_10) --------------------------------------
struct U4 { uint32_t x, y, z, w; };
inline U4 philox_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += W0; k1 += W1;
  }
  return U4{c0, c1, c2, c3};
}
// host slot s of sample `key`: stream 0
inline U4 slot_bits(uint64_t key, uint32_t s) { return philox_host(s, 0u, 0u, 0u, (uint32_t)key, (uint32_t)(key >> 32)); }
// uniform double in [0, 1): 53 bits of (x, y)
inline double slot_u(uint64_t key, uint32_t s) {
  const U4 r = slot_bits(key, s);
  const uint64_t b = ((uint64_t)r.x << 32) | r.y;
  return (double)(b >> 11) * (1.0 / 9007199254740992.0);
}
// standard normal (Box-Muller on the slot's two 53-bit uniforms; u1 in (0, 1])
inline double slot_n(uint64_t key, uint32_t s) {
  const U4 r = slot_bits(key, s);
  const uint64_t a = ((uint64_t)r.x << 32) | r.y, b = ((uint64_t)r.z << 32) | r.w;
  const double u1 = (double)((a >> 11) + 1) * (1.0 / 9007199254740992.0);
  const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
}

enum Slot : uint32_t {
  S_SUB0 = 0, S_DEFORM = 4, S_FLIP = 5, S_ROT = 6, S_SHEAR = 9, S_SCALE = 12, S_SHIFT = 15, S_NL_SCALE = 18, S_NL_STD = 19,
  S_GAMMA_GATE = 20, S_GAMMA = 21, S_BIAS_GATE = 22, S_BF_SCALE = 23, S_BF_STD = 24, S_RES_GATE = 25, S_SPACING = 26,
  S_RES_STD = 27, S_NOISE_GATE = 28, S_NOISE_STD = 29
};

inline int align256(int v) { return (v + 255) & ~255; }

// utils/generation.py:39-71: shear(x) @ shear(y) @ shear(z) @ Rx @ Ry @ Rz, rows scaled; float64, then cast to float32
void affine_matrix(const double rot[3], const double sh[3], const double sc[3], float A[9]) {
  auto mul = [](const double a[9], const double b[9], double o[9]) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  };
  const double c0 = std::cos(rot[0]), c1 = std::cos(rot[1]), c2 = std::cos(rot[2]);
  const double s0 = std::sin(rot[0]), s1 = std::sin(rot[1]), s2 = std::sin(rot[2]);
  const double shx[9] = {1, 0, 0, sh[1], 1, 0, sh[2], 0, 1}, shy[9] = {1, sh[0], 0, 0, 1, 0, 0, sh[2], 1},
               shz[9] = {1, 0, sh[0], 0, 1, sh[1], 0, 0, 1};
  const double rx[9] = {1, 0, 0, 0, c0, -s0, 0, s0, c0}, ry[9] = {c1, 0, s1, 0, 1, 0, -s1, 0, c1},
               rz[9] = {c2, -s2, 0, s2, c2, 0, 0, 0, 1};
  double t0[9], t1[9];
  mul(shx, shy, t0); mul(t0, shz, t1); mul(t1, rx, t0); mul(t0, ry, t1); mul(t1, rz, t0);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[3 * i + j] = (float)(t0[3 * i + j] * sc[i]);
}

// utils/generation.py:74-81 in float32: t = -h..h, g = exp(-((t / sigma)^2) / 2), normalised by the ascending sum
int gaussian_taps(double sigma, float* taps, int cap) {
  const int half = (int)std::ceil(3.0 * sigma);
  const int n = 2 * half + 1;
  if (n > cap) return -1;
  const float sg = (float)sigma;
  float sum = 0.f;
  for (int t = 0; t < n; ++t) {
    const float q = (float)(t - half) / sg;
    taps[t] = std::exp(-(q * q) / 2.f);
    sum += taps[t];
  }
  for (int t = 0; t < n; ++t) taps[t] = taps[t] / sum;
  return n;
}

int draw(const KeyedCtx& C, uint64_t key, fsg_keyed_draws& d) {
  const fsg_keyed_config& c = C.cfg;
  std::memset(&d, 0, sizeof d);
  d.key = key;
  d.ntab = c.nlabels;
  const int span = c.max_subclusters - c.min_subclusters + 1;
  for (int m = 0; m < c.meta_labels; ++m) {
    int pick = c.min_subclusters + (int)(slot_u(key, S_SUB0 + m) * span);
    d.subclusters[m] = pick > c.max_subclusters ? c.max_subclusters : pick;
  }
  // ---- deformation
  if (slot_u(key, S_DEFORM) < c.deform_prob) {
    d.deform_active = 1;
    d.flip = slot_u(key, S_FLIP) < c.flip_prb ? 1 : 0;
    for (int a = 0; a < 3; ++a) {
      d.rotations[a] = (2 * c.max_rotation * slot_u(key, S_ROT + a) - c.max_rotation) / 180.0 * 3.141592653589793;
      d.shears[a] = 2 * c.max_shear * slot_u(key, S_SHEAR + a) - c.max_shear;
      d.scalings[a] = 1 + (2 * c.max_scaling * slot_u(key, S_SCALE + a) - c.max_scaling);
    }
    affine_matrix(d.rotations, d.shears, d.scalings, d.A);
    for (int a = 0; a < 3; ++a) {
      // float32 centre + float64 shift (affine_nonrigid.py:271-290); with no room the shift is exactly 0
      const double centre = (double)(float)((c.shape[a] - 1) / 2.0);
      double room = (double)((float)(c.shape[a] - c.size[a]) / 2.f);
      if (room < 0) room = 0;
      d.c2[a] = centre + (2 * (room * slot_u(key, S_SHIFT + a)) - room);
    }
    if (c.nonlinear) {
      d.nonlinear = 1;
      d.nonlin_scale = c.nonlin_scale_min + slot_u(key, S_NL_SCALE) * (c.nonlin_scale_max - c.nonlin_scale_min);
      for (int a = 0; a < 3; ++a) d.field_dims[a] = (int)std::nearbyint(d.nonlin_scale * (double)c.shape[a]);
      d.nonlin_std = c.nonlin_std_max * slot_u(key, S_NL_STD);
    }
  }
  // ---- gamma
  if (slot_u(key, S_GAMMA_GATE) < c.gamma_prob) {
    d.gamma_active = 1;
    d.gamma = std::exp(c.gamma_std * slot_n(key, S_GAMMA));
  }
  // ---- bias field
  if (slot_u(key, S_BIAS_GATE) < c.bias_prob) {
    d.bias_active = 1;
    d.bf_scale = c.bf_scale_min + slot_u(key, S_BF_SCALE) * (c.bf_scale_max - c.bf_scale_min);
    for (int a = 0; a < 3; ++a) {
      const int v = (int)std::nearbyint(d.bf_scale * (double)c.shape[a]);
      d.bias_dims[a] = v < 1 ? 1 : v;
    }
    d.bf_std = c.bf_std_min + (c.bf_std_max - c.bf_std_min) * slot_u(key, S_BF_STD);
  }
  // ---- resampling (synthseg.py:63-84)
  for (int a = 0; a < 3; ++a) d.low_shape[a] = c.shape[a];
  if (slot_u(key, S_RES_GATE) < c.resample_prob) {
    d.resample_active = 1;
    d.spacing = c.min_resolution + (c.max_resolution - c.min_resolution) * slot_u(key, S_SPACING);
    d.u_std = slot_u(key, S_RES_STD);
    for (int a = 0; a < 3; ++a) {
      double sd = (0.85 + 0.3 * d.u_std) * 1.6094379124341003 / 3.141592653589793 * d.spacing / c.resolution[a];
      if (d.spacing <= c.resolution[a]) sd = 0.0;
      d.stds[a] = sd;
      d.low_shape[a] = (int)((double)c.shape[a] * c.resolution[a] / d.spacing);
      d.blur_ntaps[a] = sd > 0 ? 2 * (int)std::ceil(3.0 * sd) + 1 : 0;
    }
  }
  // ---- noise
  if (slot_u(key, S_NOISE_GATE) < c.noise_prob) {
    d.noise_active = 1;
    d.noise_std = c.noise_std_min + (c.noise_std_max - c.noise_std_min) * slot_u(key, S_NOISE_STD);
    d.noise_std32 = (float)d.noise_std;
  }
  // ---- layout of the device parameter block
  int off = 0;
  d.off_mm8 = off; off = align256(off + 8 * 4);
  d.off_slots = off; off = align256(off + FSG_MM_NSLOTS * FSG_MM_SLOT_STRIDE * 4);
  d.off_mus = off; off = align256(off + 256 * 4);
  d.off_sigmas = off; off = align256(off + 256 * 4);
  d.off_bias = off; off = align256(off + (d.bias_active ? d.bias_dims[0] * d.bias_dims[1] * d.bias_dims[2] * 4 : 0));
  d.off_field = off; off = align256(off + (d.nonlinear ? d.field_dims[0] * d.field_dims[1] * d.field_dims[2] * 12 : 0));
  d.block_bytes = off;
  return 0;
}

// ---- the draw kernel -------------------------------------------------------------------------------------------------------
using fsg_ride::DrawK;

__global__ __launch_bounds__(256) void keyed_draw_kernel(const DrawK P) { fsg_ride::keyed_draw_body(P, (int)blockIdx.x); }

// the draw job's argument block and its number of workgroups
int fill_draw(const KeyedCtx& C, const fsg_keyed_draws& d, void* block_dev, DrawK& P, unsigned& grid) {
  if (!block_dev || ((uintptr_t)block_dev & 15)) return FSG_E_BADARG;
  char* base = (char*)block_dev;
  P.key = d.key;
  P.mm8 = (int32_t*)(base + d.off_mm8);
  P.slots = (int32_t*)(base + d.off_slots);
  P.mus = (float*)(base + d.off_mus);
  P.sigmas = (float*)(base + d.off_sigmas);
  P.bias = (float*)(base + d.off_bias);
  P.field = (float*)(base + d.off_field);
  P.nlabels = C.cfg.nlabels; P.nseed = C.cfg.n_seed_labels; P.tie = C.cfg.tie_classes;
  P.nbias = d.bias_active ? d.bias_dims[0] * d.bias_dims[1] * d.bias_dims[2] : 0;
  P.nfield = d.nonlinear ? d.field_dims[0] * d.field_dims[1] * d.field_dims[2] * 3 : 0;
  // the reference scales by float32 tensors: std32 * randn (synthseg.py:172-176: torch.tensor(bf_std) float32;
  // affine_nonrigid.py:318: a Python float times a float32 tensor = float32 multiply by the rounded scalar)
  P.bias_std = (float)d.bf_std;
  P.field_std = (float)d.nonlin_std;
  std::memcpy(P.seed_labels, C.cfg.seed_labels, 256);
  std::memcpy(P.gen_classes, C.cfg.generation_classes, 256);
  const int work = ((P.nbias + 3) >> 2) + ((P.nfield + 3) >> 2);
  grid = 1u + (unsigned)((work + 255) / 256);
  return 0;
}

int launch_draw(const KeyedCtx& C, const fsg_keyed_draws& d, void* block_dev, void* stream) {
  DrawK P;
  unsigned grid = 0;
  const int rc = fill_draw(C, d, block_dev, P, grid);
  if (rc) return rc;
  hipLaunchKernelGGL(keyed_draw_kernel, dim3(grid), dim3(256), 0, fsg_stream(stream), P);
  FSG_RETURN_LAUNCH();
}

}  // namespace

extern "C" {

int fsg_keyed_create(const fsg_keyed_config* cfg, void** ctx) {
  if (!cfg || !ctx) return FSG_E_BADARG;
  const fsg_keyed_config& c = *cfg;
  for (int a = 0; a < 3; ++a)
    if (c.shape[a] <= 0 || c.shape[a] > KT_CAP || c.size[a] <= 0 || !(c.resolution[a] > 0)) return FSG_E_BADARG;
  if (c.meta_labels < 1 || c.meta_labels > 4 || c.min_subclusters < 1 || c.max_subclusters < c.min_subclusters ||
      c.max_subclusters - c.min_subclusters >= 16)
    return FSG_E_BADARG;
  if (c.nlabels < 1 || c.nlabels > 256 || c.n_seed_labels < 0 || c.n_seed_labels > 256) return FSG_E_BADARG;
  for (int j = 0; j < c.n_seed_labels; ++j)
    if (c.seed_labels[j] >= c.nlabels || c.generation_classes[j] >= c.nlabels) return FSG_E_BADARG;
  if (!(c.min_resolution > 0) || c.max_resolution < c.min_resolution) return FSG_E_BADARG;
  KeyedCtx* K = new (std::nothrow) KeyedCtx();
  if (!K) return FSG_E_BADARG;
  K->cfg = c;
  std::memset(K->tab, 0, sizeof K->tab);
  // largest grids the configuration can draw -> size of the parameter block
  int64_t nb = 1, nf = 3;
  for (int a = 0; a < 3; ++a) {
    int b = (int)std::nearbyint(c.bf_scale_max * c.shape[a]);
    K->max_bias[a] = b < 1 ? 1 : b;
    K->max_field[a] = (int)std::nearbyint(c.nonlin_scale_max * c.shape[a]);
    nb *= K->max_bias[a];
    nf *= K->max_field[a] > 0 ? K->max_field[a] : 1;
  }
  int64_t off = 0;
  off = align256((int)(off + 32));
  off = align256((int)(off + FSG_MM_NSLOTS * FSG_MM_SLOT_STRIDE * 4));
  off = align256((int)(off + 1024));
  off = align256((int)(off + 1024));
  off += (nb * 4 + 255) / 256 * 256;
  off += (nf * 4 + 255) / 256 * 256;
  K->block_bytes = off;
  *ctx = K;
  return 0;
}

int fsg_keyed_destroy(void* ctx) {
  delete (KeyedCtx*)ctx;
  return 0;
}

int fsg_keyed_set_table(void* ctx, int kind, int axis, int n, const fsg_tap* table_dev) {
  KeyedCtx* K = (KeyedCtx*)ctx;
  if (!K || kind < 0 || kind > 3 || axis < 0 || axis > 2 || n < 1 || n > KT_CAP) return FSG_E_BADARG;
  K->tab[kind][axis][n] = table_dev;
  return 0;
}

int64_t fsg_keyed_block_bytes(void* ctx) { return ctx ? ((KeyedCtx*)ctx)->block_bytes : (int64_t)FSG_E_BADARG; }

int fsg_keyed_draw(void* ctx, uint64_t key, fsg_keyed_draws* out) {
  if (!ctx || !out) return FSG_E_BADARG;
  return draw(*(KeyedCtx*)ctx, key, *out);
}

int fsg_keyed_fill_block(void* ctx, const fsg_keyed_draws* d, void* block_dev, void* stream) {
  if (!ctx || !d) return FSG_E_BADARG;
  return launch_draw(*(KeyedCtx*)ctx, *d, block_dev, stream);
}

int fsg_keyed_sample_run(void* ctx, const int64_t* iv, int niv, fsg_keyed_draws* draws_out, void* stream) {
  KeyedCtx* K = (KeyedCtx*)ctx;
  if (!K || !iv || niv < FSG_KEYED_I_COUNT) return FSG_E_BADARG;
  const fsg_keyed_config& c = K->cfg;
  fsg_keyed_draws d;
  int rc = draw(*K, (uint64_t)iv[FSG_KEYED_I_KEY], d);
  if (rc) return rc;
  if (draws_out) *draws_out = d;
  if (d.block_bytes > K->block_bytes) return FSG_E_TOOBIG;
  char* base = (char*)(uintptr_t)iv[FSG_KEYED_I_BLOCK];
  if (!base) return FSG_E_BADARG;

  fsg_sample_plan q;
  std::memset(&q, 0, sizeof q);
  for (int a = 0; a < 3; ++a) q.shape[a] = c.shape[a];
  for (int m = 0; m < c.meta_labels; ++m) {
    q.label_parts[m] = (const uint8_t*)(uintptr_t)iv[FSG_KEYED_I_BANK + 4 * (d.subclusters[m] - c.min_subclusters) + m];
    if (!q.label_parts[m]) return FSG_E_BADARG;
  }
  if (iv[FSG_KEYED_I_CODES] && iv[FSG_KEYED_I_CODE_TUPLES]) {  // the subject's code volume: selection = one byte of a tuple per meta label
    q.label_codes = (const uint16_t*)(uintptr_t)iv[FSG_KEYED_I_CODES];
    q.code_tuples = (const uint8_t*)(uintptr_t)iv[FSG_KEYED_I_CODE_TUPLES];
    q.code_ntuples = (int32_t)iv[FSG_KEYED_I_CODE_NTUPLES];
    q.code_stride = (int32_t)iv[FSG_KEYED_I_CODE_STRIDE];
    if (q.code_stride < 4 * (c.max_subclusters - c.min_subclusters + 1) + 1) return FSG_E_BADARG;
    for (int m = 0; m < 4; ++m)
      q.code_sel[m] = m < c.meta_labels ? 4 * (d.subclusters[m] - c.min_subclusters) + m : q.code_stride - 1;
  }
  q.mus = (const float*)(base + d.off_mus);
  q.sigmas = (const float*)(base + d.off_sigmas);
  q.ntab = d.ntab;
  q.gmm_seed = d.key;
  q.gmm_stream = 1;
  if (d.deform_active) {
    q.deform_active = 1;
    for (int a = 0; a < 3; ++a) {
      q.deform.shape[a] = c.shape[a];
      q.deform.centre[a] = (float)((c.size[a] - 1) / 2.0);
      q.deform.c2[a] = (float)d.c2[a];
    }
    for (int a = 0; a < 9; ++a) q.deform.A[a] = d.A[a];
    q.deform.flip = d.flip;
    if (d.nonlinear) {
      const fsg_tap* t[3];
      for (int a = 0; a < 3; ++a) {
        if (d.field_dims[a] < 1 || d.field_dims[a] > KT_CAP) return FSG_E_TOOBIG;
        t[a] = K->tab[FSG_KT_FIELD][a][d.field_dims[a]];
        if (!t[a]) return FSG_E_NOTABLE;
        q.deform.field_dims[a] = d.field_dims[a];
      }
      q.deform.field = (const float*)(base + d.off_field);
      q.deform.tx = t[0]; q.deform.ty = t[1]; q.deform.tz = t[2];
    }
    q.seg_in = (const float*)(uintptr_t)iv[FSG_KEYED_I_SEG_IN];
    q.seg_in_u8 = (const uint8_t*)(uintptr_t)iv[FSG_KEYED_I_SEG_IN_U8];
    q.seg_out = (float*)(uintptr_t)iv[FSG_KEYED_I_SEG_OUT];
    q.seg_out_u8 = (uint8_t*)(uintptr_t)iv[FSG_KEYED_I_SEG_OUT_U8];
  }
  q.epi.gamma = d.gamma_active ? (float)d.gamma : 0.f;
  if (d.bias_active) {
    const fsg_tap* t[3];
    for (int a = 0; a < 3; ++a) {
      if (d.bias_dims[a] > KT_CAP) return FSG_E_TOOBIG;
      t[a] = K->tab[FSG_KT_BIAS][a][d.bias_dims[a]];
      if (!t[a]) return FSG_E_NOTABLE;
      q.epi.bias_dims[a] = d.bias_dims[a];
    }
    q.epi.bias = (const float*)(base + d.off_bias);
    q.epi.bx = t[0]; q.epi.by = t[1]; q.epi.bz = t[2];
  }
  if (d.resample_active) {
    q.resample_active = 1;
    for (int a = 0; a < 3; ++a) {
      const int m = d.low_shape[a];
      if (m < 1 || m > KT_CAP) return FSG_E_TOOBIG;
      q.low_shape[a] = m;
      q.rs_tab[a] = K->tab[FSG_KT_RESAMPLE][a][m];
      q.back_tab[a] = K->tab[FSG_KT_BACK][a][m];
      if (!q.rs_tab[a] || !q.back_tab[a]) return FSG_E_NOTABLE;
      if (d.stds[a] > 0) {
        const int n = gaussian_taps(d.stds[a], q.blur_taps[a], 129);
        if (n < 0) return FSG_E_TOOBIG;
        q.blur_ntaps[a] = n;
      }
    }
  }
  if (d.noise_active) {
    q.noise_mode = 2;
    q.noise_seed = d.key;
    q.noise_stream = 2;
    q.noise_std = d.noise_std32;
  }
  q.scale01 = (int32_t)iv[FSG_KEYED_I_SCALE01];
  q.ws0 = (float*)(uintptr_t)iv[FSG_KEYED_I_WS0];
  q.ws1 = (float*)(uintptr_t)iv[FSG_KEYED_I_WS1];
  q.ws_low = (float*)(uintptr_t)iv[FSG_KEYED_I_WS_LOW];
  q.ws_rows = (float*)(uintptr_t)iv[FSG_KEYED_I_WS_ROWS];
  q.row_stride = (int32_t)iv[FSG_KEYED_I_ROW_STRIDE];
  q.mm8 = (int32_t*)(base + d.off_mm8);
  q.mm8_preset = 1;
  q.mm_slots = d.resample_active ? (int32_t*)(base + d.off_slots) : nullptr;
  q.mm_nslots = d.resample_active ? FSG_MM_NSLOTS : 0;
  q.out = (float*)(uintptr_t)iv[FSG_KEYED_I_OUT];
  q.trace_events = (void**)(uintptr_t)iv[FSG_KEYED_I_TRACE_EVENTS];
  q.trace_ids = (int32_t*)(uintptr_t)iv[FSG_KEYED_I_TRACE_IDS];
  q.trace_cap = (int32_t)iv[FSG_KEYED_I_TRACE_CAP];
  q.ev_blur_begin = (void*)(uintptr_t)iv[FSG_KEYED_I_EV_BLUR_BEGIN];
  q.ev_blur_end = (void*)(uintptr_t)iv[FSG_KEYED_I_EV_BLUR_END];

  // ---- look-ahead: the caller names the next sample of this stream; its draw job rides in this sample's floor(min) launch -----
  const int64_t flags = iv[FSG_KEYED_I_FLAGS];
  const bool draw_done = (flags & 1) != 0;  // this sample's block was filled beside the previous sample's floor(min) pass
  DrawK next_draw;
  int32_t rode = 0;
  char* nbase = (char*)(uintptr_t)iv[FSG_KEYED_I_NEXT_BLOCK];
  if (nbase && (flags & 4)) {
    fsg_keyed_draws dn;
    rc = draw(*K, (uint64_t)iv[FSG_KEYED_I_NEXT_KEY], dn);
    if (rc) return rc;
    if (dn.block_bytes > K->block_bytes) return FSG_E_TOOBIG;
    unsigned grid = 0;
    rc = fill_draw(*K, dn, nbase, next_draw, grid);
    if (rc) return rc;
    q.ride_draw = &next_draw;
    q.ride_draw_blocks = grid;
    q.rode = &rode;
  }

  if (!draw_done && q.trace_events && q.trace_ids && q.trace_cap > 1) {  // stage trace: an event before the draw kernel, the next one behind it
    hipError_t e = hipEventRecord((hipEvent_t)q.trace_events[0], fsg_stream(stream));
    if (e != hipSuccess) return (int)e;
    q.trace_ids[0] = FSG_ST_BEGIN;
    q.trace_ids[q.trace_cap] = 1;
    q.trace_start = 1;
    q.trace_first_id = FSG_ST_DRAW;
  }
  if (!draw_done) {
    rc = launch_draw(*K, d, base, stream);
    if (rc) return rc;
  }
  rc = fsg_sample_run(&q, stream);
  if (draws_out) draws_out->rode = rode;
  return rc;
}

}  // extern "C"
