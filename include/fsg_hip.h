/*
 * fsg_hip.h -- C ABI of libfsg_hip.so: the MI355X (gfx950) kernels behind the
 * FetalSynthGen per-volume synthesis hot path.
 *
 * The reference (Medical-Image-Analysis-Laboratory/fetalsyngen) has no FFI on this path: it is
 * pure PyTorch tensor code.  The drop-in boundary is therefore the reference's Python class
 * surface (mirrored in fetalsyngen_amd/), and THIS header is what that mirror binds with ctypes
 * instead of calling ATen.  Each entry point names the reference code it replaces
 * (paths relative to /root/reference/fetalsyngen/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all buffers;
 *   - volumes are C-contiguous (x, y, z) with z fastest, fp32 unless the name says otherwise;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous, no host sync inside,
 *     no allocation inside (safe to capture into a hipGraph);
 *   - return value: 0 = success, >0 = hipError_t of the launch, <0 = FSG_E_* argument error;
 *   - float arithmetic follows the reference's operation order with FMA contraction OFF in the
 *     coordinate / interpolation paths, so label volumes and sampling positions are bit-identical
 *     to the reference's CPU path on identical inputs.
 */
#ifndef FSG_HIP_H
#define FSG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSG_ABI_VERSION 1

#define FSG_E_BADARG (-1)   /* null pointer / non-positive size / bad enum */
#define FSG_E_TOOBIG (-2)   /* size exceeds what the kernel indexes (2^31-1 voxels per volume) */
#define FSG_E_ALIGN  (-3)   /* pointer not aligned as the kernel requires */

/* One output sample of a separable linear resample along one axis:
 * value = w_lo * src[lo] + w_hi * src[hi].  lo < 0 marks "outside" (result 0 for the voxel). */
typedef struct fsg_tap {
  int32_t lo;
  int32_t hi;
  float w_lo;
  float w_hi;
} fsg_tap;

/* Parameters of the spatial deformation (affine o nonlinear field), see fsg_coords_* / fsg_warp_*.
 * Passed BY POINTER from host memory; copied into the kernel arguments at launch. */
struct fsg_epilogue;

typedef struct fsg_deform {
  int32_t shape[3];      /* grid being generated == shape of the volumes being sampled            */
  float A[9];            /* row-major 3x3 affine, fp32 (affine_nonrigid.py:265-269)                */
  float centre[3];       /* (size-1)/2 of the configured size (affine_nonrigid.py:77-84)           */
  float c2[3];           /* rotation centre + shift, rounded to fp32 (affine_nonrigid.py:271-290)  */
  int32_t flip;          /* sample from the volume mirrored along axis 0 (affine_nonrigid.py:180)  */
  int32_t field_dims[3]; /* coarse displacement grid (s0,s1,s2); all 0 => no nonlinear field       */
  const float* field;    /* DEVICE (s0,s1,s2,3) fp32, channel-last, already scaled by nonlin_std   */
  const fsg_tap* tx;     /* DEVICE per-axis zoom tables coarse->shape, lengths shape[0..2]         */
  const fsg_tap* ty;
  const fsg_tap* tz;
  const float* rows;     /* DEVICE, optional (NULL = off): per-(x,y) coarse rows from fsg_deform_rows_f32 */
  int32_t row_stride;    /* floats per row in `rows`, >= 3*field_dims[2] (+ bias_dims[2] when fused)      */
} fsg_deform;

int fsg_abi_version(void);
const char* fsg_error_string(int code);

/* Process-wide tuning switches (tests use them to cross-check the tuned kernels against the plain
 * ones); returns the previous flags.  Results are within the documented tolerances either way. */
#define FSG_TUNE_GENERIC_WARP 1  /* per-voxel field evaluation instead of the row-wise LDS kernels */
#define FSG_TUNE_PRECISE_MATH 2  /* OCML powf/expf in the gamma/bias epilogue instead of v_log/v_exp */
#define FSG_TUNE_GENERIC_ZOOM 4  /* per-voxel 8-tap zoom instead of the row-wise LDS kernels */
#define FSG_TUNE_NO_PREFETCH 8   /* row-wise zoom without the register-prefetch pipeline */
#define FSG_TUNE_NO_PATCH 32     /* row kernel (4 waves, own rows) instead of the 16-wave lockstep patch kernel */
#define FSG_TUNE_BUFFER_LOADS 64 /* opt in: patch kernel body on raw buffer loads (fewer instructions, slower in r01) */
#define FSG_TUNE_BRICK 16        /* opt in: uint8-label warps through the LDS brick kernel (experimental, slower in r01) */
int fsg_set_tuning(int flags);

/* ---- RNG ------------------------------------------------------------------------------------ */
/* Standard-normal field from Philox4x32-10 keyed (seed, stream_id), element e uses counter e/4,
 * lane e%4, Box-Muller.  Exactly the values the fused kernels below use when noise == NULL, so
 * tests can hand the same noise to the CPU oracle.  Replaces torch.randn(shape, device=...)
 * (rand_gmm.py:146-148, synthseg.py:230-232) in device-RNG mode. */
int fsg_randn_f32(float* out, size_t n, uint64_t seed, uint64_t stream_id, void* stream);

/* ---- K1: GMM intensity draw (generator/intensity/rand_gmm.py:146-149) ------------------------- */
/* out[v] = max(mus[l] + sigmas[l] * z[v], 0), l = labels[v] < ntab.
 * noise != NULL: z = noise (host-tape RNG mode); else z = Philox(seed, stream_id). */
int fsg_gmm_sample_u8(const uint8_t* labels, size_t n, const float* mus, const float* sigmas, int ntab,
                      const float* noise, uint64_t seed, uint64_t stream_id, float* out, void* stream);
int fsg_gmm_sample_i64(const int64_t* labels, size_t n, const float* mus, const float* sigmas, int ntab,
                       const float* noise, uint64_t seed, uint64_t stream_id, float* out, void* stream);
/* Same with the seed map given as the per-meta-label volumes load_seeds would sum (rand_gmm.py:90-99):
 * label = l0 + l1 + l2 + l3 formed on the fly (l1..l3 may be NULL).  Pointers 4-byte aligned, out 16-byte. */
int fsg_gmm_sample_u8x4(const uint8_t* l0, const uint8_t* l1, const uint8_t* l2, const uint8_t* l3, size_t n,
                        const float* mus, const float* sigmas, int ntab, const float* noise, uint64_t seed,
                        uint64_t stream_id, float* out, void* stream);
/* Same, and additionally resets `nmin` + `nmax` min/max keys at `mm` (as fsg_minmax_init would): the GMM draw is
 * the first kernel of a sample, folding the reset into it saves a launch.  mm may be NULL. */
int fsg_gmm_sample_u8x4_mm(const uint8_t* l0, const uint8_t* l1, const uint8_t* l2, const uint8_t* l3, size_t n,
                           const float* mus, const float* sigmas, int ntab, const float* noise, uint64_t seed,
                           uint64_t stream_id, float* out, int32_t* mm, int nmin, int nmax, void* stream);

/* Per-label count / sum / sum of squares of `values` (wave-level reductions); accumulates into the
 * caller-zeroed outputs.  Used to validate device-RNG GMM draws against mus/sigmas. */
int fsg_label_stats_u8(const uint8_t* labels, const float* values, size_t n, int nlabels,
                       unsigned long long* count, double* sum, double* sumsq, void* stream);

/* ---- separable linear zoom (utils/generation.py:310-397 myzoom_torch) -------------------------- */
/* src (sx,sy,sz,nch) -> dst (dx,dy,dz,nch), per-axis tables of lengths dx,dy,dz.  x, then y, then z,
 * each step w_lo*a + w_hi*b in fp32 without FMA.  Also the axis-aligned trilinear resample of
 * synthseg.py:87-104 (tables built from the float64 positions; lo<0 => 0). */
int fsg_zoom3d_f32(const float* src, int sx, int sy, int sz, int nch, const fsg_tap* tx, const fsg_tap* ty,
                   const fsg_tap* tz, float* dst, int dx, int dy, int dz, void* stream);

/* K7+K8 fused (synthseg.py:104 + :230-233): dst = max(zoom(src) + noise_std * z, 0); z as in K1.
 * noise_mode 0: no noise (dst = zoom(src)); 1: noise pointer; 2: Philox(seed, stream_id). */
int fsg_resample_noise_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                           const fsg_tap* tz, float* dst, int dx, int dy, int dz, int noise_mode,
                           const float* noise, uint64_t seed, uint64_t stream_id, float noise_std,
                           void* stream);

/* K9 pass A (synthseg.py:111-112): min and max of zoom(src) without storing it; mm[0]=min, mm[1]=max as
 * order-preserving int32 keys (see fsg_minmax_init). */
int fsg_zoom3d_minmax_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                          const fsg_tap* tz, int dx, int dy, int dz, int32_t* mm, void* stream);
/* K9 pass B (+K10): mode 0: dst = y / max            (synthseg.py:112, what FetalSynthGen.sample returns)
 *                   mode 1: t = y / max; dst = (t - mn/max) / (1 - mn/max)   (+ data/datasets.py:311) */
int fsg_zoom3d_normalise_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                             const fsg_tap* tz, float* dst, int dx, int dy, int dz, const int32_t* mm,
                             int mode, void* stream);

/* ---- K2/K3: deformation coordinates (affine_nonrigid.py:64-84, :299-366) ----------------------- */
/* Reset six int32 keys to (+inf,+inf,+inf,-inf,-inf,-inf) / two keys to (+inf,-inf). */
int fsg_minmax_init(int32_t* mm, int npairs_min, int npairs_max, void* stream);
/* Optional accelerator for fsg_coords_minmax_f32 / fsg_warp_*: x/y-interpolate the coarse displacement grid
 * (and the coarse bias grid of `epi_host`, may be NULL) once per (x,y) column into `rows`
 * (shape[0]*shape[1]*row_stride floats, caller-owned workspace); then set d->rows / d->row_stride.
 * Results are bit-identical with or without it. */
int fsg_deform_rows_f32(const fsg_deform* d_host, const struct fsg_epilogue* epi_host, float* rows, int row_stride,
                        void* stream);
/* min / max over the grid of the clamped coordinates, before margin subtraction; mm6 = {min x,y,z,
 * max x,y,z} as ordered int32 keys.  Reads only the coarse field. */
int fsg_coords_minmax_f32(const fsg_deform* d_host, int32_t* mm6, void* stream);
/* What the warp actually needs: keys in mm3[0..2] whose floor equals floor(min) of the clamped coordinate
 * per axis (caller initialises them with fsg_minmax_init(mm3, 3, 0)).  Evaluates the six faces of the grid
 * first; the full pass exits immediately once every axis has a voxel below 1 (coordinates are >= 0, so
 * floor(min) is then 0).  Usable wherever fsg_warp_* takes `mm6` (only entries 0..2 are read there).
 * Returns FSG_E_TOOBIG for coarse grids beyond the row kernels' capacity: use fsg_coords_minmax_f32. */
int fsg_coords_floormin_f32(const fsg_deform* d_host, int32_t* mm3, void* stream);
/* Materialise xx2, yy2, zz2 (after subtracting floor(min)) -- the tensors
 * SpatialDeformation.generate_deformation_and_flip returns.  mm6 from fsg_coords_minmax_f32. */
int fsg_coords_f32(const fsg_deform* d_host, const int32_t* mm6, float* xx, float* yy, float* zz, void* stream);

/* ---- K3+K4(+K5) fused warp (affine_nonrigid.py:164-193 + utils/generation.py:204-288) ---------- */
/* For every voxel of the grid: recompute its sampling position from the coarse field (no coordinate
 * volumes in HBM), then
 *   out_lin = trilinear(src_lin)  (strict >0 validity, x->y->z blend, 0 outside)  [+ optional epilogue]
 *   out_nn  = nearest(src_nn)     (round-half-even, index clamp)
 * Either pair may be NULL.  Epilogue on out_lin (synthseg.py:274 and :178-182), each optional:
 *   gamma > 0:      v = 300 * (v/300)^gamma
 *   bias != NULL:   v = v * exp(zoom(bias))  with bias (b0,b1,b2) and per-axis tables bx,by,bz. */
typedef struct fsg_epilogue {
  float gamma;           /* <= 0: skip */
  int32_t bias_dims[3];
  const float* bias;     /* DEVICE, NULL: skip */
  const fsg_tap* bx;
  const fsg_tap* by;
  const fsg_tap* bz;
} fsg_epilogue;
int fsg_warp_f32(const fsg_deform* d_host, const int32_t* mm6, const float* src_lin, float* out_lin,
                 const float* src_nn, float* out_nn, const fsg_epilogue* epi_host, void* stream);
/* Same with uint8 label volumes for the nearest leg (device-resident streaming path). */
int fsg_warp_f32_u8(const fsg_deform* d_host, const int32_t* mm6, const float* src_lin, float* out_lin,
                    const uint8_t* src_nn, uint8_t* out_nn, const fsg_epilogue* epi_host, void* stream);

/* Label volume read as uint8, deformed labels written as float32 (exact for 0..255): the reference keeps
 * segmentations as float32 tensors, the device-resident copy is uint8.  Served by the LDS brick kernel only
 * (FSG_TUNE_BRICK set, shape[2] % 4 == 0, 16-byte aligned fp32 volumes); returns FSG_E_ALIGN otherwise and
 * the caller uses fsg_warp_f32 / fsg_warp_f32_u8. */
int fsg_warp_f32_u8_to_f32(const fsg_deform* d_host, const int32_t* mm6, const float* src_lin, float* out_lin,
                           const uint8_t* src_nn, float* out_nn, const fsg_epilogue* epi_host, void* stream);

/* Generic gather with explicit coordinate volumes (utils/generation.py:204-288 fast_3D_interp_torch).
 * mode 0 = linear (default_value outside), 1 = nearest.  src (sx,sy,sz); npts coordinates. */
int fsg_interp3d_f32(const float* src, int sx, int sy, int sz, const float* ii, const float* jj,
                     const float* kk, size_t npts, int mode, float default_value, float* dst, void* stream);

/* ---- K5 stand-alone (synthseg.py:250-275, :144-188) ------------------------------------------- */
int fsg_gamma_f32(const float* x, size_t n, float gamma, float* out, void* stream);
int fsg_bias_mul_f32(const float* x, int nx, int ny, int nz, const float* bias, int b0, int b1, int b2,
                     const fsg_tap* bx, const fsg_tap* by, const fsg_tap* bz, float* out, void* stream);

/* ---- K6: separable Gaussian blur, one axis pass (utils/generation.py:84-110) -------------------- */
/* dst = conv1d(src, taps) along `axis` with zero padding (no border renormalisation); ntaps odd.
 * taps: DEVICE pointer.  src != dst. */
int fsg_blur_axis_f32(const float* src, float* dst, int nx, int ny, int nz, int axis, const float* taps,
                      int ntaps, void* stream);
/* Tuned path: taps passed from HOST memory (copied into the kernel arguments, <= 129 taps).  Needs
 * 16-byte aligned volumes and a z extent that is a multiple of 4; returns FSG_E_ALIGN otherwise (the
 * caller then uses fsg_blur_axis_f32).  Same result contract as fsg_blur_axis_f32. */
int fsg_blur_axis_taps_host_f32(const float* src, float* dst, int nx, int ny, int nz, int axis,
                                const float* taps_host, int ntaps, void* stream);

/* ---- K8 stand-alone (synthseg.py:206-235) ------------------------------------------------------ */
int fsg_add_noise_f32(const float* x, size_t n, const float* noise, uint64_t seed, uint64_t stream_id,
                      float noise_std, float* out, void* stream);

/* ---- K9/K10 stand-alone reductions / scaling --------------------------------------------------- */
int fsg_reduce_minmax_f32(const float* x, size_t n, int32_t* mm, void* stream); /* mm[0]=min key, mm[1]=max key */
/* mode 0: out = x / max;  mode 1: out = (x - min) / (max - min) (0*x if flat);  mode 2: (x-min)/(max-min)*255 */
int fsg_scale_f32(const float* x, size_t n, const int32_t* mm, int mode, float* out, void* stream);
/* Decode ordered keys on the host side helper (pure function, no GPU). */
float fsg_key_to_float(int32_t key);

/* ---- whole-sample launch sequence -------------------------------------------------------------------------- */
/* One call = the fused kernel sequence of FetalSynthGen.sample (generator/model.py:231-276) for the
 * seeds-based path: GMM draw -> [rows, margins, warp(+gamma+bias, labels)] -> [blur x,y,z, resample+noise,
 * zoom-back min/max, zoom-back+normalise] on `stream`.  All random quantities are inputs (drawn by the host in
 * the reference's order).  Workspaces are caller-owned and may be reused by the next call on the same stream.
 * Returns FSG_E_ALIGN / FSG_E_TOOBIG for configurations the fused kernels do not cover (the caller then issues
 * the entry points above one by one). */
typedef struct fsg_sample_plan {
  int32_t shape[3];
  /* K1 */
  const uint8_t* label_parts[4]; /* per-meta-label seed volumes (disjoint supports), [1..3] may be NULL      */
  const float* mus;
  const float* sigmas;
  int32_t ntab;
  const float* gmm_noise;        /* NULL: Philox(gmm_seed, gmm_stream)                                       */
  uint64_t gmm_seed, gmm_stream;
  /* K2..K5 */
  int32_t deform_active;
  fsg_deform deform;             /* rows / row_stride are filled in by the call                              */
  const float* seg_in;           /* float32 label volume and its deformed copy (deform_active only)          */
  float* seg_out;
  fsg_epilogue epi;              /* gamma <= 0: off; bias NULL: off (applied stand-alone when not deforming)  */
  /* K6..K8 */
  int32_t resample_active;
  int32_t low_shape[3];
  const fsg_tap* rs_tab[3];      /* full -> low resolution tables                                            */
  const fsg_tap* back_tab[3];    /* low -> full resolution tables                                            */
  int32_t blur_ntaps[3];         /* 0: axis not blurred                                                      */
  float blur_taps[3][129];
  int32_t noise_mode;            /* 0 none, 1 `noise` pointer, 2 Philox(noise_seed, noise_stream)            */
  const float* noise;
  uint64_t noise_seed, noise_stream;
  float noise_std;
  /* K9/K10 */
  int32_t scale01;               /* 1: also apply the dataset's [0,1] scaling (data/datasets.py:311)         */
  /* workspaces + result */
  float* ws0;                    /* shape[] floats each                                                       */
  float* ws1;
  float* ws_low;                 /* >= prod(low_shape) floats                                                 */
  float* ws_rows;                /* shape[0]*shape[1]*row_stride floats, or NULL                              */
  int32_t row_stride;
  int32_t* mm8;                  /* 8 x int32                                                                 */
  float* out;                    /* shape[] floats                                                            */
  void* ev_blur_begin;           /* optional hipEvent_t pair recorded around the blur passes (fsg_event_*)     */
  void* ev_blur_end;
} fsg_sample_plan;
int fsg_sample_run(const fsg_sample_plan* plan_host, void* stream);

/* hipEvent helpers for ctypes callers (timing on the launch stream). elapsed_ms synchronises on `end`. */
void* fsg_event_create(void);
int fsg_event_destroy(void* event);
int fsg_event_elapsed_ms(void* begin, void* end, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* FSG_HIP_H */
