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

#define FSG_ABI_VERSION 3  /* 2: fsg_sample_plan grew (mm_slots .. seg_out_u8) after version 1 shipped; 3: label_codes .. code_sel, FSG_KEYED_I_CODES */

#define FSG_E_BADARG (-1)   /* null pointer / non-positive size / bad enum */
#define FSG_E_TOOBIG (-2)   /* size exceeds what the kernel indexes (2^31-1 voxels per volume) */
#define FSG_E_ALIGN  (-3)   /* pointer not aligned as the kernel requires */
#define FSG_E_NOTABLE (-4)  /* keyed mode: a tap table this sample needs was not registered (fsg_keyed_set_table) */

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
#define FSG_TUNE_PRECISE_MATH 2  /* OCML powf/expf in the gamma/bias epilogue instead of v_log/v_exp; slice acquisition
                                    forward in the CUDA source's operation order instead of pair loads + lerps */
#define FSG_TUNE_GENERIC_ZOOM 4  /* per-voxel 8-tap zoom instead of the row-wise LDS kernels */
#define FSG_TUNE_NO_PREFETCH 8   /* row-wise zoom without the register-prefetch pipeline */
#define FSG_TUNE_NO_PATCH 32     /* row kernel (4 waves, own rows) instead of the 16-wave lockstep patch kernel */
#define FSG_TUNE_BUFFER_LOADS 64 /* opt in: patch kernel body on raw buffer loads (fewer instructions, slower in r01) */
#define FSG_TUNE_ROW_ZOOM 256    /* every zoom through the row-per-wave kernels (r01 default for K9) */
#define FSG_TUNE_TILE_ZOOM 512   /* every zoom through the tile kernel (default: only the noise epilogues) */
#define FSG_TUNE_NO_BLUR_FUSE 1024 /* blur: y and z passes as two launches */
#define FSG_TUNE_SA_DIRECT 128   /* slice-acquisition adjoint (interp_psf): direct global atomics, no LDS pre-summation */
#define FSG_TUNE_SPLIT_HEAD 2048 /* fsg_sample_run: GMM draw, per-row coarse values and six-face minimum as three launches */
#define FSG_TUNE_SLAB_ZOOM 8192 /* the slab zoom kernel also for the noise epilogues (default there: tile kernel) */
#define FSG_TUNE_NO_LEAN 4096  /* fused warp: the r01 patch kernel body instead of the lean body (fsg_warp_lean.hip) */
#define FSG_TUNE_NO_SEED_CODES 65536 /* A/B: the one-launch head reads the four label volumes even when the plan carries a code volume */
#define FSG_TUNE_SA_FWD_DIRECT 131072 /* slice-acquisition forward (linear PSF, no volume mask): direct global gathers for every slice */
#define FSG_TUNE_SA_FWD_PLATE 262144  /* ... the LDS plate kernel for every slice (default: chosen per slice by its orientation) */
#define FSG_TUNE_WAVE_ZOOM 32768 /* opt in: zooms without a noise draw through the wave kernel (independent waves; slower in r03) */
#define FSG_TUNE_NO_BLUR_RS 16384 /* fsg_sample_run: blur x3 + K7 as separate launches instead of the fused blur+resample pair */
#define FSG_TUNE_BRICK 16        /* opt in: uint8-label warps through the LDS brick kernel (experimental, slower in r01) */
int fsg_set_tuning(int flags);
/* Work shape of the fused warp kernel (speed only, results identical): 0 = 4x4 rows x 64 voxels per lockstep step,
 * 1 = 8x8 rows x 16 voxels, 2 = 4x8 rows x 32, 3 = 8x4 rows x 32, 4 = 4x16 rows x 16.  Returns the previous value. */
int fsg_warp_set_variant(int variant);

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

/* Tile shape of the zoom kernels: output y rows per workgroup (1..32, default 16) and the LDS floats reserved for the
 * x-blended source-row window (default 12288); a tile whose window does not fit is evaluated without staging. */
int fsg_zoom_set_tuning(int y_rows, int cap_floats);
/* K9 pass A (synthseg.py:111-112): min and max of zoom(src) without storing it; mm[0]=min, mm[1]=max as
 * order-preserving int32 keys (see fsg_minmax_init). */
int fsg_zoom3d_minmax_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                          const fsg_tap* tz, int dx, int dy, int dz, int32_t* mm, void* stream);
/* K9 pass B (+K10): mode 0: dst = y / max            (synthseg.py:112, what FetalSynthGen.sample returns)
 *                   mode 1: t = y / max; dst = (t - mn/max) / (1 - mn/max)   (+ data/datasets.py:311) */
int fsg_zoom3d_normalise_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                             const fsg_tap* tz, float* dst, int dx, int dy, int dz, const int32_t* mm,
                             int mode, void* stream);

/* Sharded form of the K9 pair (what fsg_sample_run uses): `slots` = nslots (2..64) slots of FSG_MM_SLOT_STRIDE int32 each,
 * slot s = {min key, max key, unused...}, every slot initialised by the caller to {key(+inf), key(-inf)} (the values
 * fsg_minmax_init writes).  The min/max pass updates slot (workgroup index % nslots); the normalise pass reduces the
 * slots itself.  Same results as the unsharded pair; the keys no longer sit on one contended address. */
#define FSG_MM_SLOT_STRIDE 16
int fsg_zoom3d_minmax_sharded_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                                  const fsg_tap* tz, int dx, int dy, int dz, int32_t* slots, int nslots, void* stream);
int fsg_zoom3d_normalise_sharded_f32(const float* src, int sx, int sy, int sz, const fsg_tap* tx, const fsg_tap* ty,
                                     const fsg_tap* tz, float* dst, int dx, int dy, int dz, const int32_t* slots, int nslots,
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
 * segmentations as float32 tensors, the device-resident copy is uint8.  Served by the lean warp kernel (per-row
 * coarse values prepared, coarse grids <= 32 entries along z, shape[2] <= 512) or, with FSG_TUNE_BRICK, by the LDS
 * brick kernel (shape[2] % 4 == 0, 16-byte aligned fp32 volumes); returns FSG_E_ALIGN otherwise and the caller
 * uses fsg_warp_f32 / fsg_warp_f32_u8. */
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

/* y pass then z pass of the same blur in ONE launch (the intermediate volume stays in LDS): bit-identical to
 * fsg_blur_axis_taps_host_f32(axis 1) followed by (axis 2).  Serves identical taps on both axes (the isotropic blur of
 * RandResample), radius 1..8, 16-byte aligned volumes, nz % 4 == 0; FSG_E_ALIGN otherwise.  Algorithmic bytes: two passes (16 B/voxel) for 8 B/voxel of HBM traffic. */
int fsg_blur_yz_taps_host_f32(const float* src, float* dst, int nx, int ny, int nz, const float* taps_y_host, int ntaps_y,
                              const float* taps_z_host, int ntaps_z, void* stream);

/* ---- K8 stand-alone (synthseg.py:206-235) ------------------------------------------------------ */
int fsg_add_noise_f32(const float* x, size_t n, const float* noise, uint64_t seed, uint64_t stream_id,
                      float noise_std, float* out, void* stream);

/* ---- K9/K10 stand-alone reductions / scaling --------------------------------------------------- */
int fsg_reduce_minmax_f32(const float* x, size_t n, int32_t* mm, void* stream); /* mm[0]=min key, mm[1]=max key */
/* mode 0: out = x / max;  mode 1: out = (x - min) / (max - min) (0*x if flat);  mode 2: (x-min)/(max-min)*255 */
int fsg_scale_f32(const float* x, size_t n, const int32_t* mm, int mode, float* out, void* stream);
/* dst[0..nbytes) = src[0..nbytes) by a kernel on `stream`; both 16-byte aligned, nbytes a multiple of 16.  `src` may be
 * pinned host memory: the per-sample parameter arena is uploaded this way so that the sample stays in one hardware queue. */
int fsg_copy_bytes(void* dst, const void* src, size_t nbytes, void* stream);
/* Decode ordered keys on the host side helper (pure function, no GPU). */
float fsg_key_to_float(int32_t key);

/* ---- SR-artifact slice-stack simulation ------------------------------------------------- */
/* Slice acquisition and its adjoint: the two operators of generator/artifacts/simulate_reco.py
 * (Scanner.scan :386-407 calls the forward, PSFreconstruction :38-54 the adjoint).  They replace the
 * reference's own CUDA extension, generator/artifacts/svort/slice_acquisition/slice_acq_cuda.cpp:22-160
 * (`forward`, `adjoint_forward`) and its kernels slice_acq_cuda_kernel.cu:17-171, :472-693.
 *   transforms (n,3,4) fp32 row-major [R|t], "translation first": p = R (pixel + t)
 *   vol (D,H,W) fp32, x fastest; psf (pd,ph,pw) fp32; slices (n,h,w) fp32
 *   vol_mask (D,H,W) / slices_mask (n,h,w): one byte per element (torch.bool), or NULL
 * mode selects which of the reference's two arithmetics is followed:
 *   FSG_SA_LINEAR      CUDA kernel, interp_psf=false: trilinear volume sample per PSF tap (:110-161, :607-666)
 *   FSG_SA_NEAREST_PSF CUDA kernel, interp_psf=true : nearest voxel (round half away from zero), PSF
 *                      re-interpolated at that voxel (:71-109, :572-605)
 *   FSG_SA_TORCH       the CPU fallback slice_acq.py:266-546 (nearest voxel by round-half-even, strict inside
 *                      test, raw PSF taps > 0, normalisation where weight > 1e-2, no per-pixel weight in the adjoint);
 *                      a 1x1x1 PSF without slices_weight takes the fallback's grid_sample route (slice_acq.py:381-384,
 *                      :445-480: trilinear, zero padding, slices_mask applied) in the forward
 * PSF limits: each extent <= 64, pd*ph*pw <= 4096 (it is staged in LDS), else FSG_E_TOOBIG. */
#define FSG_SA_LINEAR 0
#define FSG_SA_NEAREST_PSF 1
#define FSG_SA_TORCH 2
/* slices = A(vol) / weight where weight > 0 (CUDA modes; > 1e-2 in FSG_SA_TORCH), 0 elsewhere and outside
 * slices_mask; slices_weight (may be NULL) receives the weights.  Every element of both outputs is written. */
int fsg_slice_acq_forward_f32(const float* transforms, const float* vol, const uint8_t* vol_mask, const float* psf, int pd,
                              int ph, int pw, const uint8_t* slices_mask, float* slices, float* slices_weight, int D, int H,
                              int W, int n, int h, int w, float res_slice, int mode, void* stream);
/* vol = A^T(slices), vol_weight = A^T(1) (may be NULL); both are zeroed by the call, then accumulated with
 * fp32 atomics (as the reference does: the summation order is not deterministic).  CUDA modes skip pixels whose
 * PSF weight inside the volume is < 0.5 and divide each contribution by that weight (:560, :600, :616).
 * slice_ids (DEVICE, n x int32, may be NULL): transform z of the launch applies to slices[slice_ids[z]] (and that
 * row of slices_mask) -- the reconstruction keeps a random subset of the slices (simulate_reco.py:768-769)
 * without copying them.
 * Equalisation is a separate launch (fsg_equalize_f32), as in the reference (:1062-1074). */
int fsg_slice_acq_adjoint_f32(const float* transforms, const float* psf, int pd, int ph, int pw, const float* slices,
                              const uint8_t* slices_mask, const int32_t* slice_ids, const uint8_t* vol_mask, float* vol,
                              float* vol_weight, int D, int H, int W, int n, int h, int w, float res_slice, int mode,
                              void* stream);
/* Shape of the LDS pre-summation of the FSG_SA_NEAREST_PSF adjoint: accumulator cells (value + weight, 8 B each,
 * 256..18432), PSF planes per chunk, and the footprint (pixel pitch * 15 + PSF width, voxels) up to which 16x16-pixel
 * tiles are used instead of 8x8.  z_chunk 0 (default) derives the planes per chunk, and a larger capacity when one
 * plane does not fit, from the pixel pitch and the PSF size.  Defaults 3072 / 0 / 20.  Results do not depend on it beyond fp32 summation order. */
int fsg_slice_acq_set_tuning(int cap_cells, int z_chunk, int t16_extent);
/* vol[i] /= vol_weight[i] where vol_weight[i] > threshold (0 for the CUDA kernel :681-691, 1e-2 for the
 * fallback slice_acq.py:542-543); then vol[i] *= vol_mask[i] if vol_mask (fallback :544-545).  Either of
 * vol_weight / vol_mask may be NULL, not both. */
int fsg_equalize_f32(float* vol, const float* vol_weight, const uint8_t* vol_mask, float threshold, size_t n, void* stream);

/* ---- SR-artifact volumetric helpers, fsg_artifacts.hip ------------------------------------ */
/* out = clamp(sum_g exp(-(((x-c0)/s0)^2 + ((y-c1)/s1)^2 + ((z-c2)/s2)^2) / 2), 0, 1) over a (D,H,W) grid with
 * x the LAST axis -- generator/artifacts/utils.py:125-160 `mog_3d_tensor`, including its convention that a
 * centre is unpacked as (x0,y0,z0) (callers pass first-axis indices first: the reference's own transposition).
 * centers, sigmas: DEVICE (k,3) fp32.  tables: caller-owned workspace of k*3*max(D,H,W) floats. */
int fsg_mog3d_f32(const float* centers, const float* sigmas, int k, int D, int H, int W, float* tables, float* out,
                  void* stream);
/* Raw fractal Perlin noise sum_q amps[q] * perlin_q (generator/artifacts/utils.py:224-384) on an (n0,n1,n2) grid and
 * its min / max (ordered keys, mm[0], mm[1]; caller initialises with fsg_minmax_init(mm,1,1)).  Per octave q:
 * grads[q] DEVICE (r0+1,r1+1,r2+1,3) unit gradients (tileable wrap already applied), lins[q] DEVICE
 * linspace(0,r_a,n_a) for the three axes back to back, res[3q..3q+2] = (r0,r1,r2).  grads/lins/res/amps: HOST arrays. */
int fsg_perlin_fractal_f32(const float* const* grads, const float* const* lins, const int32_t* res, const float* amps,
                           int noct, int n0, int n1, int n2, float* out, int32_t* mm, void* stream);
/* out = (1 - w) a + w b, the spatially weighted merge of simulate_reco.py:704, augmentation/artifacts.py:125, :337.
 *   w_mode 0: w is the weight volume; 1: w is raw Perlin noise, normalised on the fly as
 *             clamp((w + increase - min) / (max - min), 0, 1) with w_mm from fsg_perlin_fractal_f32 (utils.py:386-387);
 *   seg != NULL: w *= (seg > 0)                                              (artifacts.py:336-337);
 *   b_mode 1: b is the multi-scale noise field, b' = clamp(a + std * b / max|b|, 0, 2 max a) with b_mm / a_mm the
 *             min/max keys of b and a                                        (artifacts.py:322-327).
 * out may be NULL (then only w_out, the weight actually used, is written); w_out may be NULL. */
int fsg_blend_f32(const float* a, const float* b, const float* w, size_t n, int w_mode, const int32_t* w_mm, float increase,
                  const float* seg, int b_mode, const int32_t* b_mm, const int32_t* a_mm, float std, float* out,
                  float* w_out, void* stream);
/* Scanner.add_noise (simulate_reco.py:236-256), in place: s = sqrt((s + sigma z1)^2 + (sigma z2)^2) where s > threshold.
 * noise1/noise2 dense fields (host-tape mode) or both NULL: Philox(seed, stream_id), two normals per pixel. */
int fsg_slice_noise_f32(float* slices, size_t n, float threshold, float sigma, const float* noise1, const float* noise2,
                        uint64_t seed, uint64_t stream_id, void* stream);
/* Scanner.signal_void (simulate_reco.py:258-298), in place on the nvoid slices slice_ids[t]:
 * s *= 1 - A exp(sx x'^2 + sy y'^2); params (nvoid,7) = {yc, xc, cos, sin, A, sx, sy}; ylin (h), xlin (w). */
int fsg_slice_void_f32(float* slices, int h, int w, const int32_t* slice_ids, const float* params, int nvoid,
                       const float* ylin, const float* xlin, void* stream);
/* sums[i] = sum of slice i (simulate_reco.py:409), deterministic. */
int fsg_slice_sums_f32(const float* slices, int n, size_t hw, float* sums, void* stream);

/* The reference picks random voxels of a mask as `torch.where(mask)` + `torch.randperm(count)[:k]`
 * (simulate_reco.py:661-664, augmentation/artifacts.py:110-113, :199-202, :565-567), materialising three int64 index
 * lists of the whole mask.  Here: per-bucket counts (buckets of FSG_NZ_BUCKET = 4096 voxels, raster order), a host
 * prefix sum, then the flat index of the rank[q]-th voxel of bucket[q] satisfying the predicate for each request
 * (-1 if the bucket holds fewer).  Predicate on (float)v: mode 0: v > value, 1: v == value, 2: v != value. */
#define FSG_NZ_BUCKET 4096
int fsg_nonzero_count_f32(const float* v, size_t n, int mode, float value, int32_t* counts, void* stream);
int fsg_nonzero_count_u8(const uint8_t* v, size_t n, int mode, float value, int32_t* counts, void* stream);
int fsg_nonzero_select_f32(const float* v, size_t n, int mode, float value, const int32_t* bucket, const int32_t* rank,
                           int nreq, long long* out, void* stream);
int fsg_nonzero_select_u8(const uint8_t* v, size_t n, int mode, float value, const int32_t* bucket, const int32_t* rank,
                          int nreq, long long* out, void* stream);

/* out[offsets[b] + r] = values[e] for the r-th voxel e of bucket b with `pred[e]` satisfying the predicate: the
 * boolean-mask gather `values[mask]` (augmentation/artifacts.py:78-80) in raster order; offsets = exclusive prefix sum
 * of fsg_nonzero_count_f32's counts (DEVICE, int64). */
int fsg_compact_f32(const float* values, const float* pred, size_t n, int mode, float value, const long long* offsets,
                    float* out, void* stream);
/* Element-wise glue of the artifact stages.  op 0: a + b; 1: (a > value) as 0/1; 2: (a == value) as 0/1; 3: a * b;
 * 4: a * (b > value); 5: max(a, b); 6: ((a - b) > value) as 0/1; 7: (a <= value) as 0/1.  b may be NULL for ops 1, 2, 7. */
int fsg_ewise_f32(const float* a, const float* b, size_t n, int op, float value, float* out, void* stream);

/* One axis pass of a capped distance transform: dst[v] = min over |t| <= radius (inside the volume) of
 * src'[v + t e_axis] + cost(t), cost = t^2 (metric 0) or |t| (metric 1); first != 0: src is a mask (set -> 0, else 1e9).
 * Passes over axes 0,1,2 give the squared Euclidean / city-block distance to the mask where it is <= radius^2 / radius:
 * `dist <= radius^2` is the reference's zero-padded conv3d with skimage's ball(radius) > 0 (augmentation/artifacts.py:484-499)
 * without its (2r+1)^3 taps; `city-block dist <= k` is k successive ball(1) dilations (artifacts.py:587-589).  src != dst. */
int fsg_dist_pass_f32(const float* src, float* dst, int n0, int n1, int n2, int axis, int radius, int metric, int first,
                      void* stream);
/* SimulatedBoundaries' fuzzy mask (augmentation/artifacts.py:565-604) fused: k = clamp(rint(p n_dilate - 1), 0) with
 * p = mog where mask_modif added voxels to mask (0 elsewhere); m = mask_modif * (dist <= max(k-1, 0)), dist = city-block
 * distance to mask; out = image * m (out/image may be NULL), mask_out = m (may be NULL). */
int fsg_boundary_mask_f32(const float* image, const float* mask, const float* mask_modif, const float* mog, const float* dist,
                          int n_dilate, size_t n, float* out, float* mask_out, void* stream);
/* out = a * (u < p), u ~ U[0,1) from Philox(seed, stream_id) per element: device-RNG thinning of a voxel set. */
int fsg_bernoulli_keep_f32(const float* a, size_t n, float p, uint64_t seed, uint64_t stream_id, float* out, void* stream);
/* out[idx[q]] = value for q < k (idx DEVICE int64; entries outside [0, n) are skipped). */
int fsg_scatter_const_f32(float* out, size_t n, const long long* idx, int k, float value, void* stream);

/* ---- whole-sample launch sequence -------------------------------------------------------------------------- */
/* One call = the fused kernel sequence of FetalSynthGen.sample (generator/model.py:231-276) for the
 * seeds-based path: GMM draw -> [rows, margins, warp(+gamma+bias, labels)] -> [blur x,y,z, resample+noise,
 * zoom-back min/max, zoom-back+normalise] on `stream`.  All random quantities are inputs (drawn by the host in
 * the reference's order).  Workspaces are caller-owned and may be reused by the next call on the same stream.
 * Returns FSG_E_ALIGN / FSG_E_TOOBIG for configurations the fused kernels do not cover (the caller then issues
 * the entry points above one by one). */
/* Head of a sample as ONE launch: the GMM draw (as fsg_gmm_sample_u8x4), the per-row coarse values (as
 * fsg_deform_rows_f32) and the six-face pass of fsg_coords_floormin_f32 share no data, so their workgroups run side by side.
 * mm3 must already hold initialised keys (fsg_minmax_init, or key values uploaded with the parameters): nothing in this
 * launch resets them.  fsg_coords_floormin_rest_f32 is the conditional full pass that completes the floor(min) keys. */
int fsg_sample_head_f32(const uint8_t* l0, const uint8_t* l1, const uint8_t* l2, const uint8_t* l3, size_t n,
                        const float* mus, const float* sigmas, int ntab, const float* noise, uint64_t seed,
                        uint64_t stream_id, float* out, const fsg_deform* d, const fsg_epilogue* epi, float* rows,
                        int row_stride, int32_t* mm3, void* stream);
int fsg_coords_floormin_rest_f32(const fsg_deform* d, int32_t* mm3, void* stream);
/* The same launch with the seed labels of a subject as ONE uint16 code volume instead of four uint8 volumes (6 instead of 8
 * bytes per voxel): codes[v] indexes `tuples`, ntuples rows of `stride` bytes, row c = the value every seed volume of the
 * subject holds at the voxels with code c (built once per subject by the caller: the distinct columns of the stacked seed
 * volumes); sel[m] = the byte of a row that belongs to the volume selected for meta label m (a zero byte for an absent one).
 * label = (tuples[c][sel[0]] + .. + tuples[c][sel[3]]) & 255 -- what fsg_sample_head_f32 adds up voxel by voxel
 * (rand_gmm.py:91-99) -- so the output is bit-identical.  n % 8 == 0, n <= 2^30, ntuples <= FSG_CODES_MAX, Philox noise only;
 * otherwise FSG_E_ALIGN / FSG_E_TOOBIG and the caller uses fsg_sample_head_f32.  (One-byte codes for subjects with <= 256
 * columns were measured and dropped: the kernel follows the width of a wave's read request, not its bytes -- 35.1 / 33.0 / 38.3 us
 * with 4 / 8 / 16-byte loads of uint8 codes against 31.6 / 30.8 with 8 / 16-byte loads of uint16 codes.) */
#define FSG_CODES_MAX 2048
int fsg_sample_head_codes_f32(const uint16_t* codes, const uint8_t* tuples, int ntuples, int stride, const int32_t sel[4],
                              size_t n, const float* mus, const float* sigmas, int ntab, uint64_t seed, uint64_t stream_id,
                              float* out, const fsg_deform* d, const fsg_epilogue* epi, float* rows, int row_stride,
                              int32_t* mm3, void* stream);

/* Builds codes (n uint16) and tuples (cap rows of `stride` bytes; row c, byte j = the value of parts[j] at the voxels with code
 * c; bytes >= nparts stay 0) from the nparts <= 64 uint8 volumes of a subject in one pass (csrc/fsg_codes.hip).  parts: HOST
 * array of device pointers.  work: fsg_seed_codes_work_bytes() of device scratch (16-byte aligned).  *count_dev receives the
 * number of distinct columns: usable only if it is <= cap (<= FSG_CODES_MAX) -- read it after synchronising.  The numbering of
 * the codes depends on the order in which the lanes met the columns (run to run); codes and tuples are consistent with each
 * other, which is all fsg_sample_head_codes_f32 needs.  No reference counterpart: the reference re-reads and adds up the four
 * selected NIfTI volumes per sample (rand_gmm.py:91-99). */
size_t fsg_seed_codes_work_bytes(void);
int fsg_seed_codes_build(const uint8_t* const* parts, int nparts, size_t n, int stride, uint16_t* codes, uint8_t* tuples, int cap,
                         void* work, size_t work_bytes, int32_t* count_dev, void* stream);

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
  int32_t mm8_preset;            /* 1: the keys already hold [+inf x4 | -inf x4] (uploaded with the parameters): no reset
                                  *    launch-side, and the head of the sample runs as one launch (fsg_sample_head_f32)  */
  float* out;                    /* shape[] floats                                                            */
  void* ev_blur_begin;           /* optional hipEvent_t pair recorded around the blur passes (fsg_event_*)     */
  void* ev_blur_end;
  int32_t* mm_slots;             /* optional: mm_nslots initialised slots (see fsg_zoom3d_minmax_sharded_f32) for the min / max
                                    of the zoom-back; the [0,1] scaling then reads them instead of mm8[3..4] */
  int32_t mm_nslots;
  const uint8_t* seg_in_u8;      /* optional uint8 copy of seg_in (same values): the label gather then reads 1 B/voxel
                                    (fsg_warp_f32_u8_to_f32); seg_in stays required as the fallback source */
  /* parameter upload as part of the call (optional): arena_bytes (multiple of 16) from the device-visible pinned block
   * arena_host to arena_dev -- the block every pointer above with small per-sample content points into -- before the first
   * kernel.  The caller must not reuse arena_host before work enqueued on `stream` AFTER this call has completed. */
  const void* arena_host;
  void* arena_dev;
  uint64_t arena_bytes;
  /* overlap != 0: the upload and the head of the sample (K1 + per-row coarse values + face minima) run on a side stream
   * owned by the library, ordered behind the point of the PREVIOUS call on `stream` after which ws0 / ws_rows are no longer
   * touched (its blur), so that they execute beside the previous sample's resampling tail; `stream` waits for them before
   * the warp.  ws_seq identifies consecutive uses of the workspace: the previous call's ordering point is honoured only if
   * it carried ws_seq - 1 and the same ws0, otherwise the side stream is ordered behind everything enqueued on `stream`
   * so far.  Requires arena_dev to be private to this call and not reused by the next one (a ring of >= 2 blocks).
   * Results are those of overlap == 0. */
  int32_t overlap;
  uint64_t ws_seq;
  /* optional: the deformed labels as uint8 (needs seg_in_u8; seg_out may then be NULL): the device-resident streaming hand-over
   * (reference data/datasets.py:315-323 converts the labels after the fact) without a conversion pass */
  uint8_t* seg_out_u8;
  /* optional stage trace (measurement only): trace_events = trace_cap hipEvent_t handles (fsg_event_create), trace_ids =
   * trace_cap + 1 ints.  An event is recorded before the first launch (id FSG_ST_BEGIN) and after every launch of the
   * sample (id = the FSG_ST_* of that launch); trace_ids[trace_cap] receives the number of events recorded.  The time
   * between two consecutive events is the later one's launch plus one barrier packet (each record is one). */
  void** trace_events;
  int32_t* trace_ids;
  int32_t trace_cap;
  int32_t trace_start;           /* events the caller has already recorded (fsg_keyed_sample_run: 1, the one before its draw
                                  * kernel); the first event of this call then carries trace_first_id instead of FSG_ST_BEGIN */
  int32_t trace_first_id;
  /* optional (ABI 3): the subject's seed labels as one code volume (fsg_sample_head_codes_f32); label_parts stay valid and are
   * what every path without the one-launch head reads */
  const uint16_t* label_codes;
  const uint8_t* code_tuples;
  int32_t code_ntuples, code_stride;
  int32_t code_sel[4];
  /* Look-ahead (ABI 3; set by fsg_keyed_sample_run when the caller names the next sample, zero otherwise): the keyed draw job
   * of the NEXT sample -- it depends on nothing of this one -- goes out beside this sample's conditional floor(min) pass
   * instead of as a launch of its own.  ride_draw: host pointer to a library-internal argument block, valid for the call.
   * *rode (may be NULL) receives 1 when the job really went out. */
  const void* ride_draw;
  uint32_t ride_draw_blocks;
  int32_t* rode;
} fsg_sample_plan;
enum {
  FSG_ST_BEGIN = 0, FSG_ST_UPLOAD = 1, FSG_ST_DRAW = 2, FSG_ST_HEAD = 3, FSG_ST_FLOORMIN = 4, FSG_ST_WARP = 5, FSG_ST_BLUR_X = 6,
  FSG_ST_BLUR_Y = 7, FSG_ST_BLUR_Z = 8, FSG_ST_BLUR_YZ = 9, FSG_ST_K7 = 10, FSG_ST_K9A = 11, FSG_ST_K9B = 12, FSG_ST_GMM = 13,
  FSG_ST_ROWS = 14, FSG_ST_POINTWISE = 15, FSG_ST_BLUR_RS_X = 16, FSG_ST_BLUR_RS_YZ = 17, FSG_ST_COUNT = 18
};
int fsg_sample_run(const fsg_sample_plan* plan_host, void* stream);
/* Layout check for FFI mirrors of the struct: which = 0 -> sizeof(fsg_sample_plan); 1 / 2 / 3 / 4 -> offsetof blur_taps / out /
 * seg_in_u8 / ws_seq; anything else -> -1.  Callable without a GPU. */
int64_t fsg_sample_plan_layout(int which);
/* B samples with one call: plan b runs on streams[b % nstreams] (hipStream_t handles).  The caller orders those streams
 * behind the upload of every plan's parameters and waits for them afterwards; per sample the work is exactly
 * fsg_sample_run's (reference: B consecutive FetalSynthGen.sample calls, generator/model.py:231-276; the reference's
 * DataLoader collates B such samples, data/datasets.py:310-325, docs/datasets.md:4-6). */
/* The plan from two flat arrays (one FFI call instead of a field-by-field fill): iv[FSG_PLAN_I_*] integers and pointers as
 * int64, fv[FSG_PLAN_F_*] doubles that hold float32 values exactly, taps = 3 x FSG_PLAN_TAPS_STRIDE floats (row a = the
 * blur taps of axis a, FSG_PLAN_I_BLUR_NTAPS + a of them).  fsg_sample_pack_run = pack + fsg_sample_run. */
enum {
  FSG_PLAN_I_SHAPE = 0, FSG_PLAN_I_LABEL_PARTS = 3, FSG_PLAN_I_MUS = 7, FSG_PLAN_I_SIGMAS = 8, FSG_PLAN_I_NTAB = 9,
  FSG_PLAN_I_GMM_NOISE = 10, FSG_PLAN_I_GMM_SEED = 11, FSG_PLAN_I_GMM_STREAM = 12, FSG_PLAN_I_DEFORM_ACTIVE = 13,
  FSG_PLAN_I_FLIP = 14, FSG_PLAN_I_FIELD_DIMS = 15, FSG_PLAN_I_FIELD = 18, FSG_PLAN_I_FIELD_TABS = 19, FSG_PLAN_I_SEG_IN = 22,
  FSG_PLAN_I_SEG_OUT = 23, FSG_PLAN_I_SEG_IN_U8 = 24, FSG_PLAN_I_BIAS_DIMS = 25, FSG_PLAN_I_BIAS = 28, FSG_PLAN_I_BIAS_TABS = 29,
  FSG_PLAN_I_RESAMPLE_ACTIVE = 32, FSG_PLAN_I_LOW_SHAPE = 33, FSG_PLAN_I_RS_TABS = 36, FSG_PLAN_I_BACK_TABS = 39,
  FSG_PLAN_I_BLUR_NTAPS = 42, FSG_PLAN_I_NOISE_MODE = 45, FSG_PLAN_I_NOISE = 46, FSG_PLAN_I_NOISE_SEED = 47,
  FSG_PLAN_I_NOISE_STREAM = 48, FSG_PLAN_I_SCALE01 = 49, FSG_PLAN_I_WS0 = 50, FSG_PLAN_I_WS1 = 51, FSG_PLAN_I_WS_LOW = 52,
  FSG_PLAN_I_WS_ROWS = 53, FSG_PLAN_I_ROW_STRIDE = 54, FSG_PLAN_I_MM8 = 55, FSG_PLAN_I_MM8_PRESET = 56, FSG_PLAN_I_OUT = 57,
  FSG_PLAN_I_EV_BEGIN = 58, FSG_PLAN_I_EV_END = 59, FSG_PLAN_I_MM_SLOTS = 60, FSG_PLAN_I_MM_NSLOTS = 61, FSG_PLAN_I_ARENA_HOST = 62,
  FSG_PLAN_I_ARENA_DEV = 63, FSG_PLAN_I_ARENA_BYTES = 64, FSG_PLAN_I_OVERLAP = 65, FSG_PLAN_I_WS_SEQ = 66, FSG_PLAN_I_SEG_OUT_U8 = 67,
  FSG_PLAN_I_TRACE_EVENTS = 68, FSG_PLAN_I_TRACE_IDS = 69, FSG_PLAN_I_TRACE_CAP = 70, FSG_PLAN_I_COUNT = 71
};
enum { FSG_PLAN_F_A = 0, FSG_PLAN_F_CENTRE = 9, FSG_PLAN_F_C2 = 12, FSG_PLAN_F_GAMMA = 15, FSG_PLAN_F_NOISE_STD = 16, FSG_PLAN_F_COUNT = 17 };
#define FSG_PLAN_TAPS_STRIDE 132
int fsg_sample_plan_pack(fsg_sample_plan* plan, const int64_t* iv, int niv, const double* fv, int nfv, const float* taps);
int fsg_sample_pack_run(const int64_t* iv, int niv, const double* fv, int nfv, const float* taps, void* stream);
int fsg_sample_run_batch(const fsg_sample_plan* plans, int nplans, void* const* streams, int nstreams);

/* ---- K6 + K7 (+ K8) fused per axis (csrc/fsg_blur_rs.hip) -----------------------------------------------------------
 * RandResample.__call__ (generator/augmentation/synthseg.py:50-107) = gaussian_blur_3d (utils/generation.py:84-110) then the
 * axis-aligned trilinear down-sampling (synthseg.py:84-104 -> utils/generation.py:227-278), RandNoise after it (synthseg.py:
 * 230-233).  Blur and resampling are separable linear operators, so per axis they combine into one FIR with position-dependent
 * coefficients (w_lo k[.] + w_hi k[. - 1]) evaluated for the m low-res outputs only; the blurred full-resolution volume is
 * never formed.  Two launches: x (reads N, writes N m0/n0), then y + z + noise (reads N m0/n0, writes M).  Results equal
 * the unfused sequence (fsg_blur_axis_* x3, fsg_resample_noise_f32) up to float32 rounding of a re-ordered linear map --
 * within the blur's own tolerance (atol 1e-3 on the 0..255 scale).  Zero-padded un-renormalised borders, outputs whose
 * position is outside (0, n-1] are 0 (lo < 0 in the table), noise and the clamp at 0 after, as in the reference.
 * taps: odd counts 3..17 (radius 1..8) on all three axes; n2 % 4 == 0, n2 <= 512; otherwise FSG_E_ALIGN (callers then use
 * the unfused entry points).  fsg_blur_resample_supported: 1 when both launches accept the configuration (no GPU call). */
int fsg_blur_resample_supported(int n0, int n1, int n2, int m0, int m1, int m2, int ntaps_x, int ntaps_y, int ntaps_z);
int fsg_blur_resample_x_f32(const float* src, int n0, int n1, int n2, const fsg_tap* tx, int m0, const float* taps_host,
                            int ntaps, float* dst, void* stream);
int fsg_blur_resample_yz_noise_f32(const float* src, int m0, int n1, int n2, const fsg_tap* ty, const fsg_tap* tz, int m1,
                                   int m2, const float* taps_y_host, int ntaps_y, const float* taps_z_host, int ntaps_z,
                                   int noise_mode, const float* noise, uint64_t seed, uint64_t stream_id, float noise_std,
                                   float* dst, void* stream);

/* ---- keyed mode: every per-sample draw from a counter-based generator keyed (base_seed, sample index) --------------------
 * The reference draws a sample's ~30 scalars and its small tensors from numpy's / torch's GLOBAL generators, one Python call
 * each, in an order that is part of its behaviour (rand_gmm.py:82-85,:120-148, affine_nonrigid.py:140-145,
 * :248-263,:284,:303-318, synthseg.py:263-265,:157-176,:63-78,:218-232).  `rng="reference"` / `"device"` replay that tape on
 * the host (~100 interpreter round trips per sample).  Keyed mode keeps the DISTRIBUTIONS and the arithmetic that turns draws
 * into parameters, but takes every uniform / normal from Philox4x32-10 under the sample's 64-bit key (a fixed counter per
 * draw, so gates do not shift later draws): the scalars are drawn in C inside fsg_keyed_sample_run, the small tensors (GMM
 * tables, coarse displacement grid, bias grid) by a one-launch device kernel straight into the sample's parameter block, the
 * two large fields in the kernels that consume them (as in "device" mode).  The host hands over pointers and the key.
 * A sample depends on its key only -- not on the process, the GPU count or what ran before ("(base_seed,
 * sample_index) keyed RNG").  Parity: fsg_keyed_draws exports what was drawn; tests feed it to the oracle.
 */
typedef struct fsg_keyed_config {
  int32_t shape[3];
  int32_t size[3];                 /* SpatialDeformation(size=...)                                   */
  double resolution[3];
  int32_t min_subclusters, max_subclusters, meta_labels;   /* ImageFromSeeds; meta_labels <= 4       */
  int32_t nlabels;                 /* max(seed_labels) + 1 <= 256                                    */
  int32_t n_seed_labels;           /* len(seed_labels) == len(generation_classes) <= 256             */
  int32_t tie_classes;             /* generation_classes != seed_labels (rand_gmm.py:139-145)        */
  uint8_t seed_labels[256];
  uint8_t generation_classes[256];
  double deform_prob, flip_prb, max_rotation, max_shear, max_scaling;
  int32_t nonlinear;
  double nonlin_scale_min, nonlin_scale_max, nonlin_std_max;
  double gamma_prob, gamma_std;
  double bias_prob, bf_scale_min, bf_scale_max, bf_std_min, bf_std_max;
  double resample_prob, min_resolution, max_resolution;
  double noise_prob, noise_std_min, noise_std_max;
} fsg_keyed_config;

typedef struct fsg_keyed_draws {
  uint64_t key;
  int32_t subclusters[4];          /* mlabel m+1 -> number of sub-clusters                           */
  int32_t ntab;
  int32_t deform_active, flip;
  double rotations[3], shears[3], scalings[3];
  float A[9];
  double c2[3];
  int32_t nonlinear;
  double nonlin_scale, nonlin_std;
  int32_t field_dims[3];
  int32_t gamma_active;
  double gamma;
  int32_t bias_active;
  double bf_scale, bf_std;
  int32_t bias_dims[3];
  int32_t resample_active;
  double spacing, u_std, stds[3];
  int32_t low_shape[3];
  int32_t blur_ntaps[3];
  int32_t noise_active;
  double noise_std;
  float noise_std32;
  /* byte offsets into the sample's device parameter block of what the draw kernel writes there */
  int32_t off_mm8, off_slots, off_mus, off_sigmas, off_bias, off_field, block_bytes;
  int32_t rode; /* fsg_keyed_sample_run: 1 when the draw job of the NEXT sample went out with this one (look-ahead) */
} fsg_keyed_draws;

enum { FSG_KT_RESAMPLE = 0, FSG_KT_BACK = 1, FSG_KT_FIELD = 2, FSG_KT_BIAS = 3 };
/* Host-only object (no HIP call, usable without a GPU).  *ctx receives the handle. */
int fsg_keyed_create(const fsg_keyed_config* cfg, void** ctx);
int fsg_keyed_destroy(void* ctx);
/* Registers the DEVICE tap table of (kind, axis, n): FSG_KT_RESAMPLE / FSG_KT_BACK: n = low-res size along `axis`;
 * FSG_KT_FIELD / FSG_KT_BIAS: n = coarse grid size along `axis`.  The tables are the ones the other modes use (built by the
 * host mirror with the reference's own torch calls, utils/generation.py:315-363, synthseg.py:84-102), so sampling positions
 * stay bit-identical.  A sample that needs an unregistered table fails with FSG_E_NOTABLE. */
int fsg_keyed_set_table(void* ctx, int kind, int axis, int n, const fsg_tap* table_dev);
/* Bytes of the per-sample device parameter block for the largest grids this configuration can draw. */
int64_t fsg_keyed_block_bytes(void* ctx);
/* Host draws of the sample `key` (no device work): what fsg_keyed_sample_run will use. */
int fsg_keyed_draw(void* ctx, uint64_t key, fsg_keyed_draws* out);
/* One sample: draws + the draw kernel + fsg_sample_run's launch sequence.  iv (int64): FSG_KEYED_I_*.
 * bank: FSG_KEYED_I_BANK + 4 * (n_sub - min_subclusters) + (mlabel - 1) -> uint8 seed volume of (n_sub, mlabel).
 * draws_out may be NULL. */
enum {
  FSG_KEYED_I_KEY = 0, FSG_KEYED_I_OUT = 1, FSG_KEYED_I_SEG_OUT = 2, FSG_KEYED_I_SEG_OUT_U8 = 3, FSG_KEYED_I_SEG_IN = 4,
  FSG_KEYED_I_SEG_IN_U8 = 5, FSG_KEYED_I_BLOCK = 6, FSG_KEYED_I_WS0 = 7, FSG_KEYED_I_WS1 = 8, FSG_KEYED_I_WS_LOW = 9,
  FSG_KEYED_I_WS_ROWS = 10, FSG_KEYED_I_ROW_STRIDE = 11, FSG_KEYED_I_SCALE01 = 12, FSG_KEYED_I_TRACE_EVENTS = 13,
  FSG_KEYED_I_TRACE_IDS = 14, FSG_KEYED_I_TRACE_CAP = 15, FSG_KEYED_I_BANK = 16, FSG_KEYED_I_EV_BLUR_BEGIN = 16 + 64,
  FSG_KEYED_I_EV_BLUR_END = 16 + 65,
  /* optional code volume of the subject (0 = none): uint16 codes, uint8 tuples [ntuples][stride] whose byte
   * 4 * (n_sub - min_subclusters) + (mlabel - 1) is the value of seed volume (n_sub, mlabel) and whose byte stride - 1 is 0 */
  FSG_KEYED_I_CODES = 16 + 66, FSG_KEYED_I_CODE_TUPLES = 16 + 67, FSG_KEYED_I_CODE_NTUPLES = 16 + 68, FSG_KEYED_I_CODE_STRIDE = 16 + 69,
  /* look-ahead (optional).  FLAGS bit 0: the block of THIS sample is already filled (the previous call carried its draw job);
   * bit 2: NEXT_KEY / NEXT_BLOCK name the sample the caller will run next on this stream -- its draw job then rides in this
   * sample's floor(min) launch (fsg_keyed_draws::rode says whether it did: not when the deformation gate is off).  A caller that
   * then runs something else simply does not set bit 0.  (r03: the next sample's GMM draw beside the zoom-back launches was built
   * and measured too -- 224 -> 230-244 us per step, the two jobs slow each other down -- and removed.) */
  FSG_KEYED_I_FLAGS = 16 + 70, FSG_KEYED_I_NEXT_KEY = 16 + 71, FSG_KEYED_I_NEXT_BLOCK = 16 + 72,
  FSG_KEYED_I_COUNT = 16 + 73
};
int fsg_keyed_sample_run(void* ctx, const int64_t* iv, int niv, fsg_keyed_draws* draws_out, void* stream);
/* The draw kernel alone (tests): fills the parameter block of `draws` at block_dev. */
int fsg_keyed_fill_block(void* ctx, const fsg_keyed_draws* draws, void* block_dev, void* stream);

/* Releases the library-owned side streams / events of the opt-in head overlap (fsg_sample_plan::overlap); they otherwise
 * live for the life of the process.  Synchronise the launch streams first.  No reference counterpart (the reference owns
 * no streams). */
int fsg_pipeline_teardown(void);
/* float32 -> float16 (round to nearest even) copy of a volume: the optional half-precision image of the output side. */
int fsg_cast_f32_to_f16(const float* x, size_t n, void* out_f16, void* stream);

/* hipEvent helpers for ctypes callers (timing on the launch stream). elapsed_ms synchronises on `end`. */
void* fsg_event_create(void);
int fsg_event_record(void* event, void* stream);
int fsg_event_destroy(void* event);
int fsg_event_elapsed_ms(void* begin, void* end, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* FSG_HIP_H */
