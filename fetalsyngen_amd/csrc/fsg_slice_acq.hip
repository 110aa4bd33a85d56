// fsg_slice_acq.hip -- slice acquisition (volume -> motion-corrupted slice stacks) and its adjoint
// (slices -> volume), the two operators behind the reference's SR-artifact simulation
// (generator/artifacts/simulate_reco.py: Scanner.scan :386-407, PSFreconstruction :38-54).
//
// Replaces the reference's own CUDA extension (generator/artifacts/svort/slice_acquisition/
// slice_acq_cuda_kernel.cu: forward :17-171, adjoint forward :472-670, equalize :672-693) and, for
// seed-matched comparison with the reference's CPU run, its torch fallback (slice_acq.py:266-546),
// whose arithmetic is different (see include/fsg_hip.h, FSG_SA_*).
//
// MI355X shape of the work: one 256-thread workgroup = a 16x16 pixel tile of ONE slice, so the slice's
// rigid transform is wave-uniform (scalar registers) and the per-tap products r_ij * tap_offset are formed
// once per workgroup into LDS; every lane then walks the PSF raster, and the footprints of neighbouring
// pixels overlap in the volume, so the 2x2x2 gathers of a wave mostly hit lines its neighbours just
// touched (the 216 MiB of a 384^3 volume sit in the 256 MiB Infinity Cache).  The PSF itself lives in LDS
// (uniform reads are broadcasts; the interp_psf branch gathers its 2x2x2 neighbourhood from there).
#include "fsg_common.h"

int g_sa_cap = 3072;        // LDS accumulator cells (value + weight)
int g_sa_zc = 3;            // PSF planes per accumulation chunk
int g_sa_t16_extent = 20;   // use 16x16 tiles while pitch*15 + PSF width <= this
int g_sa_auto = 1;          // derive planes-per-chunk (and a larger capacity if needed) from pitch and PSF size

namespace {

struct SaParams {
  const float* tr;          // (n, 3, 4)
  const float* psf;         // (pd, ph, pw)
  const uint8_t* vmask;     // (D,H,W) or null
  const uint8_t* smask;     // (n,h,w) or null
  const int32_t* sid;       // adjoint only: slice z of the launch reads slices[sid[z]] (null: identity)
  int D, H, W;
  int pd, ph, pw;
  int n, h, w;
  float res;
  // forward, linear PSF: which slices a launch takes (per workgroup, from the slice's own transform): 0 all, 1 those with
  // sa_direct_cost(T) <= fwd_thr (direct gathers), 2 the others (plate kernel)
  int fwd_select;
  float fwd_thr;
  float fwd_wy;  // weight of the slice x axis' y component in sa_direct_cost (10 for PSFs of 400 elements and more, 30 below)
};

constexpr int SA_TILE = 16;
// Direct gathers cost 4.2 + ~10 |T10| + ~35 |T20| ms for the 80 x 320^2 x 441-tap stack of the bench (lanes run along the
// slice's x axis: its y component spreads a wave's gather over rows, its z component over planes 590 KB apart), the plate
// kernel 5.7 ms aligned, 6-9.3 ms for every other orientation
// (profiles/r03_i_slice_acq_forward.txt): direct gathers only win for nearly aligned slices.  For PSFs under 400 elements the
// plate's set-up weighs more and a z tilt is tolerated longer (crossover 35 |T20| ~ 10 at a pixel pitch of one voxel), but a
// y tilt is not (crossover |T10| ~ 0.35: the few taps no longer share the rows a wave's gather touches), and both scale with
// the pixel pitch -- section 8 of the same file, stacks as SimulateMotion draws them.
__device__ __forceinline__ float sa_direct_cost(const float* T, float wy) { return wy * fabsf(T[4]) + 35.f * fabsf(T[8]); }
constexpr int SA_MAX_PSF = 4096;  // elements (16 KB of LDS)
constexpr int SA_MAX_AXIS = 64;

// LDS image of the PSF and of the per-axis products of one slice's rotation with the tap offsets.
struct SaLds {
  float* psf;   // [pd*ph*pw]
  float* tx;    // [3][pw]   r_{a,0} * ix_p
  float* ty;    // [3][ph]   r_{a,1} * iy_p
  float* tz;    // [3][pd]   r_{a,2} * iz_p
};

__device__ __forceinline__ SaLds sa_lds_layout(float* base, const SaParams& P) {
  SaLds L;
  L.psf = base;
  L.tx = L.psf + P.pd * P.ph * P.pw;
  L.ty = L.tx + 3 * P.pw;
  L.tz = L.ty + 3 * P.ph;
  return L;
}

__device__ __forceinline__ void sa_fill_lds(const SaLds& L, const SaParams& P, const float* __restrict__ T) {
  const int tid = threadIdx.y * SA_TILE + threadIdx.x;
  const int np = P.pd * P.ph * P.pw;
  for (int e = tid; e < np; e += SA_TILE * SA_TILE) L.psf[e] = P.psf[e];
  // float * int in the reference == float * (float)int
  for (int e = tid; e < 3 * P.pw; e += SA_TILE * SA_TILE) {
    const int a = e / P.pw, i = e - a * P.pw;
    L.tx[e] = T[a * 4 + 0] * (float)(i - P.pw / 2);
  }
  for (int e = tid; e < 3 * P.ph; e += SA_TILE * SA_TILE) {
    const int a = e / P.ph, i = e - a * P.ph;
    L.ty[e] = T[a * 4 + 1] * (float)(i - P.ph / 2);
  }
  for (int e = tid; e < 3 * P.pd; e += SA_TILE * SA_TILE) {
    const int a = e / P.pd, i = e - a * P.pd;
    L.tz[e] = T[a * 4 + 2] * (float)(i - P.pd / 2);
  }
}

// pixel centre in voxel coordinates (slice_acq_cuda_kernel.cu:46-56): the pixel offset is formed in double
// (the literal `2.` promotes), the rotation in fp32.
__device__ __forceinline__ void sa_centre(const SaParams& P, const float* __restrict__ T, int ix, int iy, float& xc,
                                          float& yc, float& zc) {
  const float _x = (float)(((double)ix - (P.w - 1) / 2.) * (double)P.res + (double)T[3]);
  const float _y = (float)(((double)iy - (P.h - 1) / 2.) * (double)P.res + (double)T[7]);
  const float _z = T[11];
  xc = T[0] * _x + T[1] * _y + T[2] * _z;
  yc = T[4] * _x + T[5] * _y + T[6] * _z;
  zc = T[8] * _x + T[9] * _y + T[10] * _z;
  xc = (float)((double)xc + (P.W - 1) / 2.);
  yc = (float)((double)yc + (P.H - 1) / 2.);
  zc = (float)((double)zc + (P.D - 1) / 2.);
}

// interp_psf branch (:79-101): PSF value at the snapped voxel, trilinear in the PSF grid.  false = outside.
__device__ __forceinline__ bool sa_psf_at(const SaLds& L, const SaParams& P, const float* __restrict__ T, float dx,
                                          float dy, float dz, float& val) {
  const float xp = (float)((double)(T[0] * dx + T[4] * dy + T[8] * dz) + (P.pw - 1) / 2.);
  const float yp = (float)((double)(T[1] * dx + T[5] * dy + T[9] * dz) + (P.ph - 1) / 2.);
  const float zp = (float)((double)(T[2] * dx + T[6] * dy + T[10] * dz) + (P.pd - 1) / 2.);
  if (xp < 0 || yp < 0 || zp < 0 || xp >= (float)(P.pw - 1) || yp >= (float)(P.ph - 1) || zp >= (float)(P.pd - 1))
    return false;
  const float xf = floorf(xp), yf = floorf(yp), zf = floorf(zp);
  const float wx = xp - xf, wy = yp - yf, wz = zp - zf;
  const int sy = P.pw, sz = P.pw * P.ph;
  const float* q = L.psf + (int)zf * sz + (int)yf * sy + (int)xf;
  float v = 0.f;
  v += (1 - wx) * (1 - wy) * (1 - wz) * q[0];
  v += wx * (1 - wy) * (1 - wz) * q[1];
  v += (1 - wx) * wy * (1 - wz) * q[sy];
  v += (1 - wx) * (1 - wy) * wz * q[sz];
  v += wx * wy * (1 - wz) * q[1 + sy];
  v += wx * (1 - wy) * wz * q[1 + sz];
  v += (1 - wx) * wy * wz * q[sy + sz];
  v += wx * wy * wz * q[sy + sz + 1];
  val = v;
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// forward, CUDA-kernel semantics.  NN = interp_psf.
// ---------------------------------------------------------------------------------------------------------
template <bool NN, bool VM>
__global__ __launch_bounds__(SA_TILE* SA_TILE) void sa_forward_kernel(SaParams P, const float* __restrict__ vol,
                                                                      float* __restrict__ slices,
                                                                      float* __restrict__ weights) {
  extern __shared__ float smem[];
  const int in = blockIdx.z;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  const SaLds L = sa_lds_layout(smem, P);
  sa_fill_lds(L, P, T);
  __syncthreads();
  const int ix = blockIdx.x * SA_TILE + threadIdx.x, iy = blockIdx.y * SA_TILE + threadIdx.y;
  if (ix >= P.w || iy >= P.h) return;
  const size_t idx = ((size_t)in * P.h + iy) * P.w + ix;
  if (P.smask && !P.smask[idx]) {
    slices[idx] = 0.f;
    if (weights) weights[idx] = 0.f;
    return;
  }
  float xc, yc, zc;
  sa_centre(P, T, ix, iy, xc, yc, zc);
  const int Sy = P.W, Sz = P.H * P.W;
  const float hx = (float)(P.W - 1), hy = (float)(P.H - 1), hz = (float)(P.D - 1);
  float val = 0.f, weight = 0.f;
  int ip = 0;
  for (int kz = 0; kz < P.pd; ++kz) {
    const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
    for (int ky = 0; ky < P.ph; ++ky) {
      const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
      for (int kx = 0; kx < P.pw; ++kx, ++ip) {
        float pv = L.psf[ip];
        if (pv == 0.f) continue;  // wave-uniform
        const float x = xc + L.tx[kx] + yx + zx;
        const float y = yc + L.tx[P.pw + kx] + yy + zy;
        const float z = zc + L.tx[2 * P.pw + kx] + yz + zz;
        if (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz) continue;
        if (NN) {
          const float xr = roundf(x), yr = roundf(y), zr = roundf(z);
          const int iv = (int)zr * Sz + (int)yr * Sy + (int)xr;
          if (VM && !P.vmask[iv]) continue;
          if (!sa_psf_at(L, P, T, xr - xc, yr - yc, zr - zc, pv)) continue;
          val += pv * vol[iv];
          weight += pv;
        } else {
          const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
          const float wx = x - xf, wy = y - yf, wz = z - zf;
          const int iv = (int)zf * Sz + (int)yf * Sy + (int)xf;
          const float* __restrict__ q = vol + iv;
          float pw_;
#define SA_CORNER(OFF, WEXPR)                     \
  if (!VM || P.vmask[iv + (OFF)]) {               \
    pw_ = (WEXPR) * pv;                           \
    val += pw_ * q[(OFF)];                        \
    weight += pw_;                                \
  }
          SA_CORNER(0, (1 - wx) * (1 - wy) * (1 - wz))
          SA_CORNER(1, wx * (1 - wy) * (1 - wz))
          SA_CORNER(Sy, (1 - wx) * wy * (1 - wz))
          SA_CORNER(Sz, (1 - wx) * (1 - wy) * wz)
          SA_CORNER(1 + Sy, wx * wy * (1 - wz))
          SA_CORNER(1 + Sz, wx * (1 - wy) * wz)
          SA_CORNER(Sy + Sz, (1 - wx) * wy * wz)
          SA_CORNER(Sy + Sz + 1, wx * wy * wz)
#undef SA_CORNER
        }
      }
    }
  }
  const bool good = weight > 0.f;
  slices[idx] = good ? val / weight : 0.f;
  if (weights) weights[idx] = good ? weight : 0.f;
}


// ---------------------------------------------------------------------------------------------------------
// forward, interp_psf = false, no volume mask: the acquisition of Scanner.scan (simulate_reco.py:386-396).
// Same sampling positions as sa_forward_kernel; the 2x2x2 blend is evaluated as lerps on x-pairs fetched with one
// 8-byte load each (4 loads and ~20 VALU per tap instead of 8 loads and ~55), and the weight sum uses that the eight
// trilinear weights of a tap add up to its PSF value.  Differs from the operation order of the CUDA source by fp32
// rounding only (1e-7 relative; nvcc contracts that source to FMAs anyway); FSG_TUNE_PRECISE_MATH selects
// sa_forward_kernel instead.
// ---------------------------------------------------------------------------------------------------------
typedef float sa_f2u __attribute__((ext_vector_type(2), aligned(4)));

// The per-tap work is a dependent chain (position -> address -> 2x2x2 gather -> blend), and the CUDA-ordered kernel
// waits for each tap's gathers before starting the next tap: at 8 waves per SIMD it runs at gather latency, not at
// throughput.  Here the taps with a non-zero PSF value are compacted once per workgroup into LDS as
// (rotated offset, value) -- one ds_read_b128 per tap -- and walked four at a time with the bounds test turned into
// a predicate (safe address, zero weight), so 16 independent 8-byte gathers are in flight per lane before the first
// blend.  Waves whose pixels all lie farther from the volume than the PSF radius leave at once.
__global__ __launch_bounds__(SA_TILE* SA_TILE) void sa_forward_linear_fast_kernel(SaParams P, const float* __restrict__ vol,
                                                                                  float* __restrict__ slices,
                                                                                  float* __restrict__ weights) {
  extern __shared__ float smem[];
  // XCD-aware work order: workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so with the
  // natural (tile, slice) order all eight L2s stream the same plate of the volume.  Linear id L -> XCD L % 8 owns the
  // slices {x, x+8, ...}, walking each slice tile by tile: every plate is fetched into one L2 only.
  const int tiles_x = (P.w + SA_TILE - 1) / SA_TILE, tiles = tiles_x * ((P.h + SA_TILE - 1) / SA_TILE);
  const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
  const int in = (m / tiles) * 8 + xcd, tile = m % tiles;
  if (in >= P.n) return;
  const int bx = tile % tiles_x, by = tile / tiles_x;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  if (P.fwd_select == 1 && sa_direct_cost(T, P.fwd_wy) > P.fwd_thr) return;  // this slice is the plate kernel's (uniform)
  const int np = P.pd * P.ph * P.pw;
  float4* taps = reinterpret_cast<float4*>(smem);  // [<= np] (ox, oy, oz, psf)
  __shared__ int ntaps_s;
  const int tid = threadIdx.y * SA_TILE + threadIdx.x;
  if (tid < 64) {  // wave 0: raster-order compaction by ballot prefix
    int count = 0;
    for (int base = 0; base < np; base += 64) {
      const int e = base + tid;
      const float pv = e < np ? P.psf[e] : 0.f;
      const bool nz = pv != 0.f;
      const unsigned long long bal = __ballot(nz);
      if (nz) {
        const int kx = e % P.pw, ky = (e / P.pw) % P.ph, kz = e / (P.pw * P.ph);
        const float fx = (float)(kx - P.pw / 2), fy = (float)(ky - P.ph / 2), fz = (float)(kz - P.pd / 2);
        taps[count + __popcll(bal & ((1ull << tid) - 1ull))] =
            make_float4(T[0] * fx + T[1] * fy + T[2] * fz, T[4] * fx + T[5] * fy + T[6] * fz,
                        T[8] * fx + T[9] * fy + T[10] * fz, pv);
      }
      count += __popcll(bal);
    }
    if (tid == 0) ntaps_s = count;
  }
  __syncthreads();
  const int nt = ntaps_s;
  const int ix = bx * SA_TILE + threadIdx.x, iy = by * SA_TILE + threadIdx.y;
  const bool inside = ix < P.w && iy < P.h;
  const size_t idx = ((size_t)in * P.h + (inside ? iy : 0)) * P.w + (inside ? ix : 0);
  bool live = inside && !(P.smask && !P.smask[idx]);
  float xc, yc, zc;
  sa_centre(P, T, ix, iy, xc, yc, zc);
  const float hx = (float)(P.W - 1), hy = (float)(P.H - 1), hz = (float)(P.D - 1);
  const float rad = 0.5f * sqrtf((float)(P.pw * P.pw + P.ph * P.ph + P.pd * P.pd)) + 1.f;
  if (xc < -rad || yc < -rad || zc < -rad || xc > hx + rad || yc > hy + rad || zc > hz + rad) live = false;
  float val = 0.f, weight = 0.f;
  if (__any(live)) {
    const int Sy = P.W, Sz = P.H * P.W;
    for (int t = 0; t < nt; t += 4) {
      float pvv[4], wxx[4], wyy[4], wzz[4];
      const float* qq[4];
      bool anyin = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 tp = taps[min(t + u, nt - 1)];
        const float x = xc + tp.x, y = yc + tp.y, z = zc + tp.z;
        const bool ok = live && (t + u < nt) && !(x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz);
        const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
        wxx[u] = x - xf; wyy[u] = y - yf; wzz[u] = z - zf;
        pvv[u] = ok ? tp.w : 0.f;
        qq[u] = ok ? vol + ((int)zf * Sz + (int)yf * Sy + (int)xf) : vol;
        anyin |= ok;
      }
      if (!__any(anyin)) continue;
      sa_f2u p00[4], p10[4], p01[4], p11[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p00[u] = *reinterpret_cast<const sa_f2u*>(qq[u]);
        p10[u] = *reinterpret_cast<const sa_f2u*>(qq[u] + Sy);
        p01[u] = *reinterpret_cast<const sa_f2u*>(qq[u] + Sz);
        p11[u] = *reinterpret_cast<const sa_f2u*>(qq[u] + Sz + Sy);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float wx = wxx[u], wy = wyy[u], wz = wzz[u];
        const float a00 = p00[u].x + wx * (p00[u].y - p00[u].x), a10 = p10[u].x + wx * (p10[u].y - p10[u].x);
        const float a01 = p01[u].x + wx * (p01[u].y - p01[u].x), a11 = p11[u].x + wx * (p11[u].y - p11[u].x);
        const float b0 = a00 + wy * (a10 - a00), b1 = a01 + wy * (a11 - a01);
        val += pvv[u] * (b0 + wz * (b1 - b0));
        weight += pvv[u];
      }
    }
  }
  if (!inside) return;
  const bool good = weight > 0.f;
  slices[idx] = good ? val / weight : 0.f;
  if (weights) weights[idx] = good ? weight : 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// forward, interp_psf = false, no volume mask, the plate of the volume in LDS (r03; selected by default for this case).
//
// The kernel above is bound by the L1/TA gather path: four 8-byte wave-gathers per PSF tap, each charged per cache line its
// pixels touch.  Here a wave owns a 4 x 4 pixel tile (a workgroup the same 16 x 16 tile: four tiles per wave, one after the
// other) and its 64 lanes are 16 pixels x 4 tap phases: phase p walks taps p, p + 4, ... of the compacted list, so the four
// lane groups read four neighbouring taps of the same 16 pixels.  The taps are walked in chunks of whole PSF planes; for a chunk
// the wave copies the bounding box of everything its 16 pixels sample there -- the tile's pixel centres +- the chunk's rotated
// tap extent, clipped to the volume -- into its own 8 KB of LDS with coalesced row loads, and the 2 x 2 x 2 corners come from
// LDS.  As many consecutive planes as fit form a chunk; a plane that does not fit by itself takes the direct gathers of the
// kernel above.  A small tile keeps the box of an OBLIQUE plate small (an 8 x 8 tile's box is 3 300-6 500 floats per plane,
// mostly empty: profiles/r03_g_notes.txt); the price is the order of the sum over the taps -- four partial sums per pixel,
// combined (p0 + p1) + (p2 + p3) -- i.e. fp32 rounding against the kernel above (the tests' tolerances, 1e-5 relative).
// No workgroup barrier after the tap compaction.
// ---------------------------------------------------------------------------------------------------------
constexpr int SAP_CAP = 2048;  // floats of plate per wave

__device__ __forceinline__ void sap_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float sap_uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

__global__ __launch_bounds__(256) void sa_forward_plate_kernel(SaParams P, const float* __restrict__ vol,
                                                               float* __restrict__ slices, float* __restrict__ weights) {
  extern __shared__ float smem[];
  const int tiles_x = (P.w + SA_TILE - 1) / SA_TILE, tiles = tiles_x * ((P.h + SA_TILE - 1) / SA_TILE);
  const int xcd = blockIdx.x & 7, mq = blockIdx.x >> 3;
  const int in = (mq / tiles) * 8 + xcd, tile = mq % tiles;
  if (in >= P.n) return;
  const int bx = tile % tiles_x, by = tile / tiles_x;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  if (P.fwd_select == 2 && !(sa_direct_cost(T, P.fwd_wy) > P.fwd_thr)) return;  // this slice is the direct kernel's (uniform)
  const int np = P.pd * P.ph * P.pw;
  float4* taps = reinterpret_cast<float4*>(smem);              // [<= np] (ox, oy, oz, psf), raster order
  int* pstart = reinterpret_cast<int*>(taps + np);             // [pd + 1] first compacted tap of every PSF plane
  float* pext = reinterpret_cast<float*>(pstart + P.pd + 1);   // [pd][6] min / max rotated offset of the plane's tap rectangle
  // [4 waves][SAP_CAP], 16-byte aligned.  (As an OFFSET into smem: rounding the pointer up through uintptr_t makes it a generic
  // pointer, and every access to the plate went out as a flat load / store through the vector-memory path -- 150 M vector-memory
  // reads per launch, more than the direct kernel's gathers: profiles/r03_i_slice_acq_forward.txt.)
  float* boxes = smem + ((4 * np + (P.pd + 1) + 6 * P.pd + 3) & ~3);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (tid < 64) {  // wave 0: raster-order compaction by ballot prefix (as sa_forward_linear_fast_kernel) + plane starts
    int count = 0;
    const int pp = P.ph * P.pw;
    for (int base = 0; base < np; base += 64) {
      const int e = base + tid;
      const float pv = e < np ? P.psf[e] : 0.f;
      const bool nz = pv != 0.f;
      const unsigned long long bal = __ballot(nz);
      const int before = count + __popcll(bal & ((1ull << tid) - 1ull));
      const int kz = e / pp;
      if (e < np && e - kz * pp == 0) pstart[kz] = before;
      if (nz) {
        const int kx = e % P.pw, ky = (e / P.pw) % P.ph;
        const float fx = (float)(kx - P.pw / 2), fy = (float)(ky - P.ph / 2), fz = (float)(kz - P.pd / 2);
        taps[before] = make_float4(T[0] * fx + T[1] * fy + T[2] * fz, T[4] * fx + T[5] * fy + T[6] * fz,
                                   T[8] * fx + T[9] * fy + T[10] * fz, pv);
      }
      count += __popcll(bal);
    }
    if (tid == 0) pstart[P.pd] = count;
    if (tid < P.pd) {  // a linear map takes its extremes over a rectangle at the corners
      const float fz = (float)(tid - P.pd / 2);
      const float x0 = (float)(-(P.pw / 2)), x1 = (float)(P.pw - 1 - P.pw / 2), y0 = (float)(-(P.ph / 2)), y1 = (float)(P.ph - 1 - P.ph / 2);
      for (int a = 0; a < 3; ++a) {
        const float c0 = T[4 * a] * x0 + T[4 * a + 1] * y0, c1 = T[4 * a] * x1 + T[4 * a + 1] * y0;
        const float c2 = T[4 * a] * x0 + T[4 * a + 1] * y1, c3 = T[4 * a] * x1 + T[4 * a + 1] * y1;
        const float zt = T[4 * a + 2] * fz;
        pext[6 * tid + a] = fminf(fminf(c0, c1), fminf(c2, c3)) + zt - 0.01f;      // (margin: the rounding of the per-tap products)
        pext[6 * tid + 3 + a] = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3)) + zt + 0.01f;
      }
    }
  }
  __syncthreads();  // the only workgroup barrier
  const int l = lane & 15, phase = lane >> 4;
  const float hx = (float)(P.W - 1), hy = (float)(P.H - 1), hz = (float)(P.D - 1);
  const float rad = 0.5f * sqrtf((float)(P.pw * P.pw + P.ph * P.ph + P.pd * P.pd)) + 1.f;
  const int Sy = P.W, Sz = P.H * P.W;
  float* box = boxes + (size_t)wave * SAP_CAP;
  const float BIG = 3.0e38f;
  for (int tt = 0; tt < 4; ++tt) {
    const int t16 = wave * 4 + tt;
    const int ix = bx * SA_TILE + (t16 & 3) * 4 + (l & 3), iy = by * SA_TILE + (t16 >> 2) * 4 + (l >> 2);
    const bool inside = ix < P.w && iy < P.h;
    const size_t idx = ((size_t)in * P.h + (inside ? iy : 0)) * P.w + (inside ? ix : 0);
    bool live = inside && !(P.smask && !P.smask[idx]);
    float xc, yc, zc;
    sa_centre(P, T, ix, iy, xc, yc, zc);
    if (xc < -rad || yc < -rad || zc < -rad || xc > hx + rad || yc > hy + rad || zc > hz + rad) live = false;
    float val = 0.f, weight = 0.f;
    if (__any(live)) {
      const float cx0 = sap_uni(fsg_wave_min(live ? xc : BIG)), cx1 = sap_uni(fsg_wave_max(live ? xc : -BIG));
      const float cy0 = sap_uni(fsg_wave_min(live ? yc : BIG)), cy1 = sap_uni(fsg_wave_max(live ? yc : -BIG));
      const float cz0 = sap_uni(fsg_wave_min(live ? zc : BIG)), cz1 = sap_uni(fsg_wave_max(live ? zc : -BIG));
      int kz = 0;
      while (kz < P.pd) {
        // ---- the longest run of planes [kz, kz1) whose plate fits ----
        int kz1 = kz, X0 = 0, Y0 = 0, Z0 = 0, ex = 0, ey = 0, ez = 0, pitch = 16;
        float o0 = BIG, o1 = BIG, o2 = BIG, o3 = -BIG, o4 = -BIG, o5 = -BIG;
        for (int k = kz; k < P.pd; ++k) {
          const float a0 = fminf(o0, sap_uni(pext[6 * k])), a1 = fminf(o1, sap_uni(pext[6 * k + 1])), a2 = fminf(o2, sap_uni(pext[6 * k + 2]));
          const float b0 = fmaxf(o3, sap_uni(pext[6 * k + 3])), b1 = fmaxf(o4, sap_uni(pext[6 * k + 4])), b2 = fmaxf(o5, sap_uni(pext[6 * k + 5]));
          // samples outside [0, h) contribute nothing: the plate is clipped to the volume
          // (BOTH ends clamped into the volume: a tile whose samples all lie beyond a face keeps a one-voxel box inside it --
          // its samples are rejected by the bounds test and never read the box)
          const int tx0 = min(max((int)floorf(cx0 + a0), 0), P.W - 1), tx1 = min(max((int)floorf(cx1 + b0) + 1, 0), P.W - 1);
          const int ty0 = min(max((int)floorf(cy0 + a1), 0), P.H - 1), ty1 = min(max((int)floorf(cy1 + b1) + 1, 0), P.H - 1);
          const int tz0 = min(max((int)floorf(cz0 + a2), 0), P.D - 1), tz1 = min(max((int)floorf(cz1 + b2) + 1, 0), P.D - 1);
          const int nx = max(tx1 - tx0 + 1, 1), ny = max(ty1 - ty0 + 1, 1), nz = max(tz1 - tz0 + 1, 1);
          const int pt = nx <= 4 ? 4 : (nx <= 8 ? 8 : (nx <= 16 ? 16 : 32));  // row pitch: a slice tilted towards z has a narrow box along x
          if (nx > 32 || pt * ny * nz > SAP_CAP) break;
          o0 = a0; o1 = a1; o2 = a2; o3 = b0; o4 = b1; o5 = b2;
          X0 = tx0; Y0 = ty0; Z0 = tz0; ex = nx; ey = ny; ez = nz; pitch = pt;
          kz1 = k + 1;
        }
        const bool plate = kz1 > kz;  // uniform
        if (!plate) kz1 = kz + 1;     // this plane by direct gathers
        const int t_beg = __builtin_amdgcn_readfirstlane(pstart[kz]), t_end = __builtin_amdgcn_readfirstlane(pstart[kz1]);
        if (plate && t_end > t_beg) {
          // ---- copy the plate: 64 / pitch rows per wave-instruction, four instructions in flight ----
          const int nrow = ey * ez, rpi = 64 / pitch, xl = lane & (pitch - 1), rsub = lane / pitch;
          const float inv_ey = 1.0f / (float)ey;
          const bool xin = xl < ex;
          for (int r0 = 0; r0 < nrow; r0 += 4 * rpi) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int r = r0 + u * rpi + rsub;
              const int rz = (int)(((float)r + 0.5f) * inv_ey), ry = r - rz * ey;
              v[u] = (r < nrow && xin) ? vol[(size_t)(Z0 + rz) * Sz + (size_t)(Y0 + ry) * Sy + (X0 + xl)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int r = r0 + u * rpi + rsub;
              if (r < nrow) box[r * pitch + xl] = v[u];
            }
          }
          sap_wave_sync();
        }
        const int py = pitch, pz = pitch * ey, org = Z0 * pz + Y0 * py + X0;
        for (int t = t_beg + phase; t < t_end; t += 8) {  // two taps of this phase per trip
          float pvv[2], wxx[2], wyy[2], wzz[2];
          const float* qq[2];
          int qi[2];
          bool anyin = false;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int tu = t + 4 * u;
            const float4 tp = taps[min(tu, t_end - 1)];
            const float x = xc + tp.x, y = yc + tp.y, z = zc + tp.z;
            const bool ok = live && (tu < t_end) && !(x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz);
            const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
            wxx[u] = x - xf; wyy[u] = y - yf; wzz[u] = z - zf;
            pvv[u] = ok ? tp.w : 0.f;
            const int xi = (int)xf, yi = (int)yf, zi = (int)zf;
            qq[u] = ok ? vol + (zi * Sz + yi * Sy + xi) : vol;
            qi[u] = ok ? zi * pz + yi * py + xi - org : 0;
            anyin |= ok;
          }
          if (!__any(anyin)) continue;
          sa_f2u p00[2], p10[2], p01[2], p11[2];
          if (plate) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const float* b = box + qi[u];
              p00[u] = sa_f2u{b[0], b[1]};
              p10[u] = sa_f2u{b[py], b[py + 1]};
              p01[u] = sa_f2u{b[pz], b[pz + 1]};
              p11[u] = sa_f2u{b[pz + py], b[pz + py + 1]};
            }
          } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              p00[u] = *reinterpret_cast<const sa_f2u*>(qq[u]);
              p10[u] = *reinterpret_cast<const sa_f2u*>(qq[u] + Sy);
              p01[u] = *reinterpret_cast<const sa_f2u*>(qq[u] + Sz);
              p11[u] = *reinterpret_cast<const sa_f2u*>(qq[u] + Sz + Sy);
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float wx = wxx[u], wy = wyy[u], wz = wzz[u];
            const float a00 = p00[u].x + wx * (p00[u].y - p00[u].x), a10 = p10[u].x + wx * (p10[u].y - p10[u].x);
            const float a01 = p01[u].x + wx * (p01[u].y - p01[u].x), a11 = p11[u].x + wx * (p11[u].y - p11[u].x);
            const float b0 = a00 + wy * (a10 - a00), b1 = a01 + wy * (a11 - a01);
            val += pvv[u] * (b0 + wz * (b1 - b0));
            weight += pvv[u];
          }
        }
        if (plate && t_end > t_beg) sap_wave_sync();  // the next chunk's copy overwrites the plate
        kz = kz1;
      }
    }
    // the four phases of a pixel: (p0 + p1) + (p2 + p3)
    val += __shfl_xor(val, 16, FSG_WAVE);
    weight += __shfl_xor(weight, 16, FSG_WAVE);
    val += __shfl_xor(val, 32, FSG_WAVE);
    weight += __shfl_xor(weight, 32, FSG_WAVE);
    if (inside && phase == 0) {
      const bool good = weight > 0.f;
      slices[idx] = good ? val / weight : 0.f;
      if (weights) weights[idx] = good ? weight : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// adjoint, CUDA-kernel semantics (:472-670): pass 1 = pixel weight, pass 2 = scatter with fp32 atomics.
// ---------------------------------------------------------------------------------------------------------
template <bool NN, bool VM>
__global__ __launch_bounds__(SA_TILE* SA_TILE) void sa_adjoint_kernel(SaParams P, const float* __restrict__ slices,
                                                                      float* __restrict__ vol,
                                                                      float* __restrict__ vol_weight) {
  extern __shared__ float smem[];
  const int in = blockIdx.z;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  const SaLds L = sa_lds_layout(smem, P);
  sa_fill_lds(L, P, T);
  __syncthreads();
  const int ix = blockIdx.x * SA_TILE + threadIdx.x, iy = blockIdx.y * SA_TILE + threadIdx.y;
  if (ix >= P.w || iy >= P.h) return;
  const size_t idx = ((size_t)(P.sid ? P.sid[in] : in) * P.h + iy) * P.w + ix;
  if (P.smask && !P.smask[idx]) return;
  const float s = slices[idx];
  float xc, yc, zc;
  sa_centre(P, T, ix, iy, xc, yc, zc);
  const int Sy = P.W, Sz = P.H * P.W;
  const float hx = (float)(P.W - 1), hy = (float)(P.H - 1), hz = (float)(P.D - 1);

  float weight = 0.f;
  int ip = 0;
  for (int kz = 0; kz < P.pd; ++kz) {
    const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
    for (int ky = 0; ky < P.ph; ++ky) {
      const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
      for (int kx = 0; kx < P.pw; ++kx, ++ip) {
        float pv = L.psf[ip];
        if (pv == 0.f) continue;
        const float x = xc + L.tx[kx] + yx + zx;
        const float y = yc + L.tx[P.pw + kx] + yy + zy;
        const float z = zc + L.tx[2 * P.pw + kx] + yz + zz;
        if (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz) continue;
        if (NN && !sa_psf_at(L, P, T, roundf(x) - xc, roundf(y) - yc, roundf(z) - zc, pv)) continue;
        weight += pv;
      }
    }
  }
  if (weight < 0.5f) return;  // border

  ip = 0;
  for (int kz = 0; kz < P.pd; ++kz) {
    const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
    for (int ky = 0; ky < P.ph; ++ky) {
      const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
      for (int kx = 0; kx < P.pw; ++kx, ++ip) {
        float pv = L.psf[ip];
        if (pv == 0.f) continue;
        const float x = xc + L.tx[kx] + yx + zx;
        const float y = yc + L.tx[P.pw + kx] + yy + zy;
        const float z = zc + L.tx[2 * P.pw + kx] + yz + zz;
        if (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz) continue;
        if (NN) {
          const float xr = roundf(x), yr = roundf(y), zr = roundf(z);
          if (!sa_psf_at(L, P, T, xr - xc, yr - yc, zr - zc, pv)) continue;
          pv /= weight;
          const int iv = (int)zr * Sz + (int)yr * Sy + (int)xr;
          if (VM && !P.vmask[iv]) continue;
          unsafeAtomicAdd(vol + iv, pv * s);
          if (vol_weight) unsafeAtomicAdd(vol_weight + iv, pv);
        } else {
          const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
          const float wx = x - xf, wy = y - yf, wz = z - zf;
          const int iv = (int)zf * Sz + (int)yf * Sy + (int)xf;
          pv /= weight;
          float pw_;
#define SA_CORNER(OFF, WEXPR)                                            \
  if (!VM || P.vmask[iv + (OFF)]) {                                      \
    pw_ = (WEXPR) * pv;                                                  \
    unsafeAtomicAdd(vol + iv + (OFF), pw_ * s);                          \
    if (vol_weight) unsafeAtomicAdd(vol_weight + iv + (OFF), pw_);       \
  }
          SA_CORNER(0, (1 - wx) * (1 - wy) * (1 - wz))
          SA_CORNER(1, wx * (1 - wy) * (1 - wz))
          SA_CORNER(Sy, (1 - wx) * wy * (1 - wz))
          SA_CORNER(Sz, (1 - wx) * (1 - wy) * wz)
          SA_CORNER(1 + Sy, wx * wy * (1 - wz))
          SA_CORNER(1 + Sz, wx * (1 - wy) * wz)
          SA_CORNER(Sy + Sz, (1 - wx) * wy * wz)
          SA_CORNER(Sy + Sz + 1, wx * wy * wz)
#undef SA_CORNER
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// adjoint, interp_psf = true, with on-chip pre-summation.
//
// The scatter form issues two fp32 atomics per (pixel, tap): 7e9 of them for a 200-slice reconstruction at
// 384^3, and global float atomics execute memory-side at a fixed byte rate (MI355X_MICROARCH.md, "Global float
// atomics": ~1.3 TB/s for well-shaped wave instructions, 17x less for 64 lanes in 64 rows) -- the direct kernel
// above spends 3/4 of its time there.  A tile of neighbouring pixels revisits the same voxels many times (taps
// are 1 voxel apart, pixels 0.5-2 voxels apart), so this kernel sums a tile's contributions in LDS first:
//   * workgroup = T x T pixel tile of one slice (T = 16: one pixel per thread; T = 8: wave g of the workgroup
//     takes the PSF rows ky = g mod 4 of every pixel of the tile);
//   * the PSF is walked in chunks of `zc` planes; per chunk the axis-aligned bounding box of the voxels the
//     tile can reach is derived from the slice transform; if it fits the LDS accumulator (`cap` cells for
//     value and weight) contributions go to LDS with ds_add_f32, and the box is flushed with one global
//     atomic per touched cell in row order (contiguous x runs); otherwise the chunk falls back to direct
//     global atomics.  A cell index outside the box (cannot happen by construction) also falls back.
// Same sums as the direct kernel up to fp32 summation order (which is nondeterministic there as well).
// ---------------------------------------------------------------------------------------------------------
// roundf(x) for 0 <= x < 2^23 in two instructions: floor(x + pred(0.5)).  Ties n + 0.5 still reach n + 1 (the sum rounds up to
// the integer), and the one value below a tie, n + 0.5 - ulp, stays at n (with + 0.5 the sum for 0.5 - 2^-25 would round to 1).
// Checked over all 1 258 291 200 floats of the range (tests/test_round_identity.py keeps the ties and their neighbours).
__device__ __forceinline__ float sa_round_pos(float x) { return floorf(x + 0.49999997f); }

__device__ __forceinline__ bool sa_psf_at_fast(const SaLds& L, const SaParams& P, const float* __restrict__ T, float dx,
                                               float dy, float dz, float& val) {
  // float adds: fl32(fl64(a + b)) == fl32(a + b) for fp32 a, b (53 >= 2*24 + 2), so no double is needed here
  const float xp = (T[0] * dx + T[4] * dy + T[8] * dz) + (float)(P.pw - 1) * 0.5f;
  const float yp = (T[1] * dx + T[5] * dy + T[9] * dz) + (float)(P.ph - 1) * 0.5f;
  const float zp = (T[2] * dx + T[6] * dy + T[10] * dz) + (float)(P.pd - 1) * 0.5f;
  // 0 <= p < n - 1  <=>  (unsigned)floor(p) < n - 1 (n - 1 is an integer; -0.0 and NaN pass both forms): three compares, not six
  const float xf = floorf(xp), yf = floorf(yp), zf = floorf(zp);
  const int jx = (int)xf, jy = (int)yf, jz = (int)zf;
  if ((unsigned)jx >= (unsigned)(P.pw - 1) || (unsigned)jy >= (unsigned)(P.ph - 1) || (unsigned)jz >= (unsigned)(P.pd - 1))
    return false;
  const float wx = xp - xf, wy = yp - yf, wz = zp - zf;
  const int sy = P.pw, sz = P.pw * P.ph;
  const float* q = L.psf + jz * sz + jy * sy + jx;
  const float a00 = q[0] + wx * (q[1] - q[0]), a10 = q[sy] + wx * (q[sy + 1] - q[sy]);
  const float a01 = q[sz] + wx * (q[sz + 1] - q[sz]), a11 = q[sz + sy] + wx * (q[sz + sy + 1] - q[sz + sy]);
  const float b0 = a00 + wy * (a10 - a00), b1 = a01 + wy * (a11 - a01);
  val = b0 + wz * (b1 - b0);
  return true;
}

template <int T_, bool VM>
__global__ __launch_bounds__(256) void sa_adjoint_nn_lds_kernel(SaParams P, const float* __restrict__ slices,
                                                                float* __restrict__ vol, float* __restrict__ vol_weight,
                                                                int zc, int cap) {
  constexpr int NPIX = T_ * T_, G = 256 / NPIX;
  extern __shared__ float smem[];
  const int in = blockIdx.z;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  const SaLds L = sa_lds_layout(smem, P);
  float* part = L.tz + 3 * P.pd;  // [256] partial pixel weights
  float* accv = part + 256;
  float* accw = accv + cap;
  {
    // sa_fill_lds with a 256-thread linear id
    const int tid = threadIdx.x, np = P.pd * P.ph * P.pw;
    for (int e = tid; e < np; e += 256) L.psf[e] = P.psf[e];
    for (int e = tid; e < 3 * P.pw; e += 256) { const int a = e / P.pw, i = e - a * P.pw; L.tx[e] = T[a * 4 + 0] * (float)(i - P.pw / 2); }
    for (int e = tid; e < 3 * P.ph; e += 256) { const int a = e / P.ph, i = e - a * P.ph; L.ty[e] = T[a * 4 + 1] * (float)(i - P.ph / 2); }
    for (int e = tid; e < 3 * P.pd; e += 256) { const int a = e / P.pd, i = e - a * P.pd; L.tz[e] = T[a * 4 + 2] * (float)(i - P.pd / 2); }
  }
  __syncthreads();
  const int tid = threadIdx.x;
  const int p = tid % NPIX, g = tid / NPIX;
  const int ix0 = blockIdx.x * T_, iy0 = blockIdx.y * T_;
  const int ix = ix0 + p % T_, iy = iy0 + p / T_;
  bool live = ix < P.w && iy < P.h;
  const size_t idx = ((size_t)(P.sid ? P.sid[in] : in) * P.h + (live ? iy : 0)) * P.w + (live ? ix : 0);
  if (live && P.smask && !P.smask[idx]) live = false;
  const float s = live ? slices[idx] : 0.f;
  float xc, yc, zc_;
  sa_centre(P, T, ix, iy, xc, yc, zc_);
  const int Sy = P.W, Sz = P.H * P.W;
  const float hx = (float)(P.W - 1), hy = (float)(P.H - 1), hz = (float)(P.D - 1);

  // tile extent in slice coordinates
  const int ixl = ix0, ixh = min(ix0 + T_ - 1, P.w - 1), iyl = iy0, iyh = min(iy0 + T_ - 1, P.h - 1);
  const float alo = ((float)ixl - (float)(P.w - 1) * 0.5f) * P.res + T[3] - (float)(P.pw / 2);
  const float ahi = ((float)ixh - (float)(P.w - 1) * 0.5f) * P.res + T[3] + (float)(P.pw - 1 - P.pw / 2);
  const float blo = ((float)iyl - (float)(P.h - 1) * 0.5f) * P.res + T[7] - (float)(P.ph / 2);
  const float bhi = ((float)iyh - (float)(P.h - 1) * 0.5f) * P.res + T[7] + (float)(P.ph - 1 - P.ph / 2);
  const int dimv[3] = {P.W, P.H, P.D};
  // axis-aligned box of the positions the tile's pixels can reach through PSF planes [clo, chi] (slice coordinates), clipped to
  // the volume: origin o, extent b; returns whether the unclipped box lies inside the volume by the fp32 slack of the position
  // sums -- then no tap of the tile needs the per-tap inside test (uniform)
  auto reach = [&](float clo, float chi, int* o, int* b) -> bool {
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float r0 = T[a * 4], r1 = T[a * 4 + 1], r2 = T[a * 4 + 2];
      const float off = (float)(dimv[a] - 1) * 0.5f;
      const float mn = off + fminf(r0 * alo, r0 * ahi) + fminf(r1 * blo, r1 * bhi) + fminf(r2 * clo, r2 * chi);
      const float mx = off + fmaxf(r0 * alo, r0 * ahi) + fmaxf(r1 * blo, r1 * bhi) + fmaxf(r2 * clo, r2 * chi);
      // voxel = round(position): [floor(mn + .5), floor(mx + .5)], widened by the slack
      const int lo = max((int)floorf(mn + 0.5f - 2e-3f), 0), hi = min((int)floorf(mx + 0.5f + 2e-3f), dimv[a] - 1);
      o[a] = lo;
      b[a] = hi - lo + 1;
      inside = inside && mn - 2e-3f >= 0.f && mx + 2e-3f < (float)(dimv[a] - 1);
    }
    return inside;
  };
  int o_all[3], b_all[3];
  const bool inside_all = reach(T[11] + (float)(0 - P.pd / 2), T[11] + (float)(P.pd - 1 - P.pd / 2), o_all, b_all);

  // pass 1: pixel weight (slice_acq_cuda_kernel.cu:515-560), PSF rows split over the G waves of a pixel
  float wsum = 0.f;
  if (live) {
    for (int kz = 0; kz < P.pd; ++kz) {
      const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
      for (int ky = g; ky < P.ph; ky += G) {
        const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
        const int ip0 = (kz * P.ph + ky) * P.pw;
        for (int kx = 0; kx < P.pw; ++kx) {
          float pv = L.psf[ip0 + kx];
          if (pv == 0.f) continue;
          const float x = xc + L.tx[kx] + yx + zx;
          const float y = yc + L.tx[P.pw + kx] + yy + zy;
          const float z = zc_ + L.tx[2 * P.pw + kx] + yz + zz;
          if (!inside_all && (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz)) continue;
          if (!sa_psf_at_fast(L, P, T, sa_round_pos(x) - xc, sa_round_pos(y) - yc, sa_round_pos(z) - zc_, pv)) continue;
          wsum += pv;
        }
      }
    }
  }
  if (G > 1) {
    part[tid] = wsum;
    __syncthreads();
    wsum = 0.f;
#pragma unroll
    for (int q = 0; q < G; ++q) wsum += part[q * NPIX + p];
  }
  if (wsum < 0.5f) live = false;
  const float inv = live ? 1.f / wsum : 0.f;

  // LDS cell layout: a chunk of PSF planes of a pixel tile is an oblique plate in the volume, whose axis-aligned
  // bounding box can hold 10x more cells than the plate.  Cells are therefore indexed by the two volume axes (u, v)
  // other than the one the slice normal leans on most (a), plus the offset k of the a-coordinate from the plate's
  // mid-plane at (u, v): K = plate thickness along a, a few cells.  x is kept the fastest index for the flush.
  const float n0 = T[2], n1 = T[6], n2 = T[10];  // slice normal in volume axes (x, y, z)
  const int A = (fabsf(n0) >= fabsf(n1) && fabsf(n0) >= fabsf(n2)) ? 0 : (fabsf(n1) >= fabsf(n2) ? 1 : 2);
  const int U = A == 0 ? 1 : 0, V = A == 2 ? 1 : 2;  // U < V, so x is U whenever x is not the lean axis
  const float nA = A == 0 ? n0 : (A == 1 ? n1 : n2), nU = U == 0 ? n0 : n1, nV = V == 1 ? n1 : n2;
  const float inv_na = 1.f / nA, slope = (fabsf(nU) + fabsf(nV)) * fabsf(inv_na);
  // tile centre in slice coordinates -> a point of the mid-plane (per chunk: its z)
  const float amid = 0.5f * (alo + ahi), bmid = 0.5f * (blo + bhi);

  for (int kz0 = 0; kz0 < P.pd; kz0 += zc) {
    const int kz1 = min(kz0 + zc, P.pd);
    const float clo = T[11] + (float)(kz0 - P.pd / 2), chi = T[11] + (float)(kz1 - 1 - P.pd / 2);
    int o[3], b[3];
    const bool inside = reach(clo, chi, o, b);
    if (b[0] <= 0 || b[1] <= 0 || b[2] <= 0) continue;  // the chunk cannot reach the volume (uniform)
    const float cmid = 0.5f * (clo + chi);
    float pm[3];  // mid-plane point in volume coordinates
#pragma unroll
    for (int a = 0; a < 3; ++a)
      pm[a] = (float)(dimv[a] - 1) * 0.5f + T[a * 4] * amid + T[a * 4 + 1] * bmid + T[a * 4 + 2] * cmid;
    // thickness of the plate along A: PSF planes (kz1-1-kz0) / |nA|, voxel rounding 1 + slope, plus one cell of slack
    const int K = (int)ceilf(((float)(kz1 - 1 - kz0) + 1.f) * fabsf(inv_na) + slope + 1.f) + 1;
    const float hb = 0.5f * (float)K;
    const int bu = b[U], bv = b[V];
    const long long cells = (long long)bu * bv * K;
    const bool use_lds = cells <= (long long)cap;
    // index = ((cv * K + k) * bu + cu) when x is U (x fastest); = ((cv * bu + cu) * K + k) when x is the lean axis
    const bool xlean = A == 0;
    if (use_lds) {
      for (int c = tid; c < (int)cells; c += 256) { accv[c] = 0.f; accw[c] = 0.f; }
      __syncthreads();
    }
    if (live) {
      for (int kz = kz0; kz < kz1; ++kz) {
        const float zx = L.tz[kz], zy = L.tz[P.pd + kz], zz = L.tz[2 * P.pd + kz];
        for (int ky = g; ky < P.ph; ky += G) {
          const float yx = L.ty[ky], yy = L.ty[P.ph + ky], yz = L.ty[2 * P.ph + ky];
          const int ip0 = (kz * P.ph + ky) * P.pw;
          for (int kx = 0; kx < P.pw; ++kx) {
            float pv = L.psf[ip0 + kx];
            if (pv == 0.f) continue;
            const float x = xc + L.tx[kx] + yx + zx;
            const float y = yc + L.tx[P.pw + kx] + yy + zy;
            const float z = zc_ + L.tx[2 * P.pw + kx] + yz + zz;
            if (!inside && (x < 0 || y < 0 || z < 0 || x >= hx || y >= hy || z >= hz)) continue;
            const float xr = sa_round_pos(x), yr = sa_round_pos(y), zr = sa_round_pos(z);
            if (!sa_psf_at_fast(L, P, T, xr - xc, yr - yc, zr - zc_, pv)) continue;
            pv *= inv;
            const int vv[3] = {(int)xr, (int)yr, (int)zr};
            const int iv = vv[2] * Sz + vv[1] * Sy + vv[0];
            if (VM && !P.vmask[iv]) continue;
            bool done = false;
            if (use_lds) {
              const int cu = vv[U] - o[U], cv = vv[V] - o[V];
              const float fu = (float)vv[U] - pm[U], fv = (float)vv[V] - pm[V];
              const int base = (int)floorf(pm[A] - (nU * fu + nV * fv) * inv_na - hb);
              const int k = vv[A] - base;
              if ((unsigned)cu < (unsigned)bu && (unsigned)cv < (unsigned)bv && (unsigned)k < (unsigned)K) {
                const int c = xlean ? (cv * bu + cu) * K + k : (cv * K + k) * bu + cu;
                atomicAdd(&accv[c], pv * s);
                atomicAdd(&accw[c], pv);
                done = true;
              }
            }
            if (!done) {
              unsafeAtomicAdd(vol + iv, pv * s);
              if (vol_weight) unsafeAtomicAdd(vol_weight + iv, pv);
            }
          }
        }
      }
    }
    if (use_lds) {
      __syncthreads();
      for (int c = tid; c < (int)cells; c += 256) {
        const float wv = accw[c];
        if (wv == 0.f) continue;
        int cu, cv, k;
        if (xlean) { k = c % K; const int r = c / K; cu = r % bu; cv = r / bu; }
        else { cu = c % bu; const int r = c / bu; k = r % K; cv = r / K; }
        int vv[3];
        vv[U] = cu + o[U];
        vv[V] = cv + o[V];
        const float fu = (float)vv[U] - pm[U], fv = (float)vv[V] - pm[V];
        vv[A] = (int)floorf(pm[A] - (nU * fu + nV * fv) * inv_na - hb) + k;
        const int iv = vv[2] * Sz + vv[1] * Sy + vv[0];
        unsafeAtomicAdd(vol + iv, accv[c]);
        if (vol_weight) unsafeAtomicAdd(vol_weight + iv, wv);
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// torch-fallback semantics (slice_acq.py:272-310): position = (shift + R((off - T) + T)) + R(pixel + T),
// strict inside test, round half to even, raw PSF value; one kernel serves forward and adjoint.
// LDS: per-tap rotated offsets (3 floats) next to the PSF.
// ---------------------------------------------------------------------------------------------------------
template <bool ADJ, bool GS = false>
__global__ __launch_bounds__(SA_TILE* SA_TILE) void sa_torch_kernel(SaParams P, const float* __restrict__ vol_in,
                                                                    float* __restrict__ slices_io,
                                                                    float* __restrict__ weights_out,
                                                                    float* __restrict__ vol_out,
                                                                    float* __restrict__ vol_weight) {
  extern __shared__ float smem[];
  const int in = blockIdx.z;
  const float* __restrict__ T = P.tr + (size_t)in * 12;
  const int np = P.pd * P.ph * P.pw;
  float* lp = smem;           // psf
  float* lo = smem + np;      // [np][3] shift + rotated tap offset
  const int tid = threadIdx.y * SA_TILE + threadIdx.x;
  const float shx = ((float)P.W - 1.f) / 2.0f, shy = ((float)P.H - 1.f) / 2.0f, shz = ((float)P.D - 1.f) / 2.0f;
  for (int e = tid; e < np; e += SA_TILE * SA_TILE) {
    lp[e] = P.psf[e];
    const int kx = e % P.pw, ky = (e / P.pw) % P.ph, kz = e / (P.pw * P.ph);
    // xyz_masked_untransformed(psf > 0, shape, 1.0) (:266-269): (index - (n-1)/2) * 1.0 in fp32
    const float ox = ((float)kx - ((float)P.pw - 1.f) / 2.f) * 1.0f;
    const float oy = ((float)ky - ((float)P.ph - 1.f) / 2.f) * 1.0f;
    const float oz = ((float)kz - ((float)P.pd - 1.f) / 2.f) * 1.0f;
    const float ax = (ox - T[3]) + T[3], ay = (oy - T[7]) + T[7], az = (oz - T[11]) + T[11];
    lo[3 * e + 0] = shx + (T[0] * ax + T[1] * ay + T[2] * az);
    lo[3 * e + 1] = shy + (T[4] * ax + T[5] * ay + T[6] * az);
    lo[3 * e + 2] = shz + (T[8] * ax + T[9] * ay + T[10] * az);
  }
  __syncthreads();
  const int ix = blockIdx.x * SA_TILE + threadIdx.x, iy = blockIdx.y * SA_TILE + threadIdx.y;
  if (ix >= P.w || iy >= P.h) return;
  const size_t idx = ((size_t)((ADJ && P.sid) ? P.sid[in] : in) * P.h + iy) * P.w + ix;
  const bool live = !P.smask || P.smask[idx];
  if (!live) {
    if (!ADJ) {
      slices_io[idx] = 0.f;
      if (weights_out) weights_out[idx] = 0.f;
    }
    return;
  }
  const float px = (((float)ix - ((float)P.w - 1.f) / 2.f) * P.res) + T[3];
  const float py = (((float)iy - ((float)P.h - 1.f) / 2.f) * P.res) + T[7];
  const float pz = (0.f * P.res) + T[11];
  const float sx = T[0] * px + T[1] * py + T[2] * pz;
  const float sy = T[4] * px + T[5] * py + T[6] * pz;
  const float sz = T[8] * px + T[9] * py + T[10] * pz;
  const float lx = shx * 2, ly = shy * 2, lz = shz * 2;
  const int Sy = P.W, Sz = P.H * P.W;
  if (GS) {
    // 1x1x1 PSF without weights: slice_acquisition_no_psf_torch (slice_acq.py:445-480) = F.grid_sample(trilinear,
    // zeros padding, align_corners=True) at R(pixel + T) / ((n-1)/2)
    const float x = ((sx / shx + 1.f) / 2.f) * ((float)P.W - 1.f);
    const float y = ((sy / shy + 1.f) / 2.f) * ((float)P.H - 1.f);
    const float z = ((sz / shz + 1.f) / 2.f) * ((float)P.D - 1.f);
    const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
    const float wx = x - xf, wy = y - yf, wz = z - zf;
    const int x0 = (int)xf, y0 = (int)yf, z0 = (int)zf;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int cx = x0 + (c & 1), cy = y0 + ((c >> 1) & 1), cz = z0 + (c >> 2);
      if (cx < 0 || cy < 0 || cz < 0 || cx >= P.W || cy >= P.H || cz >= P.D) continue;
      const float wgt = ((c & 1) ? wx : 1.f - wx) * (((c >> 1) & 1) ? wy : 1.f - wy) * ((c >> 2) ? wz : 1.f - wz);
      const int iv = cz * Sz + cy * Sy + cx;
      float v = vol_in[iv];
      if (P.vmask) v = v * (float)P.vmask[iv];
      acc += v * wgt;
    }
    slices_io[idx] = acc;
    return;
  }
  const float s = ADJ ? slices_io[idx] : 0.f;
  float val = 0.f, weight = 0.f;
  for (int e = 0; e < np; ++e) {
    const float pv = lp[e];
    if (!(pv > 0.f)) continue;
    const float x = lo[3 * e] + sx, y = lo[3 * e + 1] + sy, z = lo[3 * e + 2] + sz;
    if (!(x > 0 && y > 0 && z > 0 && x < lx && y < ly && z < lz)) continue;
    const int iv = (int)rintf(z) * Sz + (int)rintf(y) * Sy + (int)rintf(x);
    if (ADJ) {
      unsafeAtomicAdd(vol_out + iv, pv * s);
      if (vol_weight) unsafeAtomicAdd(vol_weight + iv, pv);
    } else {
      float v = vol_in[iv];
      if (P.vmask) v = v * (float)P.vmask[iv];
      val += pv * v;
      weight += pv;
    }
  }
  if (!ADJ) {
    slices_io[idx] = weight > 1e-2f ? val / weight : val;
    if (weights_out) weights_out[idx] = weight;
  }
}

// equalize (:672-693, is_grad = false) / the fallback's `weight > 1e-2` rule (slice_acq.py:539-545)
__global__ __launch_bounds__(256) void sa_equalize_kernel(float* __restrict__ vol, const float* __restrict__ w,
                                                          const uint8_t* __restrict__ vmask, float thr, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    float v = vol[e];
    if (w) {
      const float ww = w[e];
      if (ww > thr) v = v / ww;
    }
    if (vmask) v = v * (float)vmask[e];
    vol[e] = v;
  }
}

int sa_check(const SaParams& P, size_t& lds, bool torch_mode) {
  if (!P.tr || !P.psf) return FSG_E_BADARG;
  if (P.D < 2 || P.H < 2 || P.W < 2 || P.n <= 0 || P.h <= 0 || P.w <= 0 || P.pd <= 0 || P.ph <= 0 || P.pw <= 0)
    return FSG_E_BADARG;
  if ((size_t)P.D * P.H * P.W > (size_t)0x7FFFFFFF || (size_t)P.n * P.h * P.w > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  if (P.pd > SA_MAX_AXIS || P.ph > SA_MAX_AXIS || P.pw > SA_MAX_AXIS || P.pd * P.ph * P.pw > SA_MAX_PSF) return FSG_E_TOOBIG;
  if (P.n > 65535) return FSG_E_TOOBIG;
  const size_t np = (size_t)P.pd * P.ph * P.pw;
  lds = torch_mode ? np * 4 * sizeof(float) : (np + 3 * (size_t)(P.pd + P.ph + P.pw)) * sizeof(float);
  return 0;
}

dim3 sa_grid(const SaParams& P) {
  return dim3((unsigned)((P.w + SA_TILE - 1) / SA_TILE), (unsigned)((P.h + SA_TILE - 1) / SA_TILE), (unsigned)P.n);
}

}  // namespace

extern "C" {

int fsg_slice_acq_forward_f32(const float* transforms, const float* vol, const uint8_t* vol_mask, const float* psf, int pd,
                              int ph, int pw, const uint8_t* slices_mask, float* slices, float* slices_weight, int D, int H,
                              int W, int n, int h, int w, float res_slice, int mode, void* stream) {
  if (!vol || !slices || mode < 0 || mode > 2) return FSG_E_BADARG;
  SaParams P{transforms, psf, vol_mask, slices_mask, nullptr, D, H, W, pd, ph, pw, n, h, w, res_slice};
  size_t lds = 0;
  const int rc = sa_check(P, lds, mode == FSG_SA_TORCH);
  if (rc) return rc;
  const dim3 grid = sa_grid(P), block(SA_TILE, SA_TILE);
  hipStream_t st = fsg_stream(stream);
  if (mode == FSG_SA_TORCH) {
    if (pd * ph * pw == 1 && !slices_weight)
      hipLaunchKernelGGL((sa_torch_kernel<false, true>), grid, block, lds, st, P, vol, slices, slices_weight, (float*)nullptr,
                         (float*)nullptr);
    else
      hipLaunchKernelGGL((sa_torch_kernel<false, false>), grid, block, lds, st, P, vol, slices, slices_weight, (float*)nullptr,
                         (float*)nullptr);
  } else if (mode == FSG_SA_LINEAR) {
    if (vol_mask) hipLaunchKernelGGL((sa_forward_kernel<false, true>), grid, block, lds, st, P, vol, slices, slices_weight);
    else if ((g_tuning_flags & FSG_TUNE_PRECISE_MATH) || pd * ph * pw > 4000)  // (the fast kernel keeps 16 B per tap in LDS)
      hipLaunchKernelGGL((sa_forward_kernel<false, false>), grid, block, lds, st, P, vol, slices, slices_weight);
    else {
      const unsigned tiles = grid.x * grid.y, groups = (unsigned)((n + 7) / 8);
      const size_t lds_p = (size_t)pd * ph * pw * 4 * sizeof(float) + (size_t)(pd + 1) * sizeof(int) + (size_t)6 * pd * sizeof(float) +
                           16 + (size_t)4 * SAP_CAP * sizeof(float);
      // a single tap (the mask acquisition of Scanner.scan) is a plain 2x2x2 gather per pixel: nothing for a plate to reuse
      const bool only_direct = (g_tuning_flags & FSG_TUNE_SA_FWD_DIRECT) || lds_p > 64000 || pd * ph * pw == 1;
      const bool only_plate = !only_direct && (g_tuning_flags & FSG_TUNE_SA_FWD_PLATE);
      // r03: two launches, every slice taken by exactly one of them according to its own orientation (a workgroup of the other
      // launch leaves on its first instructions): direct gathers where the slice's x axis stays near the volume's x-y plane,
      // the plate of the volume in LDS (4 x 4 pixel tiles x 4 tap phases per wave) elsewhere
      // measured crossovers (101 / 441 / 697 taps at 80 x 320^2; 73-543 taps on the stacks SimulateMotion draws at 384^3)
      const bool small_psf = pd * ph * pw < 400;
      P.fwd_wy = small_psf ? 30.f : 10.f;
      P.fwd_thr = small_psf ? 10.f / fmaxf(res_slice, 0.25f) : 1.7f;
      P.fwd_select = only_direct || only_plate ? 0 : 1;
      if (!only_plate)
        hipLaunchKernelGGL(sa_forward_linear_fast_kernel, dim3(8u * tiles * groups), block,
                           (size_t)pd * ph * pw * 4 * sizeof(float), st, P, vol, slices, slices_weight);
      P.fwd_select = only_direct || only_plate ? 0 : 2;
      if (!only_direct)
        hipLaunchKernelGGL(sa_forward_plate_kernel, dim3(8u * tiles * groups), dim3(256), lds_p, st, P, vol, slices, slices_weight);
    }
  } else {
    if (vol_mask) hipLaunchKernelGGL((sa_forward_kernel<true, true>), grid, block, lds, st, P, vol, slices, slices_weight);
    else hipLaunchKernelGGL((sa_forward_kernel<true, false>), grid, block, lds, st, P, vol, slices, slices_weight);
  }
  FSG_RETURN_LAUNCH();
}

int fsg_slice_acq_adjoint_f32(const float* transforms, const float* psf, int pd, int ph, int pw, const float* slices,
                              const uint8_t* slices_mask, const int32_t* slice_ids, const uint8_t* vol_mask, float* vol,
                              float* vol_weight, int D, int H, int W, int n, int h, int w, float res_slice, int mode,
                              void* stream) {
  if (!slices || !vol || mode < 0 || mode > 2) return FSG_E_BADARG;
  SaParams P{transforms, psf, vol_mask, slices_mask, slice_ids, D, H, W, pd, ph, pw, n, h, w, res_slice};
  size_t lds = 0;
  const int rc = sa_check(P, lds, mode == FSG_SA_TORCH);
  if (rc) return rc;
  hipStream_t st = fsg_stream(stream);
  const size_t bytes = (size_t)D * H * W * sizeof(float);
  hipError_t e = hipMemsetAsync(vol, 0, bytes, st);
  if (e != hipSuccess) return (int)e;
  if (vol_weight) {
    e = hipMemsetAsync(vol_weight, 0, bytes, st);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid = sa_grid(P), block(SA_TILE, SA_TILE);
  if (mode == FSG_SA_TORCH) {
    // the fallback multiplies by vol_mask at the very end (fsg_equalize_f32), not per contribution
    hipLaunchKernelGGL((sa_torch_kernel<true, false>), grid, block, lds, st, P, (const float*)nullptr, const_cast<float*>(slices),
                       (float*)nullptr, vol, vol_weight);
  } else if (mode == FSG_SA_LINEAR) {
    if (vol_mask) hipLaunchKernelGGL((sa_adjoint_kernel<false, true>), grid, block, lds, st, P, slices, vol, vol_weight);
    else hipLaunchKernelGGL((sa_adjoint_kernel<false, false>), grid, block, lds, st, P, slices, vol, vol_weight);
  } else if (g_tuning_flags & FSG_TUNE_SA_DIRECT) {
    if (vol_mask) hipLaunchKernelGGL((sa_adjoint_kernel<true, true>), grid, block, lds, st, P, slices, vol, vol_weight);
    else hipLaunchKernelGGL((sa_adjoint_kernel<true, false>), grid, block, lds, st, P, slices, vol, vol_weight);
  } else {
    // tile: 16x16 pixels when their footprint (pixel pitch * 15 + PSF width) stays small, else 8x8; PSF planes per
    // chunk and accumulator size from the expected cell count of a chunk's plate, E^2 * K inflated for a typical
    // oblique orientation (each workgroup still decides per chunk whether ITS plate fits; if not it scatters directly)
    const int ext = pw > ph ? pw : ph;
    const bool t16 = res_slice * 15.f + (float)ext <= (float)g_sa_t16_extent;
    const int tile = t16 ? 16 : 8;
    const float E = res_slice * (float)(tile - 1) + (float)ext + 1.f;
    auto est = [&](int z) { return 1.6f * E * E * (1.5f * (float)z + 3.f); };
    int cap = g_sa_cap, zc = g_sa_zc < pd ? g_sa_zc : pd;
    if (g_sa_auto) {
      zc = 1;
      for (int z = 4; z >= 1; --z)
        if (est(z) <= (float)g_sa_cap) { zc = z; break; }
      if (est(zc) > (float)cap) cap = (int)fminf(est(zc) * 1.25f, 12288.f);
      if (zc > pd) zc = pd;
    }
    const size_t lds2 = lds + (256 + 2 * (size_t)cap) * sizeof(float);
    const dim3 grid2((unsigned)((w + tile - 1) / tile), (unsigned)((h + tile - 1) / tile), (unsigned)n);
#define SA_LAUNCH_LDS(TT, VMM)                                                                                         \
  do {                                                                                                                \
    auto kfn = sa_adjoint_nn_lds_kernel<TT, VMM>;                                                                     \
    hipError_t ea = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);     \
    if (ea != hipSuccess) return (int)ea;                                                                             \
    hipLaunchKernelGGL(kfn, grid2, dim3(256), lds2, st, P, slices, vol, vol_weight, zc, cap);                         \
  } while (0)
    if (t16) { if (vol_mask) SA_LAUNCH_LDS(16, true); else SA_LAUNCH_LDS(16, false); }
    else { if (vol_mask) SA_LAUNCH_LDS(8, true); else SA_LAUNCH_LDS(8, false); }
#undef SA_LAUNCH_LDS
  }
  FSG_RETURN_LAUNCH();
}

int fsg_slice_acq_set_tuning(int cap_cells, int z_chunk, int t16_extent) {
  if (cap_cells < 256 || cap_cells > 18432 || z_chunk < 0 || t16_extent < 0) return FSG_E_BADARG;
  g_sa_cap = cap_cells; g_sa_zc = z_chunk; g_sa_t16_extent = t16_extent;
  g_sa_auto = z_chunk == 0;
  if (g_sa_auto) g_sa_zc = 3;
  return 0;
}

int fsg_equalize_f32(float* vol, const float* vol_weight, const uint8_t* vol_mask, float threshold, size_t n, void* stream) {
  if (!vol || n == 0 || (!vol_weight && !vol_mask)) return FSG_E_BADARG;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sa_equalize_kernel, dim3((unsigned)blocks), dim3(256), 0, fsg_stream(stream), vol, vol_weight, vol_mask,
                     threshold, n);
  FSG_RETURN_LAUNCH();
}

}  // extern "C"
