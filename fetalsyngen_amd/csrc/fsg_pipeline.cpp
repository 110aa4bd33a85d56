// fsg_pipeline.cpp -- the fused per-sample launch sequence as ONE native call.
//
// Host-side only (no kernels here): enqueues, on the caller's stream, exactly the sequence
// `FetalSynthGen.sample` of the reference walks through (generator/model.py:231-276: generate -> augment),
// using the entry points declared in include/fsg_hip.h.  It exists because at ~30 us per kernel the Python
// interpreter between launches costs as much as the kernels; semantics and results are those of calling the
// entry points one by one (tests/test_hip_parity.py::test_native_pipeline_equals_stagewise).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstddef>
#include <map>
#include <mutex>
#include <utility>
#include "../../include/fsg_hip.h"

// look-ahead launches (fsg_deform.hip, fsg_zoom.hip): not part of the public header
extern "C" int fsg_internal_floormin_rest_ride(const fsg_deform* d, int32_t* mm3, const void* drawk, unsigned draw_blocks, void* stream);


extern int g_tuning_flags;

// ---- head of sample n+1 beside the tail of sample n (fsg_sample_plan::overlap) ---------------------------------------
// Per launch stream: a side stream and two events.  `ev_free` is recorded on the launch stream at the point of a call after
// which it no longer touches ws0 / ws_rows (after the blur, or after K7 when the blur's last pass lands in ws0); the next
// call's upload + head wait for it on the side stream, the launch stream waits for `ev_head` before the margins and the warp.
// The head is VALU-bound (Philox + Box-Muller), the resampling tail it runs beside is bound by LDS and latency.
namespace {
struct HeadOverlap {
  hipStream_t side = nullptr;
  hipEvent_t ev_free = nullptr, ev_head = nullptr;
  bool has_free = false;
  uint64_t seq = 0;
  const float* ws0 = nullptr;
};
// g_ho_mu guards the map AND every field of its entries: fsg_sample_run holds it from the look-up to its last access of the
// entry whenever the overlap state is involved (two host threads launching on one (device, stream) pair serialise there;
// without the overlap the lock is taken once, briefly, to invalidate a stale entry).
std::mutex g_ho_mu;
std::map<std::pair<int, hipStream_t>, HeadOverlap> g_ho;
std::atomic<bool> g_ho_used{false};  // no entry was ever created: the default path never takes the lock

// Caller holds g_ho_mu.  `err` receives the HIP error of a failed creation (partially created objects are destroyed).
HeadOverlap* head_overlap_state(hipStream_t st, bool create, hipError_t* err) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { if (err) *err = e; return nullptr; }
  auto key = std::make_pair(dev, st);
  auto it = g_ho.find(key);
  if (it != g_ho.end()) return &it->second;
  if (!create) return nullptr;
  HeadOverlap h;
  if ((e = hipStreamCreateWithFlags(&h.side, hipStreamNonBlocking)) != hipSuccess) { if (err) *err = e; return nullptr; }
  if ((e = hipEventCreateWithFlags(&h.ev_free, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&h.ev_head, hipEventDisableTiming)) != hipSuccess) {
    if (h.ev_free) (void)hipEventDestroy(h.ev_free);
    (void)hipStreamDestroy(h.side);
    if (err) *err = e;
    return nullptr;
  }
  return &g_ho.emplace(key, h).first->second;  // std::map: the address stays valid
}
}  // namespace

// Releases the side streams and events of the head overlap (they otherwise live for the life of the process).  The caller
// must have synchronised the streams it launched samples on.
extern "C" int fsg_pipeline_teardown(void) {
  std::lock_guard<std::mutex> lk(g_ho_mu);
  int rc = 0;
  for (auto& kv : g_ho) {
    HeadOverlap& h = kv.second;
    hipError_t e;
    if (h.ev_free && (e = hipEventDestroy(h.ev_free)) != hipSuccess) rc = (int)e;
    if (h.ev_head && (e = hipEventDestroy(h.ev_head)) != hipSuccess) rc = (int)e;
    if (h.side && (e = hipStreamDestroy(h.side)) != hipSuccess) rc = (int)e;
  }
  g_ho.clear();
  return rc;
}

#define FSG_TRY(expr)        \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)
#define FSG_HIP(expr)                          \
  do {                                         \
    hipError_t e_ = (expr);                    \
    if (e_ != hipSuccess) return (int)e_;      \
  } while (0)

extern "C" int fsg_sample_run(const fsg_sample_plan* p, void* stream) {
  if (!p || !p->out || !p->ws0 || !p->ws1 || !p->mm8) return FSG_E_BADARG;
  const int n0 = p->shape[0], n1 = p->shape[1], n2 = p->shape[2];
  if (n0 <= 0 || n1 <= 0 || n2 <= 0) return FSG_E_BADARG;
  const size_t n = (size_t)n0 * n1 * n2;
  if (n > (size_t)0x7FFFFFFF) return FSG_E_TOOBIG;
  hipStream_t st = (hipStream_t)stream;
  int rode = 0;  // look-ahead jobs that really went out (fsg_sample_plan::rode)
  if (p->rode) *p->rode = 0;

  // stage trace (measurement only): an event behind every launch, see fsg_sample_plan::trace_events
  int ntrace = p->trace_start > 0 ? p->trace_start : 0;
  auto mark = [&](int id) -> int {
    if (!p->trace_events || !p->trace_ids || ntrace >= p->trace_cap) return 0;
    hipError_t e = hipEventRecord((hipEvent_t)p->trace_events[ntrace], st);
    if (e != hipSuccess) return (int)e;
    p->trace_ids[ntrace++] = id;
    p->trace_ids[p->trace_cap] = ntrace;
    return 0;
  };
  FSG_TRY(mark(p->trace_start > 0 ? p->trace_first_id : FSG_ST_BEGIN));

  const bool has_gamma = p->epi.gamma > 0.f, has_bias = p->epi.bias != nullptr;
  bool head_done = false;
  fsg_deform dh = p->deform;
  const int need_h = 3 * dh.field_dims[2] + (has_bias ? p->epi.bias_dims[2] : 0);
  const bool fused_head = p->deform_active && p->mm8_preset && p->ws_rows && !(g_tuning_flags & FSG_TUNE_SPLIT_HEAD) &&
                          need_h > 0 && need_h <= p->row_stride;
  // where the upload and the head go: the side stream when the caller asked for the overlap and the head is one launch
  HeadOverlap* ho = nullptr;
  std::unique_lock<std::mutex> ho_lock(g_ho_mu, std::defer_lock);
  const bool want_overlap = p->overlap && fused_head && p->resample_active && p->arena_host;
  if (want_overlap || g_ho_used.load(std::memory_order_acquire)) {
    ho_lock.lock();
    if (want_overlap) {
      hipError_t herr = hipSuccess;
      ho = head_overlap_state(st, true, &herr);
      if (!ho) return herr != hipSuccess ? (int)herr : FSG_E_BADARG;
      g_ho_used.store(true, std::memory_order_release);
    } else {  // this call uses the workspace in launch-stream order only: a later overlapped call must not trust an old event
      HeadOverlap* old = head_overlap_state(st, false, nullptr);
      if (old) old->has_free = false;
      ho_lock.unlock();
    }
  }
  void* hstream = stream;
  if (ho) {  // ho_lock is held until this call returns
    if (!(ho->has_free && ho->ws0 == p->ws0 && ho->seq + 1 == p->ws_seq)) {
      FSG_HIP(hipEventRecord(ho->ev_free, st));  // behind everything enqueued so far
    }
    FSG_HIP(hipStreamWaitEvent(ho->side, ho->ev_free, 0));
    ho->has_free = false;
    hstream = (void*)ho->side;
  }
  if (p->arena_host) {
    if (!p->arena_dev || (p->arena_bytes & 15)) return FSG_E_BADARG;
    FSG_TRY(fsg_copy_bytes(p->arena_dev, p->arena_host, (size_t)p->arena_bytes, hstream));
    if (!ho) FSG_TRY(mark(FSG_ST_UPLOAD));
  }
  int head_rc = 0;
  if (fused_head) {
    // K1 + per-row coarse values + six-face minimum in one launch (keys arrive initialised with the parameters)
    dh.rows = nullptr;
    dh.row_stride = 0;
    head_rc = FSG_E_ALIGN;
    if (p->label_codes && p->code_tuples && !p->gmm_noise && !(g_tuning_flags & FSG_TUNE_NO_SEED_CODES))  // 6 instead of 8 B/voxel
      head_rc = fsg_sample_head_codes_f32(p->label_codes, p->code_tuples, p->code_ntuples, p->code_stride, p->code_sel, n, p->mus,
                                          p->sigmas, p->ntab, p->gmm_seed, p->gmm_stream, p->ws0, &dh, &p->epi, p->ws_rows,
                                          p->row_stride, p->mm8, hstream);
    if (head_rc == FSG_E_ALIGN || head_rc == FSG_E_TOOBIG)  // no codes, or outside their domain: the four label volumes
      head_rc = fsg_sample_head_f32(p->label_parts[0], p->label_parts[1], p->label_parts[2], p->label_parts[3], n, p->mus,
                                    p->sigmas, p->ntab, p->gmm_noise, p->gmm_seed, p->gmm_stream, p->ws0, &dh, &p->epi,
                                    p->ws_rows, p->row_stride, p->mm8, hstream);
    if (head_rc == 0) head_done = true;
    if (head_done && !ho) FSG_TRY(mark(FSG_ST_HEAD));
  }
  if (ho) {  // whatever happened on the side stream is ordered before the rest of the sample (and before any fallback)
    FSG_HIP(hipEventRecord(ho->ev_head, ho->side));
    FSG_HIP(hipStreamWaitEvent(st, ho->ev_head, 0));
  }
  if (head_rc != 0 && head_rc != FSG_E_TOOBIG && head_rc != FSG_E_ALIGN) return head_rc;
  // K1: GMM draw -> ws0; the same launch resets every min/max key of the sample (unless they arrived initialised):
  // [min x,y,z | zoom min] [zoom max | 3 unused]
  if (!head_done)
    FSG_TRY(fsg_gmm_sample_u8x4_mm(p->label_parts[0], p->label_parts[1], p->label_parts[2], p->label_parts[3], n,
                                   p->mus, p->sigmas, p->ntab, p->gmm_noise, p->gmm_seed, p->gmm_stream, p->ws0,
                                   p->mm8_preset ? nullptr : p->mm8, 4, 4, stream));
  if (!head_done) FSG_TRY(mark(FSG_ST_GMM));
  float* cur = p->ws0;
  float* other = p->ws1;

  if (p->deform_active) {
    // K2/K3: coarse rows, floor(min) margins; K4(+K5): fused warp of the image and the labels -> ws1
    if (!p->seg_in || (!p->seg_out && !p->seg_out_u8) || (p->seg_out_u8 && !p->seg_in_u8)) return FSG_E_BADARG;
    fsg_deform d = p->deform;
    const int need = 3 * d.field_dims[2] + (has_bias ? p->epi.bias_dims[2] : 0);
    if (head_done) {
      d.rows = p->ws_rows;
      d.row_stride = p->row_stride;
      if (p->ride_draw && p->ride_draw_blocks) {  // + the next sample's draw job (fsg_sample_plan::ride_draw)
        FSG_TRY(fsg_internal_floormin_rest_ride(&d, p->mm8, p->ride_draw, p->ride_draw_blocks, stream));
        rode |= 1;
      } else {
        FSG_TRY(fsg_coords_floormin_rest_f32(&d, p->mm8, stream));
      }
      FSG_TRY(mark(FSG_ST_FLOORMIN));
    } else {
      if (p->ws_rows && need > 0 && need <= p->row_stride) {
        d.rows = nullptr;
        d.row_stride = 0;
        FSG_TRY(fsg_deform_rows_f32(&d, &p->epi, p->ws_rows, p->row_stride, stream));
        FSG_TRY(mark(FSG_ST_ROWS));
        d.rows = p->ws_rows;
        d.row_stride = p->row_stride;
      } else {
        d.rows = nullptr;
        d.row_stride = 0;
      }
      int rc = fsg_coords_floormin_f32(&d, p->mm8, stream);
      if (rc == FSG_E_TOOBIG) {  // coarse grid beyond the row kernels: exact min/max (6 keys) into a side buffer
        return FSG_E_TOOBIG;     // the Python orchestration handles this rare configuration stage by stage
      }
      FSG_TRY(rc);
      FSG_TRY(mark(FSG_ST_FLOORMIN));
    }
    int rw = FSG_E_ALIGN;
    if (p->seg_out_u8) {  // uint8 labels in and out (same values: labels are integers 0..255)
      rw = fsg_warp_f32_u8(&d, p->mm8, cur, other, p->seg_in_u8, p->seg_out_u8, &p->epi, stream);
    } else {
      if (p->seg_in_u8)  // uint8 copy of the labels: 1 B/voxel gathered instead of 4 (same output)
        rw = fsg_warp_f32_u8_to_f32(&d, p->mm8, cur, other, p->seg_in_u8, p->seg_out, &p->epi, stream);
      if (rw == FSG_E_ALIGN) rw = fsg_warp_f32(&d, p->mm8, cur, other, p->seg_in, p->seg_out, &p->epi, stream);
    }
    FSG_TRY(rw);
    FSG_TRY(mark(FSG_ST_WARP));
    float* t = cur; cur = other; other = t;
  } else {
    if (has_gamma) {
      FSG_TRY(fsg_gamma_f32(cur, n, p->epi.gamma, other, stream));
      FSG_TRY(mark(FSG_ST_POINTWISE));
      float* t = cur; cur = other; other = t;
    }
    if (has_bias) {
      FSG_TRY(fsg_bias_mul_f32(cur, n0, n1, n2, p->epi.bias, p->epi.bias_dims[0], p->epi.bias_dims[1],
                               p->epi.bias_dims[2], p->epi.bx, p->epi.by, p->epi.bz, other, stream));
      FSG_TRY(mark(FSG_ST_POINTWISE));
      float* t = cur; cur = other; other = t;
    }
  }

  if (p->resample_active) {
    if (!p->ws_low) return FSG_E_BADARG;
    const int m0 = p->low_shape[0], m1 = p->low_shape[1], m2 = p->low_shape[2];
    auto mark_free = [&]() -> int {  // from here on this call touches neither ws0 nor ws_rows
      if (!ho) return 0;
      FSG_HIP(hipEventRecord(ho->ev_free, st));
      ho->has_free = true;
      ho->seq = p->ws_seq;
      ho->ws0 = p->ws0;
      return 0;
    };
    // K6 + K7 + K8 as the fused pair (csrc/fsg_blur_rs.hip): blur and down-sampling of an axis in one operator, the blurred
    // full-resolution volume never exists.  Bracketed by the caller's HIP events like the unfused blur below.
    const bool fused_rs = fsg_blur_resample_supported(n0, n1, n2, m0, m1, m2, p->blur_ntaps[0], p->blur_ntaps[1], p->blur_ntaps[2]);
    if (fused_rs) {
      if (p->ev_blur_begin) FSG_HIP(hipEventRecord((hipEvent_t)p->ev_blur_begin, st));
      FSG_TRY(fsg_blur_resample_x_f32(cur, n0, n1, n2, p->rs_tab[0], m0, p->blur_taps[0], p->blur_ntaps[0], other, stream));
      FSG_TRY(mark(FSG_ST_BLUR_RS_X));
      FSG_TRY(fsg_blur_resample_yz_noise_f32(other, m0, n1, n2, p->rs_tab[1], p->rs_tab[2], m1, m2, p->blur_taps[1],
                                             p->blur_ntaps[1], p->blur_taps[2], p->blur_ntaps[2], p->noise_mode, p->noise,
                                             p->noise_seed, p->noise_stream, p->noise_std, p->ws_low, stream));
      FSG_TRY(mark(FSG_ST_BLUR_RS_YZ));
      if (p->ev_blur_end) FSG_HIP(hipEventRecord((hipEvent_t)p->ev_blur_end, st));
      FSG_TRY(mark_free());
    } else {
    // K6: separable blur, x then y then z (optionally bracketed by the caller's HIP events: bench.py's live
    // measurement of the graded kernel on the launch stream)
    if (p->ev_blur_begin) FSG_HIP(hipEventRecord((hipEvent_t)p->ev_blur_begin, st));
    for (int axis = 0; axis < 3; ++axis) {
      const int nt = p->blur_ntaps[axis];
      if (nt <= 0) continue;
      if (axis == 1 && p->blur_ntaps[2] > 0 && !(g_tuning_flags & FSG_TUNE_NO_BLUR_FUSE)) {
        // y and z passes in one launch (intermediate in LDS); bit-identical to the two single-axis launches
        int rf = fsg_blur_yz_taps_host_f32(cur, other, n0, n1, n2, p->blur_taps[1], nt, p->blur_taps[2], p->blur_ntaps[2],
                                           stream);
        if (rf == 0) {
          FSG_TRY(mark(FSG_ST_BLUR_YZ));
          float* t = cur; cur = other; other = t;
          break;
        }
        if (rf != FSG_E_ALIGN) return rf;
      }
      int rc = fsg_blur_axis_taps_host_f32(cur, other, n0, n1, n2, axis, p->blur_taps[axis], nt, stream);
      if (rc == FSG_E_ALIGN) return FSG_E_ALIGN;  // shape outside the tuned kernels: stage-by-stage path
      FSG_TRY(rc);
      FSG_TRY(mark(FSG_ST_BLUR_X + axis));
      float* t = cur; cur = other; other = t;
    }
    if (p->ev_blur_end) FSG_HIP(hipEventRecord((hipEvent_t)p->ev_blur_end, st));
    // K7+K8: resample + noise -> low
    const bool k7_reads_ws0 = cur == p->ws0;
    if (!k7_reads_ws0) FSG_TRY(mark_free());
    FSG_TRY(fsg_resample_noise_f32(cur, n0, n1, n2, p->rs_tab[0], p->rs_tab[1], p->rs_tab[2], p->ws_low, m0, m1, m2,
                                   p->noise_mode, p->noise, p->noise_seed, p->noise_stream, p->noise_std, stream));
    FSG_TRY(mark(FSG_ST_K7));
    if (k7_reads_ws0) FSG_TRY(mark_free());
    }
    // K9 (+K10): min/max of the zoom-back, then zoom-back + normalise -> out
    if (p->mm_slots && p->mm_nslots >= 2 && p->mm_nslots <= 64) {  // keys sharded over slots: no contended address
      FSG_TRY(fsg_zoom3d_minmax_sharded_f32(p->ws_low, m0, m1, m2, p->back_tab[0], p->back_tab[1], p->back_tab[2], n0, n1,
                                            n2, p->mm_slots, p->mm_nslots, stream));
      FSG_TRY(mark(FSG_ST_K9A));
      FSG_TRY(fsg_zoom3d_normalise_sharded_f32(p->ws_low, m0, m1, m2, p->back_tab[0], p->back_tab[1], p->back_tab[2], p->out,
                                               n0, n1, n2, p->mm_slots, p->mm_nslots, p->scale01 ? 1 : 0, stream));
      FSG_TRY(mark(FSG_ST_K9B));
      if (p->rode) *p->rode = rode;
      return 0;
    }
    FSG_TRY(fsg_zoom3d_minmax_f32(p->ws_low, m0, m1, m2, p->back_tab[0], p->back_tab[1], p->back_tab[2], n0, n1, n2,
                                  p->mm8 + 3, stream));
    FSG_TRY(mark(FSG_ST_K9A));
    FSG_TRY(fsg_zoom3d_normalise_f32(p->ws_low, m0, m1, m2, p->back_tab[0], p->back_tab[1], p->back_tab[2], p->out, n0,
                                     n1, n2, p->mm8 + 3, p->scale01 ? 1 : 0, stream));
    FSG_TRY(mark(FSG_ST_K9B));
    if (p->rode) *p->rode = rode;
    return 0;
  }

  // no resampling: noise at full resolution, optional [0,1] scaling
  if (p->noise_mode != 0) {
    float* dst = p->scale01 ? other : p->out;
    FSG_TRY(fsg_add_noise_f32(cur, n, p->noise_mode == 1 ? p->noise : nullptr, p->noise_seed, p->noise_stream,
                              p->noise_std, dst, stream));
    cur = dst;
  }
  if (p->scale01) {
    FSG_TRY(fsg_reduce_minmax_f32(cur, n, p->mm8 + 3, stream));
    FSG_TRY(fsg_scale_f32(cur, n, p->mm8 + 3, 1, p->out, stream));
  } else if (cur != p->out) {
    hipError_t e = hipMemcpyAsync(p->out, cur, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  if (p->rode) *p->rode = rode;
  return 0;
}

extern "C" int64_t fsg_sample_plan_layout(int which) {
  switch (which) {
    case 0: return (int64_t)sizeof(fsg_sample_plan);
    case 1: return (int64_t)offsetof(fsg_sample_plan, blur_taps);
    case 2: return (int64_t)offsetof(fsg_sample_plan, out);
    case 3: return (int64_t)offsetof(fsg_sample_plan, seg_in_u8);
    case 4: return (int64_t)offsetof(fsg_sample_plan, ws_seq);
    case 5: return (int64_t)offsetof(fsg_sample_plan, code_sel);
    default: return -1;
  }
}

// The plan from two flat arrays instead of ~60 field writes through an FFI (the Python mirror spends 41 us filling the
// struct field by field through ctypes: profiles/r02_c_host_phases.txt).  Index layout: FSG_PLAN_I_* / FSG_PLAN_F_* in
// include/fsg_hip.h.  `taps`: 3 x FSG_PLAN_TAPS_STRIDE floats, row a = the Gaussian taps of axis a.
extern "C" int fsg_sample_plan_pack(fsg_sample_plan* p, const int64_t* iv, int niv, const double* fv, int nfv,
                                    const float* taps) {
  if (!p || !iv || !fv || niv < FSG_PLAN_I_COUNT || nfv < FSG_PLAN_F_COUNT) return FSG_E_BADARG;
  fsg_sample_plan q = {};
  for (int a = 0; a < 3; ++a) q.shape[a] = (int32_t)iv[FSG_PLAN_I_SHAPE + a];
  for (int a = 0; a < 4; ++a) q.label_parts[a] = (const uint8_t*)(uintptr_t)iv[FSG_PLAN_I_LABEL_PARTS + a];
  q.mus = (const float*)(uintptr_t)iv[FSG_PLAN_I_MUS];
  q.sigmas = (const float*)(uintptr_t)iv[FSG_PLAN_I_SIGMAS];
  q.ntab = (int32_t)iv[FSG_PLAN_I_NTAB];
  q.gmm_noise = (const float*)(uintptr_t)iv[FSG_PLAN_I_GMM_NOISE];
  q.gmm_seed = (uint64_t)iv[FSG_PLAN_I_GMM_SEED];
  q.gmm_stream = (uint64_t)iv[FSG_PLAN_I_GMM_STREAM];
  q.deform_active = (int32_t)iv[FSG_PLAN_I_DEFORM_ACTIVE];
  if (q.deform_active) {
    for (int a = 0; a < 3; ++a) q.deform.shape[a] = q.shape[a];
    for (int a = 0; a < 9; ++a) q.deform.A[a] = (float)fv[FSG_PLAN_F_A + a];
    for (int a = 0; a < 3; ++a) {
      q.deform.centre[a] = (float)fv[FSG_PLAN_F_CENTRE + a];
      q.deform.c2[a] = (float)fv[FSG_PLAN_F_C2 + a];
      q.deform.field_dims[a] = (int32_t)iv[FSG_PLAN_I_FIELD_DIMS + a];
    }
    q.deform.flip = (int32_t)iv[FSG_PLAN_I_FLIP];
    q.deform.field = (const float*)(uintptr_t)iv[FSG_PLAN_I_FIELD];
    q.deform.tx = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_FIELD_TABS];
    q.deform.ty = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_FIELD_TABS + 1];
    q.deform.tz = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_FIELD_TABS + 2];
  }
  q.seg_in = (const float*)(uintptr_t)iv[FSG_PLAN_I_SEG_IN];
  q.seg_out = (float*)(uintptr_t)iv[FSG_PLAN_I_SEG_OUT];
  q.seg_in_u8 = (const uint8_t*)(uintptr_t)iv[FSG_PLAN_I_SEG_IN_U8];
  q.epi.gamma = (float)fv[FSG_PLAN_F_GAMMA];
  for (int a = 0; a < 3; ++a) q.epi.bias_dims[a] = (int32_t)iv[FSG_PLAN_I_BIAS_DIMS + a];
  q.epi.bias = (const float*)(uintptr_t)iv[FSG_PLAN_I_BIAS];
  q.epi.bx = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_BIAS_TABS];
  q.epi.by = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_BIAS_TABS + 1];
  q.epi.bz = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_BIAS_TABS + 2];
  q.resample_active = (int32_t)iv[FSG_PLAN_I_RESAMPLE_ACTIVE];
  for (int a = 0; a < 3; ++a) {
    q.low_shape[a] = (int32_t)iv[FSG_PLAN_I_LOW_SHAPE + a];
    q.rs_tab[a] = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_RS_TABS + a];
    q.back_tab[a] = (const fsg_tap*)(uintptr_t)iv[FSG_PLAN_I_BACK_TABS + a];
    const int nt = (int)iv[FSG_PLAN_I_BLUR_NTAPS + a];
    if (nt < 0 || nt > 129 || (nt > 0 && !taps)) return FSG_E_BADARG;
    q.blur_ntaps[a] = nt;
    for (int t = 0; t < nt; ++t) q.blur_taps[a][t] = taps[a * FSG_PLAN_TAPS_STRIDE + t];
  }
  q.noise_mode = (int32_t)iv[FSG_PLAN_I_NOISE_MODE];
  q.noise = (const float*)(uintptr_t)iv[FSG_PLAN_I_NOISE];
  q.noise_seed = (uint64_t)iv[FSG_PLAN_I_NOISE_SEED];
  q.noise_stream = (uint64_t)iv[FSG_PLAN_I_NOISE_STREAM];
  q.noise_std = (float)fv[FSG_PLAN_F_NOISE_STD];
  q.scale01 = (int32_t)iv[FSG_PLAN_I_SCALE01];
  q.ws0 = (float*)(uintptr_t)iv[FSG_PLAN_I_WS0];
  q.ws1 = (float*)(uintptr_t)iv[FSG_PLAN_I_WS1];
  q.ws_low = (float*)(uintptr_t)iv[FSG_PLAN_I_WS_LOW];
  q.ws_rows = (float*)(uintptr_t)iv[FSG_PLAN_I_WS_ROWS];
  q.row_stride = (int32_t)iv[FSG_PLAN_I_ROW_STRIDE];
  q.mm8 = (int32_t*)(uintptr_t)iv[FSG_PLAN_I_MM8];
  q.mm8_preset = (int32_t)iv[FSG_PLAN_I_MM8_PRESET];
  q.out = (float*)(uintptr_t)iv[FSG_PLAN_I_OUT];
  q.ev_blur_begin = (void*)(uintptr_t)iv[FSG_PLAN_I_EV_BEGIN];
  q.ev_blur_end = (void*)(uintptr_t)iv[FSG_PLAN_I_EV_END];
  q.mm_slots = (int32_t*)(uintptr_t)iv[FSG_PLAN_I_MM_SLOTS];
  q.mm_nslots = (int32_t)iv[FSG_PLAN_I_MM_NSLOTS];
  q.arena_host = (const void*)(uintptr_t)iv[FSG_PLAN_I_ARENA_HOST];
  q.arena_dev = (void*)(uintptr_t)iv[FSG_PLAN_I_ARENA_DEV];
  q.arena_bytes = (uint64_t)iv[FSG_PLAN_I_ARENA_BYTES];
  q.overlap = (int32_t)iv[FSG_PLAN_I_OVERLAP];
  q.ws_seq = (uint64_t)iv[FSG_PLAN_I_WS_SEQ];
  q.seg_out_u8 = (uint8_t*)(uintptr_t)iv[FSG_PLAN_I_SEG_OUT_U8];
  q.trace_events = (void**)(uintptr_t)iv[FSG_PLAN_I_TRACE_EVENTS];
  q.trace_ids = (int32_t*)(uintptr_t)iv[FSG_PLAN_I_TRACE_IDS];
  q.trace_cap = (int32_t)iv[FSG_PLAN_I_TRACE_CAP];
  *p = q;
  return 0;
}

// pack + run in one call (what the Python mirror's fast path uses per sample)
extern "C" int fsg_sample_pack_run(const int64_t* iv, int niv, const double* fv, int nfv, const float* taps, void* stream) {
  fsg_sample_plan p;
  FSG_TRY(fsg_sample_plan_pack(&p, iv, niv, fv, nfv, taps));
  return fsg_sample_run(&p, stream);
}

// B samples with one native call (B volumes per call for DataLoader-style consumers).  Plan b is enqueued on
// streams[b % nstreams]; with nstreams > 1 consecutive samples overlap their kernel tails (the caller has ordered those
// streams behind the upload of the plans' parameters and joins them afterwards).  Results are those of B fsg_sample_run
// calls: the kernels, their arguments and their order per sample are the same.
extern "C" int fsg_sample_run_batch(const fsg_sample_plan* plans, int nplans, void* const* streams, int nstreams) {
  if (!plans || nplans < 0 || !streams || nstreams < 1) return FSG_E_BADARG;
  for (int b = 0; b < nplans; ++b) FSG_TRY(fsg_sample_run(&plans[b], streams[b % nstreams]));
  return 0;
}

// Thin hipEvent helpers so a ctypes caller can time sections of fsg_sample_run on the launch stream.
extern "C" void* fsg_event_create(void) {
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
extern "C" int fsg_event_record(void* e, void* stream) {
  return e ? (int)hipEventRecord((hipEvent_t)e, (hipStream_t)stream) : FSG_E_BADARG;
}
extern "C" int fsg_event_destroy(void* e) { return e ? (int)hipEventDestroy((hipEvent_t)e) : FSG_E_BADARG; }
extern "C" int fsg_event_elapsed_ms(void* begin, void* end, float* ms) {
  if (!begin || !end || !ms) return FSG_E_BADARG;
  hipError_t rc = hipEventSynchronize((hipEvent_t)end);
  if (rc != hipSuccess) return (int)rc;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end);
}
