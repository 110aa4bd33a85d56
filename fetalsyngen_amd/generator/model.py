"""Generator orchestration: mirror of `fetalsyngen.generator.model.FetalSynthGen`
(reference model.py:27-276).

Same constructor keywords, attributes (`shape, resolution, intensity_generator, spatial_deform,
resampled, biasfield, gamma, noise, artifacts, device`), methods and return tuples, same
`synth_params` schema, same stage order and the same consumption order of the numpy / torch global
generators.  `sample()` first collects every stage's random plan on the host, uploads all small
arrays in one copy, then runs the fused kernel sequence

    gmm -> coords min/max -> warp(+gamma+bias, labels) -> blur x,y,z -> resample+noise
        -> zoom min/max -> zoom+normalise

`generate()` / `augment()` remain individually callable (stage by stage, un-fused).
The optional SR-artifact stages (`blur_cortex`, `struct_noise`, `simulate_motion`, `boundaries`; mirrors in
`fetalsyngen_amd.generator.augmentation.artifacts`) are applied after `resize_back` in the reference's order
(model.py:207-219); any callable with the reference's artifact signature is accepted.
"""
from __future__ import annotations

from typing import Iterable

import numpy as np
import torch

from .. import kernels as K
from .. import rng as _rng
from .. import tables as T
from .. import _lib
from ..utils.generation import make_affine_matrix
from .augmentation.synthseg import BiasPlan, NoisePlan, RandBiasField, RandGamma, RandNoise, RandResample, ResamplePlan
from .deformation.affine_nonrigid import DeformPlan, SpatialDeformation
from .intensity.rand_gmm import GMMPlan, ImageFromSeeds


# [+inf x4 | -inf x4] as the order-preserving int32 keys of fsg_minmax_init (csrc/fsg_common.h: fsg_f2key)
_MM8_INIT = np.array([0x7F800000] * 4 + [-2139095041] * 4, dtype=np.int32)
_MM8_INIT.setflags(write=False)
# K9's min / max keys sharded over 64 slots of 16 ints (include/fsg_hip.h: FSG_MM_SLOT_STRIDE), each {key(+inf), key(-inf), 0...}:
# the min/max pass ends every workgroup with two atomics nobody waits for (csrc/fsg_zoom.hip: zoom_mm_update)
MM_NSLOTS, MM_SLOT_STRIDE = 64, 16
_MM_SLOTS_INIT = np.zeros((MM_NSLOTS, MM_SLOT_STRIDE), dtype=np.int32)
_MM_SLOTS_INIT[:, 0], _MM_SLOTS_INIT[:, 1] = 0x7F800000, -2139095041
_MM_SLOTS_INIT.setflags(write=False)


import os as _os
import weakref

_NO_MM_SLOTS = _os.environ.get("FSG_NO_MM_SLOTS", "0") == "1"  # K9's keys as ONE pair (A/B runs)
# "0" (default): parameter upload on the launch stream before the native call.  "1": upload inside the call.  "2": upload and
# head of the sample on the library's side stream beside the previous sample's resampling tail (fsg_sample_plan::overlap).
# Measured on MI355X (profiles/r02_g_head_overlap.txt): bit-identical results, but the two cross-queue dependencies per sample
# cost more than the overlap returns (replay with the host out of the way: 255 us per sample in order, 263-283 us overlapped),
# so the in-order form stays the default.
_HEAD_OVERLAP = _os.environ.get("FSG_HEAD_OVERLAP", "0")
_ARENA_BLOCK = 1 << 16  # device block of one sample's parameters (tables._StagingRing.SLOT)
_SLOW_PLAN = _os.environ.get("FSG_SLOW_PLAN", "0") == "1"  # field-by-field ctypes plan instead of the flat arrays (cross-check)


class _Ctx:
    """One prepared sample: its plans, arena offsets and (after _resolve) device views."""


class StageTrace:
    """HIP events behind every launch of ONE sample (fsg_sample_plan::trace_events): per-launch times on the launch stream,
    measured where the launches happen.  Measurement only: each record is a barrier packet (~5 us of bubble)."""

    CAP = 16

    def __init__(self):
        import ctypes as C

        lib = _lib.load()
        self.events = (C.c_void_p * self.CAP)(*[lib.fsg_event_create() for _ in range(self.CAP)])
        self.ids = np.zeros(self.CAP + 1, dtype=np.int32)
        self.meta = None

    def slots(self):
        import ctypes as C

        return C.addressof(self.events), self.ids.ctypes.data, self.CAP

    def elapsed_us(self):
        """[(stage name, microseconds between the previous event and this launch's event)]; synchronises."""
        import ctypes as C

        lib, ms, out = _lib.load(), C.c_float(), []
        n = int(self.ids[self.CAP])
        for k in range(1, n):
            _lib.check(lib.fsg_event_elapsed_ms(self.events[k - 1], self.events[k], C.byref(ms)), "fsg_event_elapsed_ms")
            out.append((_lib.STAGE_NAMES[int(self.ids[k])], ms.value * 1e3))
        return out

    def close(self):
        lib = _lib.load()
        for e in self.events:
            if e:
                lib.fsg_event_destroy(e)
        self.events = ()


class FetalSynthGen:
    def __init__(
        self,
        shape: Iterable[int],
        resolution: Iterable[float],
        device: str,
        intensity_generator: ImageFromSeeds,
        spatial_deform: SpatialDeformation,
        resampler: RandResample,
        bias_field: RandBiasField,
        noise: RandNoise,
        gamma: RandGamma,
        blur_cortex=None,
        struct_noise=None,
        simulate_motion=None,
        boundaries=None,
        rng: str | None = None,
    ):
        if not str(device).startswith("cuda"):
            raise RuntimeError(
                f"fetalsyngen_amd.FetalSynthGen runs on an MI355X only (device='cuda:N'), got device={device!r}; "
                "there is no CPU fallback"
            )
        self.shape = shape
        self.resolution = resolution
        self.intensity_generator = intensity_generator
        self.spatial_deform = spatial_deform
        self.resampled = resampler
        self.biasfield = bias_field
        self.gamma = gamma
        self.noise = noise
        self.artifacts = {
            "blur_cortex": blur_cortex,
            "struct_noise": struct_noise,
            "simulate_motion": simulate_motion,
            "boundaries": boundaries,
        }
        self.device = device
        self.rng = rng  # None: module default (fetalsyngen_amd.rng.get_mode())
        self.native_pipeline = True  # one fsg_sample_run call per sample when the inputs allow it
        self.blur_events = None      # set to a list: (begin, end, [(axis, radius)], low-res shape) per sample, HIP events recorded
                                     # around the blur (+ down-sampling, when fused) launches inside the native call
        self.blur_events_every = 1   # ... of every k-th sample only (an event record is a barrier packet: ~5.5 us of bubble)
        self._blur_tick = 0
        self.stage_traces = None     # set to a list: every fused sample appends a StageTrace (per-launch HIP events)
        self._ws = {}

    # Everything below lives and dies with ONE process: raw host addresses (`_flat`: ivp / fvp / tbp are `ndarray.ctypes.data`
    # integers), device tensors (`_ws`, `_twins`, `_arena_next`), HIP events and streams, caches keyed by `id()` of tensors of
    # this process.  None of it is configuration, all of it is rebuilt on first use -- so a pickled generator (the reference's
    # DataLoader pattern: `num_workers=2, multiprocessing_context="spawn"`, fetalsyngen/test_dl.py:17-24, docs/datasets.md:4-6)
    # carries none of it into the worker.
    _PROCESS_LOCAL = ("_ws", "_flat", "_twins", "_seen_parts", "_arena_next", "_rs_dt", "_batch_streams", "blur_events",
                      "_blur_tick", "_keyed", "stage_traces", "_pre")

    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items() if k not in self._PROCESS_LOCAL}
        return state

    def __setstate__(self, state):
        self.__dict__.update({k: v for k, v in state.items() if k not in self._PROCESS_LOCAL})
        self.blur_events, self._blur_tick, self._ws, self.stage_traces = None, 0, {}, None

    def prewarm(self, shape=None) -> int:
        """Build and upload every per-axis table this configuration can ask for (the low-res size of
        RandResample takes at most shape*(1 - min/max resolution) distinct values per axis; the coarse
        deformation / bias grids a handful).  Optional: tables are otherwise cached on first use.
        Returns the number of tables now resident."""
        shape = tuple(int(v) for v in (shape or self.shape))
        res = np.array(self.resolution, dtype=np.float64)
        rs, sd, bf = self.resampled, self.spatial_deform, self.biasfield
        n = 0
        for a in range(3):
            size = shape[a]
            lo = int(size * res[a] / max(rs.max_resolution, res[a]))
            for m in range(max(lo - 1, 1), size + 1):
                K._device_table(T._resample_axis_table(m, size), self.device)
                f = np.float64(m) / np.float64(size)
                K._device_table(T.zoom_table(m, float(1 / f), int(np.round(m * (1 / f)))), self.device)
                n += 2
            for lo_s, hi_s in ((sd.nonlin_scale_min, sd.nonlin_scale_max), (bf.scale_min, bf.scale_max)):
                for s_ in range(max(int(np.floor(lo_s * size)) - 1, 1), int(np.ceil(hi_s * size)) + 2):
                    K._device_table(T.zoom_table(s_, float(np.float64(size) / np.float64(s_)), size), self.device)
                    n += 1
        return n

    def reserve(self, shape=None, samples_in_flight: int = 8) -> int:
        """Pre-size the device memory pool for `samples_in_flight` samples whose outputs are alive at once (the host runs
        several samples ahead of the GPU, and a consumer may hold a few): allocates and releases that many image / label
        volumes and parameter arenas through torch's caching allocator, so that the first samples do not pay for
        hipMalloc calls (0.2-1 ms each) in the middle of the launch stream.  Returns the bytes reserved."""
        shape = tuple(int(v) for v in (shape or self.shape))
        dev = torch.device(self.device)
        hold = []
        for _ in range(int(samples_in_flight)):
            hold.append(torch.empty(shape, dtype=torch.float32, device=dev))   # image
            hold.append(torch.empty(shape, dtype=torch.float32, device=dev))   # labels
            hold.append(torch.empty(1 << 16, dtype=torch.uint8, device=dev))   # parameter arena
        total = sum(t.numel() * t.element_size() for t in hold)
        del hold
        return total

    # ---- native fused path -------------------------------------------------------------------------
    def _workspace(self, shape, need_rows):
        """Per (shape, stream) scratch volumes, reused by consecutive samples on that stream.

        Eviction (a fifth key appears) and growth of the row workspace drop tensors that kernels already
        enqueued may still read.  That is safe because every block here is allocated while its key's stream is
        the current one: the caching allocator hands a freed block only to later requests on the block's OWN
        allocation stream, which are ordered behind those kernels (tests/test_hip_parity.py::
        test_workspace_eviction_across_streams)."""
        dev = torch.device(self.device)
        key = (shape, dev.index, K._stream(dev).value)
        ws = self._ws.get(key)
        n = int(np.prod(shape))
        if ws is None:
            ws = {"ws0": torch.empty(n, dtype=torch.float32, device=dev),
                  "ws1": torch.empty(n, dtype=torch.float32, device=dev),
                  "low": torch.empty(n, dtype=torch.float32, device=dev),
                  "mm8": torch.empty(8, dtype=torch.int32, device=dev), "rows": None, "stride": 0}
            if len(self._ws) >= 4:
                self._ws.pop(next(iter(self._ws)))
            self._ws[key] = ws
        if need_rows > ws["stride"]:
            ws["stride"] = (max(need_rows, 64) + 3) // 4 * 4
            ws["rows"] = torch.empty(shape[0] * shape[1] * ws["stride"], dtype=torch.float32, device=dev)
        ws["seq"] = ws.get("seq", 0) + 1  # every use of the scratch set, whatever path makes it (fsg_sample_plan::ws_seq)
        return ws

    def _fill_native_plan(self, p, c, scale01, ws, out, seg_out):
        """Fill the fsg_sample_plan `p` of prepared sample `c` (see _prepare / _resolve).  `out` / `seg_out`: where the
        image and the deformed labels go (seg_out is ignored when there is no deformation).  Returns False when the
        configuration is outside the fused kernels' domain (blur radius beyond the plan's tap capacity)."""
        import ctypes as C


        dev = torch.device(self.device)
        shape, seg, spec, rplan, nplan = c.shape, c.seg, c.spec, c.rplan, c.nplan
        p.shape[:] = shape
        for q, part in enumerate(c.label_parts):
            p.label_parts[q] = part.data_ptr()
        p.mus, p.sigmas, p.ntab = c.mus.data_ptr(), c.sigmas.data_ptr(), int(c.mus.numel())
        f = c.gmm_plan.field
        if f.host is not None:
            z = f.device_tensor(dev)
            c.keep.append(z)
            p.gmm_noise = z.data_ptr()
        else:
            p.gmm_seed, p.gmm_stream = f.seed, f.stream_id
        if spec is not None:
            p.deform_active = 1
            p.deform = spec.c
            p.seg_in = seg.data_ptr()
            # only for a caller-owned device tensor (stable identity): a converted copy would be a new cache entry per call
            twin = self._label_twin(seg) if seg is c.segmentation else None
            if twin is not None:
                p.seg_in_u8 = twin.data_ptr()
            if seg_out.dtype == torch.uint8:  # uint8 labels out: written by the warp from the uint8 source
                if twin is None:
                    return False
                p.seg_out_u8 = seg_out.data_ptr()
            else:
                p.seg_out = seg_out.data_ptr()
        p.epi = K._epilogue(c.gam, c.bias_dev, c.bias_tabs, shape)
        c.keep.append(p.epi)
        if rplan.active:
            p.resample_active = 1
            p.low_shape[:] = rplan.new_size
            for a_, (t1, t2) in enumerate(zip(c.rs_tabs.ptrs, c.back_tabs.ptrs)):
                p.rs_tab[a_], p.back_tab[a_] = t1.value, t2.value
            for a_ in range(3):
                if rplan.stds[a_] > 0:
                    taps = T.gaussian_taps(float(rplan.stds[a_]))
                    if len(taps) > 129:
                        return False
                    p.blur_ntaps[a_] = len(taps)
                    C.memmove(p.blur_taps[a_], taps.ctypes.data, taps.nbytes)
        if nplan.active:
            nf = nplan.field
            p.noise_std = nplan.std32
            if nf.host is not None:
                zn = nf.device_tensor(dev)
                c.keep.append(zn)
                p.noise_mode, p.noise = 1, zn.data_ptr()
            else:
                p.noise_mode, p.noise_seed, p.noise_stream = 2, nf.seed, nf.stream_id
        p.scale01 = int(bool(scale01))
        p.ws0, p.ws1, p.ws_low = ws["ws0"].data_ptr(), ws["ws1"].data_ptr(), ws["low"].data_ptr()
        if ws["rows"] is not None:
            p.ws_rows, p.row_stride = ws["rows"].data_ptr(), ws["stride"]
        p.mm8, p.mm8_preset = c.arena.ptr(c.mm_off), 1
        if c.slots_off is not None:
            p.mm_slots, p.mm_nslots = c.arena.ptr(c.slots_off), MM_NSLOTS
        p.out = out.data_ptr()
        return True

    label_twin_budget_bytes = 1 << 30  # HBM the uint8 twins of caller-owned label volumes may take (64 at 256^3)

    def _twin_cache(self):
        return self.__dict__.setdefault("_twins", {"by_id": {}, "bytes": 0})

    def _twin_drop(self, key):
        cache = self.__dict__.get("_twins")
        ent = cache["by_id"].pop(key, None) if cache else None
        if ent is not None and ent[2] is not None:
            cache["bytes"] -= ent[2].numel()

    def register_label_twin(self, seg, twin):
        """Hand over a uint8 copy of the float32 label volume `seg` (same values) that the caller already holds -- the
        datasets do, for every subject they cache -- so that no check and no second copy is made here."""
        if twin is None or twin.dtype != torch.uint8 or twin.shape != seg.shape or twin.device != seg.device:
            raise ValueError("label twin must be a uint8 tensor of the segmentation's shape on its device")
        cache = self._twin_cache()
        key = id(seg)
        self._twin_drop(key)
        cache["by_id"][key] = [weakref.ref(seg, lambda _r, k=key, me=weakref.ref(self): me() and me()._twin_drop(k)),
                               seg._version, twin, 2]
        cache["bytes"] += twin.numel()

    def invalidate_label_twins(self):
        """Forget every cached uint8 label twin.  The cache notices a new tensor object and an in-place torch write
        (`_version`); it cannot notice a label volume rewritten through its raw pointer (another HIP library, the `fsg_*`
        entry points themselves) -- call this after such a write."""
        self.__dict__.pop("_twins", None)
        for kc in (self.__dict__.get("_keyed") or {}).values():  # keyed mode: the subjects' pointer blocks and code volumes
            for hit in kc._subjects.values():
                bank = hit[0]()
                if bank is not None:
                    bank.__dict__.pop("_seed_codes", None)
            kc._subjects.clear()
        fb = self.__dict__.get("_flat")
        if fb is not None:
            fb["validated"].clear()

    def _label_twin(self, seg):
        """uint8 copy of a float32 label volume whose values are integers in 0..255 (dseg volumes always are): the fused warp
        then gathers 1 B/voxel instead of 4, the output stays float32.  Cached per tensor OBJECT through a weak reference
        (the entry goes when the tensor dies, nothing keeps a caller's volume alive) together with its in-place version.
        A volume is converted on its SECOND sighting: a caller who passes a fresh segmentation on every call never pays the
        synchronising integer check and the extra passes, a caller who re-uses volumes pays them once.  Twins take at most
        `label_twin_budget_bytes` (oldest dropped first).  None: no twin (yet), the warp reads the float32 volume."""
        cache = self._twin_cache()
        by_id = cache["by_id"]
        key = id(seg)
        ent = by_id.get(key)
        if ent is not None and (ent[0]() is not seg or ent[1] != seg._version):
            self._twin_drop(key)
            ent = None
        if ent is None:
            if len(by_id) > 4096:
                by_id.clear()
                cache["bytes"] = 0
            by_id[key] = [weakref.ref(seg, lambda _r, k=key, me=weakref.ref(self): me() and me()._twin_drop(k)),
                          seg._version, None, 1]
            return None
        if ent[3] == 1:  # second sighting: check and convert (synchronises once per volume)
            ent[3] = 2
            ok = bool(torch.equal(seg.round(), seg)) and float(seg.min()) >= 0 and float(seg.max()) <= 255
            if ok:
                while cache["bytes"] + seg.numel() > self.label_twin_budget_bytes:
                    victim = next((k for k, e in by_id.items() if e[2] is not None), None)
                    if victim is None:
                        break
                    self._twin_drop(victim)
                if cache["bytes"] + seg.numel() <= self.label_twin_budget_bytes:
                    ent[2] = seg.to(torch.uint8)
                    cache["bytes"] += seg.numel()
        return ent[2]

    def _native_ok(self, c, labels_u8: bool = False) -> bool:
        """labels_u8: the caller wants uint8 labels -- the fused path then writes them itself and a caller-supplied uint8
        copy of the segmentation (the stage-by-stage path's way to uint8 labels) does not keep the sample off it."""
        return (self.native_pipeline and c.label_parts is not None and c.image is None and not c.has_art
                and (c.segmentation_u8 is None or labels_u8))

    def _native_operands(self, c):
        """Shape / dtype / device checks of everything the C side only sees as pointers (a mismatched volume would make
        the fused kernels gather outside a smaller buffer), and the float32 device segmentation."""
        dev = torch.device(self.device)
        seg = c.segmentation.to(dev)
        if seg.dtype != torch.float32:
            seg = seg.float()
        seg = seg.contiguous()
        shape = tuple(int(v) for v in c.shape)
        if tuple(seg.shape) != shape:
            raise ValueError(f"segmentation shape {tuple(seg.shape)} differs from the seed volumes' shape {shape}")
        if not 1 <= len(c.label_parts) <= 4:
            raise ValueError(f"{len(c.label_parts)} seed label volumes: the fused path takes 1..4")
        for q, part in enumerate(c.label_parts):
            off_dev = part.device.type != dev.type or (dev.index is not None and part.device.index != dev.index)
            if tuple(part.shape) != shape or part.dtype != torch.uint8 or not part.is_contiguous() or off_dev:
                raise ValueError(
                    f"seed label volume {q}: expected a contiguous uint8 tensor of shape {shape} on {dev}, got "
                    f"{part.dtype} {tuple(part.shape)} on {part.device} (contiguous={part.is_contiguous()})")
        if c.mus.numel() != c.sigmas.numel() or not 1 <= c.mus.numel() <= 256:
            raise ValueError(f"mus / sigmas tables of {c.mus.numel()} / {c.sigmas.numel()} entries (need equal, 1..256)")
        c.shape, c.seg = shape, seg
        c.keep.append(seg)

    def _rows_needed(self, c) -> int:
        f2 = int(c.spec.c.field_dims[2]) if c.spec is not None else 0
        b2 = int(c.bias_dev.shape[2]) if c.bias_dev is not None else 0
        return 3 * f2 + b2 if c.spec is not None else 0

    # ---- fast form of _resolve + _native_operands + _fill_native_plan + fsg_sample_run ---------------------------------
    # Same plan, built as two flat arrays and handed over with ONE native call (fsg_sample_pack_run): filling the ctypes
    # struct field by field cost 41 us per sample, the spec / view objects 28 us, the operand checks 16 us
    # (profiles/r02_c_host_phases.txt).  tests/test_hip_parity.py::test_fast_plan_equals_field_by_field_plan compares the two
    # plans byte for byte.  FSG_SLOW_PLAN=1 forces the field-by-field path.
    _I = dict(SHAPE=0, LABEL_PARTS=3, MUS=7, SIGMAS=8, NTAB=9, GMM_NOISE=10, GMM_SEED=11, GMM_STREAM=12, DEFORM_ACTIVE=13,
              FLIP=14, FIELD_DIMS=15, FIELD=18, FIELD_TABS=19, SEG_IN=22, SEG_OUT=23, SEG_IN_U8=24, BIAS_DIMS=25, BIAS=28,
              BIAS_TABS=29, RESAMPLE_ACTIVE=32, LOW_SHAPE=33, RS_TABS=36, BACK_TABS=39, BLUR_NTAPS=42, NOISE_MODE=45, NOISE=46,
              NOISE_SEED=47, NOISE_STREAM=48, SCALE01=49, WS0=50, WS1=51, WS_LOW=52, WS_ROWS=53, ROW_STRIDE=54, MM8=55,
              MM8_PRESET=56, OUT=57, EV_BEGIN=58, EV_END=59, MM_SLOTS=60, MM_NSLOTS=61, ARENA_HOST=62, ARENA_DEV=63,
              ARENA_BYTES=64, OVERLAP=65, WS_SEQ=66, SEG_OUT_U8=67, TRACE_EVENTS=68, TRACE_IDS=69, TRACE_CAP=70, COUNT=71)
    _TAPS_STRIDE = 132

    def _flat_buffers(self):
        fb = self.__dict__.get("_flat")
        if fb is None:
            iv = np.zeros(self._I["COUNT"], dtype=np.int64)
            fv = np.zeros(17, dtype=np.float64)
            tb = np.zeros((3, self._TAPS_STRIDE), dtype=np.float32)
            centre = (np.array(self.spatial_deform.size) - 1) / 2
            fb = self._flat = dict(iv=iv, fv=fv, tb=tb, ivp=iv.ctypes.data, fvp=fv.ctypes.data, tbp=tb.ctypes.data,
                                   centre=np.asarray(centre, dtype=np.float32).tolist(), validated={})
        return fb

    def _flat_plan(self, c, scale01, out, seg_out, ws, events=None):
        """The two flat arrays of prepared sample `c` (uploaded arena, no _resolve needed).  Returns False when the
        sample is outside the fused path's domain (blur radius beyond the tap capacity)."""
        I = self._I
        fb = self._flat_buffers()
        iv, fv, tb = [0] * I["COUNT"], [0.0] * 17, fb["tb"]
        arena, base = c.arena, c.arena.base
        shape = c.shape
        iv[0:3] = shape
        for q, part in enumerate(c.label_parts):
            iv[I["LABEL_PARTS"] + q] = part.data_ptr()
        iv[I["MUS"]], iv[I["SIGMAS"]], iv[I["NTAB"]] = base + c.gm_off[0], base + c.gm_off[1], c.gm_off[2]
        f = c.gmm_plan.field
        if f.host is not None:
            z = f.device_tensor(torch.device(self.device))
            c.keep.append(z)
            iv[I["GMM_NOISE"]] = z.data_ptr()
        else:
            iv[I["GMM_SEED"]], iv[I["GMM_STREAM"]] = f.seed, f.stream_id
        dplan = c.dplan
        if dplan.active:
            iv[I["DEFORM_ACTIVE"]] = 1
            iv[I["FLIP"]] = int(bool(dplan.flip))
            a_np, c2_np = dplan.A_np, dplan.c2_np  # left by _draw_all_fast (no torch round trip)
            fv[0:9] = a_np.ravel().tolist() if a_np is not None else dplan.A.reshape(-1).tolist()
            fv[9:12] = fb["centre"]
            fv[12:15] = c2_np.astype(np.float32).tolist() if c2_np is not None else dplan.c2.to(torch.float32).tolist()
            if c.sb.pending is not None:
                off, fshape = c.sb.pending
                iv[I["FIELD_DIMS"]:I["FIELD_DIMS"] + 3] = fshape[:3]
                iv[I["FIELD"]] = base + off
                iv[I["FIELD_TABS"]:I["FIELD_TABS"] + 3] = c.sb.tabs.ptrs_i
            twin = self._label_twin(c.seg) if c.seg is c.segmentation else None
            if twin is not None:
                iv[I["SEG_IN_U8"]] = twin.data_ptr()
            iv[I["SEG_IN"]] = c.seg.data_ptr()
            if seg_out.dtype == torch.uint8:  # uint8 labels out (device-resident hand-over): needs the uint8 source
                if twin is None:
                    return False
                iv[I["SEG_OUT_U8"]] = seg_out.data_ptr()
            else:
                iv[I["SEG_OUT"]] = seg_out.data_ptr()
        if c.g is not None:
            fv[15] = float(np.float32(float(c.g)))
        if c.bplan.active:
            iv[I["BIAS_DIMS"]:I["BIAS_DIMS"] + 3] = c.bplan.grid.shape
            iv[I["BIAS"]] = base + c.bias_off
            iv[I["BIAS_TABS"]:I["BIAS_TABS"] + 3] = c.bias_tabs.ptrs_i
        rplan, nplan = c.rplan, c.nplan
        if rplan.active:
            iv[I["RESAMPLE_ACTIVE"]] = 1
            iv[I["LOW_SHAPE"]:I["LOW_SHAPE"] + 3] = rplan.new_size
            iv[I["RS_TABS"]:I["RS_TABS"] + 3] = c.rs_tabs.ptrs_i
            iv[I["BACK_TABS"]:I["BACK_TABS"] + 3] = c.back_tabs.ptrs_i
            for a_ in range(3):
                if rplan.stds[a_] > 0:
                    taps = T.gaussian_taps(float(rplan.stds[a_]))
                    n = len(taps)
                    if n > 129:
                        return False
                    iv[I["BLUR_NTAPS"] + a_] = n
                    tb[a_, :n] = taps
        if nplan.active:
            nf = nplan.field
            fv[16] = nplan.std32
            if nf.host is not None:
                zn = nf.device_tensor(torch.device(self.device))
                c.keep.append(zn)
                iv[I["NOISE_MODE"]], iv[I["NOISE"]] = 1, zn.data_ptr()
            else:
                iv[I["NOISE_MODE"]], iv[I["NOISE_SEED"]], iv[I["NOISE_STREAM"]] = 2, nf.seed, nf.stream_id
        iv[I["SCALE01"]] = int(bool(scale01))
        iv[I["WS0"]], iv[I["WS1"]], iv[I["WS_LOW"]] = ws["ws0"].data_ptr(), ws["ws1"].data_ptr(), ws["low"].data_ptr()
        if ws["rows"] is not None:
            iv[I["WS_ROWS"]], iv[I["ROW_STRIDE"]] = ws["rows"].data_ptr(), ws["stride"]
        iv[I["MM8"]], iv[I["MM8_PRESET"]] = base + c.mm_off, 1
        if c.slots_off is not None:
            iv[I["MM_SLOTS"]], iv[I["MM_NSLOTS"]] = base + c.slots_off, MM_NSLOTS
        iv[I["OUT"]] = out.data_ptr()
        if events is not None:
            iv[I["EV_BEGIN"]], iv[I["EV_END"]] = events
        if self.stage_traces is not None:
            tr = StageTrace()
            tr.meta = {"shape": tuple(shape), "low_shape": tuple(rplan.new_size) if rplan.active else None,
                       "blur_ntaps": [int(iv[I["BLUR_NTAPS"] + a_]) for a_ in range(3)]}
            iv[I["TRACE_EVENTS"]], iv[I["TRACE_IDS"]], iv[I["TRACE_CAP"]] = tr.slots()
            self.stage_traces.append(tr)
        if arena.pending is not None:  # staged, not yet copied: the call uploads (Arena.stage)
            iv[I["ARENA_HOST"]], iv[I["ARENA_DEV"]], iv[I["ARENA_BYTES"]] = arena.pending[2], base, arena.pending[3]
            iv[I["OVERLAP"]], iv[I["WS_SEQ"]] = int(self._overlap_ok(c)), ws["seq"]
        fb["iv"][:] = iv
        fb["fv"][:] = fv
        return True

    def _overlap_ok(self, c) -> bool:
        """May the head of this sample run on the library's side stream, ordered only behind the previous sample's blur
        (fsg_sample_plan::overlap)?  Only if everything the head reads besides the arena was handed to the device before this
        sample's host phase: seed volumes this generator has already used (same tensor object, same in-place version),
        tap tables built in an earlier epoch, device Philox noise (a host noise field is uploaded per sample)."""
        if _HEAD_OVERLAP != "2" or not c.arena_early or c.gmm_plan is None or c.gmm_plan.field.host is not None:
            return False
        if not (c.dplan.active and c.rplan.active and c.sb.pending is not None):
            return False
        epoch = K._EPOCH[0]
        if c.sb.tabs.born >= epoch or (c.bplan.active and c.bias_tabs.born >= epoch):
            return False
        seen, ok = self.__dict__.setdefault("_seen_parts", {}), True
        for part in c.label_parts:
            key = part.data_ptr()
            hit = seen.get(key)
            if hit is None or hit[0]() is not part or hit[1] != part._version:
                if len(seen) > 1024:
                    seen.clear()
                seen[key] = (weakref.ref(part), part._version, epoch)  # first use (or rewritten in place): from the next sample on
                ok = False
            elif hit[2] >= epoch:
                ok = False
        return ok

    def _fast_operands(self, c) -> bool:
        """The operand checks of _native_operands, once per distinct (segmentation, seed volumes) set; False when the
        segmentation needs a conversion (host tensor, other dtype): the field-by-field path handles that."""
        seg = c.segmentation
        dev = torch.device(self.device)
        if not (torch.is_tensor(seg) and seg.is_cuda and seg.dtype == torch.float32 and seg.is_contiguous()):
            return False
        fb = self._flat_buffers()
        # Every tensor is checked once, as an object (id + weak reference: a new tensor at a recycled address is a new object)
        # together with the sample shape it was checked against.  Per TENSOR, not per combination: the seed volumes of a subject
        # combine in up to 6^4 ways, and a per-combination cache missed on most samples of a run.
        val = fb["validated"]
        shape = c.shape = tuple(int(v) for v in c.shape)
        known = True
        for t_ in (seg, *c.label_parts):
            hit = val.get(id(t_))
            if hit is None or hit[0]() is not t_ or hit[1] != shape:
                known = False
                break
        if not known:
            c.mus, c.sigmas = (c.arena.f32(c.gm_off[0], (c.gm_off[2],)), c.arena.f32(c.gm_off[1], (c.gm_off[2],)))
            self._native_operands(c)  # raises on a mismatch
            if c.seg is not seg:
                return False
            if len(val) > 4096:
                val.clear()
            for t_ in (seg, *c.label_parts):
                val[id(t_)] = (weakref.ref(t_), shape)
        c.seg = seg
        if c.gm_off[2] > 256 or c.gm_off[2] < 1:
            raise ValueError(f"mus / sigmas tables of {c.gm_off[2]} entries (need 1..256)")
        return True

    def _run_native_fast(self, c, scale01, labels_u8=False):
        """Prepared sample -> (image, labels) through fsg_sample_pack_run, or None (caller falls back).  labels_u8: the labels
        as a uint8 volume (written by the warp itself; without a deformation the cached uint8 copy of the input)."""

        if not self._fast_operands(c):
            return None
        dev = torch.device(self.device)
        f2 = int(c.sb.pending[1][2]) if (c.dplan.active and c.sb.pending is not None) else 0
        b2 = int(c.bplan.grid.shape[2]) if c.bplan.active else 0
        ws = self._workspace(c.shape, 3 * f2 + b2 if c.dplan.active else 0)
        out = torch.empty(c.shape, dtype=torch.float32, device=dev)
        if labels_u8:
            if c.dplan.active:
                seg_out = torch.empty(c.shape, dtype=torch.uint8, device=dev)
            else:
                seg_out = self._label_twin(c.seg) if c.seg is c.segmentation else None
                if seg_out is None:
                    return None
        else:
            seg_out = torch.empty_like(c.seg) if c.dplan.active else c.seg
        events = None
        lib = _lib.load()
        if self.blur_events is not None and c.rplan.active:
            self._blur_tick += 1
            if self._blur_tick % self.blur_events_every == 0:
                events = (lib.fsg_event_create(), lib.fsg_event_create())
        if not self._flat_plan(c, scale01, out, seg_out, ws, events):
            return None
        if events is not None:
            fbiv = self._flat["iv"]
            nt = [int(fbiv[self._I["BLUR_NTAPS"] + a_]) for a_ in range(3)]
            self.blur_events.append((events[0], events[1], [(a_, nt[a_] // 2) for a_ in range(3) if nt[a_]],
                                     tuple(int(v) for v in c.rplan.new_size)))
        fb = self._flat
        rc = lib.fsg_sample_pack_run(fb["ivp"], self._I["COUNT"], fb["fvp"], 17, fb["tbp"], K._stream(dev))
        if rc in (_lib.E_ALIGN, _lib.E_TOOBIG):
            if c.arena.pending is not None:
                c.arena.flush(False)  # outside the fused domain: the parameters (keys re-initialised) for the fallback path
            return None
        if c.arena.pending is not None:
            c.arena.flush(rc == 0)
        _lib.check(rc, "fsg_sample_pack_run")
        f32_view = c.arena.f32
        c.mus, c.sigmas = f32_view(c.gm_off[0], (c.gm_off[2],)), f32_view(c.gm_off[1], (c.gm_off[2],))
        c.seed_intensities = {"mus": c.mus, "sigmas": c.sigmas}
        return out, seg_out

    def _run_native(self, c, scale01):
        """Enqueue prepared sample `c` with one fsg_sample_run call.  Returns (image, labels), or None when the
        configuration is outside the fused kernels' domain (the caller then launches stage by stage)."""
        import ctypes as C


        dev = torch.device(self.device)
        self._native_operands(c)
        ws = self._workspace(c.shape, self._rows_needed(c))
        out = torch.empty(c.shape, dtype=torch.float32, device=dev)
        seg_out = torch.empty_like(c.seg) if c.spec is not None else c.seg
        p = _lib.SamplePlan()
        if not self._fill_native_plan(p, c, scale01, ws, out, seg_out):
            return None
        if self.blur_events is not None and c.rplan.active:  # (begin, end, n_passes) appended per sample
            lib = _lib.load()
            e0, e1 = lib.fsg_event_create(), lib.fsg_event_create()
            p.ev_blur_begin, p.ev_blur_end = e0, e1
            self.blur_events.append((e0, e1, [(a_, int(p.blur_ntaps[a_]) // 2) for a_ in range(3) if p.blur_ntaps[a_]],
                                     tuple(int(v) for v in p.low_shape)))
        rc = _lib.load().fsg_sample_run(C.byref(p), K._stream(dev))
        if rc in (_lib.E_ALIGN, _lib.E_TOOBIG):
            return None
        _lib.check(rc, "fsg_sample_run")
        return out, seg_out

    @staticmethod
    def _params(selected_seeds, seed_intensities, dplan, g, bplan, rplan, nplan, artifacts):
        return {
            "selected_seeds": selected_seeds,
            "seed_intensities": seed_intensities,
            "deform_params": dplan.params,
            "gamma_params": {"gamma": g},
            "bf_params": bplan.params,
            "resample_params": {"spacing": rplan.spacing.tolist() if rplan.active else None},
            "noise_params": {"noise_std": nplan.std32 if nplan.active else None},
            "artifacts": artifacts,
        }

    def _validated_genparams(self, d):
        if not isinstance(d, dict):
            return d
        return {k: self._validated_genparams(v) for k, v in d.items() if v is not None}

    # ---- stage-by-stage API ---------------------------------------------------------------------
    def _intensity_prior(self, image):
        img = image.to(self.device).float().contiguous()
        return K.scale(img, K.reduce_minmax(img), mode=2)  # (x-min)/(max-min)*255, ref :138

    def generate(self, image, segmentation, seeds, genparams: dict = {}):
        with _rng.use(self.rng):
            if seeds is not None:
                seeds, selected_seeds = self.intensity_generator.load_seeds(
                    seeds=seeds, genparams=genparams.get("selected_seeds", {}))
                output, seed_intensities = self.intensity_generator.sample_intensities(
                    seeds=seeds, device=self.device, genparams=genparams.get("seed_intensities", {}))
            else:
                if image is None:
                    raise ValueError(
                        "If no seeds are passed, an image must be loaded to be used as intensity prior!")
                output = self._intensity_prior(image)
                selected_seeds, seed_intensities = {}, {}
            segmentation = segmentation.to(self.device)
            image = image.to(self.device) if image is not None else None
            image, segmentation, output, deform_params = self.spatial_deform.deform(
                image=image, segmentation=segmentation, output=output,
                genparams=genparams.get("deform_params", {}))
        return output, segmentation, image, {
            "selected_seeds": selected_seeds,
            "seed_intensities": seed_intensities,
            "deform_params": deform_params,
        }

    def _apply_artifacts(self, output, segmentation, genparams):
        artifacts = {}
        for name, artifact in self.artifacts.items():
            if artifact is not None:
                output, metadata = artifact(output, segmentation, self.device, genparams.get("artifact_params", {}),
                                            resolution=self.resolution)
                artifacts[name] = metadata
        return output, artifacts

    def augment(self, image, segmentation, genparams: dict = {}):
        with _rng.use(self.rng):
            output, gamma_params = self.gamma(image, self.device, genparams=genparams.get("gamma_params", {}))
            output, bf_params = self.biasfield(output, self.device, genparams=genparams.get("bf_params", {}))
            output, factors, resample_params = self.resampled(
                output, np.array(self.resolution), self.device, genparams=genparams.get("resample_params", {}))
            output, noise_params = self.noise(output, self.device, genparams=genparams.get("noise_params", {}))
            output = self.resampled.resize_back(output, factors)
            output, artifacts = self._apply_artifacts(output, segmentation, genparams)
        return output, {
            "gamma_params": gamma_params,
            "bf_params": bf_params,
            "resample_params": resample_params,
            "noise_params": noise_params,
            "artifacts": artifacts,
        }

    # ---- fused path -----------------------------------------------------------------------------
    def sample(self, image, segmentation, seeds, genparams: dict = {}, key: int | None = None):
        """`key`: keyed mode only (`rng="keyed"`), the sample's 64-bit key."""
        out, seg, img, params = self._pipeline(image, segmentation, seeds, genparams, scale01=False, key=key)
        return out, seg, img, params

    def _draw_plans(self, shape, genparams):
        """Host draws of everything after the intensity plan, in the reference's order:
        deformation, gamma, bias field, resampling, noise.  No device work."""
        dplan = self.spatial_deform.plan(shape, random_shift=True, genparams=genparams.get("deform_params", {}))
        g = self.gamma.plan(genparams.get("gamma_params", {}))
        bplan = self.biasfield.plan(shape, genparams.get("bf_params", {}))
        res = self.__dict__.get("_res64")
        if res is None:
            res = self._res64 = np.array(self.resolution)
        rplan = self.resampled.plan(shape, res, genparams.get("resample_params", {}))
        low_shape = rplan.new_size if rplan.active else shape
        nplan = self.noise.plan(low_shape, genparams.get("noise_params", {}))
        return dplan, g, bplan, rplan, nplan

    def plan_only(self, shape, genparams: dict = {}, fast: bool | None = None):
        """Every host draw of one seeds-based sample, nothing enqueued (bench.py --dry-plan, host profiling).
        fast=None: the bulk-draw form when there are no genparams (what `_prepare` uses); False: the per-stage plan()s."""
        with _rng.use(self.rng):
            if (fast is None and not genparams and not _SLOW_PLAN) or fast:
                return self._draw_all_fast(tuple(shape))
            m2s = self.intensity_generator.draw_subclusters(genparams.get("selected_seeds", {}))
            gmm_plan = self.intensity_generator.plan_intensities(tuple(shape), genparams.get("seed_intensities", {}))
            return (m2s, gmm_plan) + self._draw_plans(tuple(shape), genparams)

    def _draw_all_fast(self, shape):
        """All host draws of a seeds-based sample WITHOUT genparams, same values and same generator states as
        `draw_subclusters` + `plan_intensities` + `_draw_plans` (tests/test_host_plans.py::test_bulk_draws_equal_per_stage_plans),
        with fewer interpreter round trips: numpy's legacy generator hands out the same doubles whether they are asked for one
        `rand()` at a time or as `random_sample(n)`, and `randint(lo, hi, size=4)` equals four scalar calls, so the draws
        between two gates are fetched in one call (gates still short-circuit: nothing behind a failed gate is drawn);
        `torch.rand(2n)` equals two `torch.rand(n)`.  Scalars are combined as Python floats (the same IEEE doubles as the
        0-d numpy arithmetic of the per-stage code)."""

        ig, sd, bf, rs_, nz, gm = (self.intensity_generator, self.spatial_deform, self.biasfield, self.resampled, self.noise,
                                   self.gamma)
        rs = np.random.random_sample
        # ---- seeds + intensities (rand_gmm.py:81-87, :120-148)
        picks = np.random.randint(ig.min_subclusters, ig.max_subclusters + 1, size=ig.meta_labels).tolist()
        m2s = {m + 1: picks[m] for m in range(ig.meta_labels)}
        nlabels = max(ig.seed_labels) + 1
        u2 = torch.rand(2 * nlabels, dtype=torch.float32).numpy()
        mus = np.float32(25) + np.float32(200) * u2[:nlabels]
        sigmas = np.float32(5) + np.float32(20) * u2[nlabels:]
        if ig.generation_classes != ig.seed_labels:
            if ig._idx is None:
                ig._idx = (np.asarray(ig.generation_classes), np.asarray(ig.seed_labels))
            z = torch.randn(len(ig.seed_labels), dtype=torch.float32).numpy()
            tied = mus[ig._idx[0]] + np.float32(25) * z
            mus[ig._idx[1]] = np.minimum(np.maximum(tied, np.float32(0)), np.float32(225))
        gmm_plan = GMMPlan(torch.from_numpy(mus), torch.from_numpy(sigmas), _rng.normal_field(shape, stream_id=1))
        # ---- deformation (affine_nonrigid.py:140-145, :248-263, :284, :303-318)
        dplan = DeformPlan()
        if rs(1)[0] < sd.prob:
            nl = bool(sd.nonlinear_transform)
            u = rs(13 if nl else 11)  # flip, rot x3, shear x3, scale x3 [, nonlin scale, nonlin std], then the gamma gate
            dplan.active = True
            dplan.flip = bool(u[0] < sd.flip_prb)
            shp, centre32, room64 = sd._shape_constants(tuple(shape)[0:3])
            ul = u.tolist()  # Python floats: the same IEEE doubles, a tenth of the cost of 0-d / 3-element numpy arithmetic
            mr2, mr, ms2, ms, mc2, mc, pi = 2 * sd.max_rotation, sd.max_rotation, 2 * sd.max_shear, sd.max_shear, 2 * sd.max_scaling, sd.max_scaling, np.pi
            rot = np.array([(mr2 * ul[1] - mr) / 180.0 * pi, (mr2 * ul[2] - mr) / 180.0 * pi, (mr2 * ul[3] - mr) / 180.0 * pi])
            shr = np.array([ms2 * ul[4] - ms, ms2 * ul[5] - ms, ms2 * ul[6] - ms])
            scl = np.array([1 + (mc2 * ul[7] - mc), 1 + (mc2 * ul[8] - mc), 1 + (mc2 * ul[9] - mc)])
            dplan.A_np = make_affine_matrix(rot, shr, scl).astype(np.float32)
            dplan.A = torch.from_numpy(dplan.A_np)
            ut = torch.rand(3, dtype=torch.float64).tolist()  # float64 draw, always consumed
            c32, r64 = sd._shape_lists(tuple(shape)[0:3])
            centre = np.array([c32[0] + (2 * (r64[0] * ut[0]) - r64[0]), c32[1] + (2 * (r64[1] * ut[1]) - r64[1]),
                               c32[2] + (2 * (r64[2] * ut[2]) - r64[2])])
            dplan.c2_np = centre
            dplan.c2 = torch.from_numpy(centre)
            nr_params = {}
            if nl:
                scale = sd.nonlin_scale_min + u[10:11] * (sd.nonlin_scale_max - sd.nonlin_scale_min)
                sc = float(scale[0])
                small = [int(round(sc * float(shp[0]))), int(round(sc * float(shp[1]))), int(round(sc * float(shp[2])))]
                std = sd.nonlin_std_max * ul[11]
                dplan.field_small = std * torch.randn([*small, 3], dtype=torch.float32)
                nr_params = {"nonlin_scale": scale, "nonlin_std": std, "size_F_small": small}
            dplan.params = {"affine": {"rotations": rot, "shears": shr, "scalings": scl}, "non_rigid": nr_params,
                            "flip": dplan.flip}
            gate_gamma = u[-1]
        else:
            gate_gamma = rs(1)[0]
        # ---- gamma (synthseg.py:263-268)
        g = np.exp(gm.gamma_std * np.random.randn(1)[0]) if gate_gamma < gm.prob else None
        # ---- bias field (synthseg.py:157-176), then the resampling gate
        bplan = BiasPlan()
        if rs(1)[0] < bf.prob:
            u = rs(3)
            bplan.active = True
            scale = bf.scale_min + u[0:1] * (bf.scale_max - bf.scale_min)
            sc = float(scale[0])
            size = [max(int(round(sc * float(shape[0]))), 1), max(int(round(sc * float(shape[1]))), 1),
                    max(int(round(sc * float(shape[2]))), 1)]
            std = bf.std_min + (bf.std_max - bf.std_min) * u[1:2]
            std32 = np.asarray(std, dtype=np.float32)
            bplan.grid = torch.from_numpy(std32 * torch.randn(size, dtype=torch.float32).numpy())
            bplan.params = {"bf_scale": scale, "bf_std": std, "bf_size": size}
            gate_res = u[2]
        else:
            gate_res = rs(1)[0]
        # ---- resampling (synthseg.py:63-80), then the noise gate
        rplan = ResamplePlan()
        res = self.__dict__.get("_res64")
        if res is None:
            res = self._res64 = np.array(self.resolution)
        if gate_res < rs_.prob:
            u = rs(3)
            rplan.active = True
            # np.random.uniform(lo, hi) == lo + (hi - lo) * random_sample()
            spacing = np.array([1.0, 1.0, 1.0]) * (rs_.min_resolution + (rs_.max_resolution - rs_.min_resolution) * float(u[0]))
            rplan.spacing = spacing
            rplan.stds, rplan.new_size, rplan.factors, rplan.tabs = T.resample_plan(tuple(shape), res, spacing, float(u[1]))
            gate_noise = u[2]
        else:
            gate_noise = rs(1)[0]
        # ---- noise (synthseg.py:218-232)
        nplan = NoisePlan()
        if gate_noise < nz.prob:
            nplan.active = True
            std = nz.std_min + (nz.std_max - nz.std_min) * rs(1)
            nplan.std32 = float(np.asarray(std, dtype=np.float32).reshape(-1)[0])
            nplan.field = _rng.normal_field(tuple(rplan.new_size if rplan.active else shape), stream_id=2)
        return m2s, gmm_plan, dplan, g, bplan, rplan, nplan

    # A sample goes through three host phases so that B samples can share one parameter upload and one native call:
    #   _prepare : every random draw, in the reference's order, and the small arrays added to the arena   (no device work)
    #   arena.upload
    #   _resolve : device views of the uploaded arrays
    # then either one fsg_sample_run / fsg_sample_run_batch call, or the stage-by-stage launches (_run_stagewise).
    def _prepare(self, image, segmentation, seeds, genparams, arena, segmentation_u8=None):
        if genparams:
            genparams = self._validated_genparams(genparams)
        dev = self.device
        ig, sd = self.intensity_generator, self.spatial_deform
        c = _Ctx()
        c.arena, c.image, c.segmentation, c.segmentation_u8, c.genparams = arena, image, segmentation, segmentation_u8, genparams
        c.labels, c.label_parts, c.gmm_plan, c.selected_seeds = None, None, None, {}
        drawn = False
        if seeds is not None:
            gs = genparams.get("selected_seeds", {})
            if hasattr(seeds, "parts") and ig.meta_labels <= 4 and not genparams and not _SLOW_PLAN:
                # the common case (device-resident SeedBank, nothing fixed by the caller): every draw of the sample in bulk
                shape = tuple(seeds.shape)
                m2s, c.gmm_plan, c.dplan, c.g, c.bplan, c.rplan, c.nplan = self._draw_all_fast(shape)
                c.label_parts, c.selected_seeds = seeds.parts(m2s), {"mlabel2subclusters": m2s}
                drawn = True
            elif hasattr(seeds, "parts") and ig.meta_labels <= 4:  # device-resident SeedBank
                m2s = ig.draw_subclusters(gs)
                c.label_parts, c.selected_seeds = seeds.parts(m2s), {"mlabel2subclusters": m2s}
                shape = tuple(c.label_parts[0].shape)
            else:
                c.labels, c.selected_seeds = ig.load_seeds(seeds=seeds, genparams=gs)
                shape = tuple(c.labels.shape)
            if not drawn:
                c.gmm_plan = ig.plan_intensities(shape, genparams.get("seed_intensities", {}))
        else:
            if image is None:
                raise ValueError(
                    "If no seeds are passed, an image must be loaded to be used as intensity prior!")
            shape = tuple(image.shape)
        c.shape = shape
        if not drawn:
            c.dplan, c.g, c.bplan, c.rplan, c.nplan = self._draw_plans(shape, genparams)
        dplan, bplan, rplan = c.dplan, c.bplan, c.rplan

        c.sb = sd.make_spec(dplan, shape, flip_in_kernel=True, arena=arena) if dplan.active else None
        c.bias_tabs, c.bias_off = None, None
        if bplan.active:
            c.bias_tabs = K.device_tables_for(self.biasfield.tables(bplan, shape), dev)
            c.bias_off = arena.add(bplan.grid.numpy())
        c.rs_tabs = c.back_tabs = None
        if rplan.active:
            # the three per-axis tables depend on (low-res size, size) only: one DeviceTables object per pair
            rs_cache = self.__dict__.setdefault("_rs_dt", {})
            rs_key = (tuple(rplan.new_size), shape)
            c.rs_tabs = rs_cache.get(rs_key)
            if c.rs_tabs is None:
                if len(rs_cache) > 4096:
                    rs_cache.clear()
                c.rs_tabs = rs_cache[rs_key] = K.DeviceTables(rplan.tabs, dev, arena)
            # zoom-back by 1 / factors, factors = new_size / size (tables.resample_plan): a function of the two shapes
            bt, new = T.zoom_tables_between(tuple(rplan.new_size), shape, True)
            c.back_tabs = K.device_tables_for(bt, dev)
        c.mm_off = arena.add(_MM8_INIT)  # the sample's min/max keys arrive initialised with its parameters
        c.slots_off = arena.add(_MM_SLOTS_INIT) if rplan.active and not _NO_MM_SLOTS else None
        c.gm_off = None
        if c.gmm_plan is not None:
            c.gm_off = (arena.add(c.gmm_plan.mus.numpy()), arena.add(c.gmm_plan.sigmas.numpy()), c.gmm_plan.mus.numel())
        c.has_art = any(a is not None for a in self.artifacts.values())
        c.keep = []
        return c

    def _resolve(self, c):
        f32_view = c.arena.f32
        c.seed_intensities, c.mus, c.sigmas = {}, None, None
        if c.gmm_plan is not None:
            c.mus, c.sigmas = f32_view(c.gm_off[0], (c.gm_off[2],)), f32_view(c.gm_off[1], (c.gm_off[2],))
            c.seed_intensities = {"mus": c.mus, "sigmas": c.sigmas}
        c.bias_dev = f32_view(c.bias_off, tuple(c.bplan.grid.shape)) if c.bplan.active else None
        c.gam = float(c.g) if c.g is not None else None
        c.spec = c.sb.build() if c.dplan.active else None

    def _synth_params(self, c, artifacts):
        return self._params(c.selected_seeds, c.seed_intensities, c.dplan, c.g, c.bplan, c.rplan, c.nplan, artifacts)

    # ---- keyed mode (fetalsyngen_amd/keyed.py, csrc/fsg_keyed.hip) -----------------------------------------------------
    def keyed_context(self, shape):
        from .. import keyed

        ctxs = self.__dict__.setdefault("_keyed", {})
        shape = tuple(int(v) for v in shape)
        kc = ctxs.get(shape)
        if kc is None:
            if len(ctxs) >= 4:
                ctxs.pop(next(iter(ctxs))).close()
            kc = ctxs[shape] = keyed.KeyedContext(self, shape)
        return kc

    def _is_keyed(self) -> bool:
        return (self.rng or _rng.get_mode()) == "keyed"

    def _keyed_applies(self, image, segmentation, seeds, genparams, segmentation_u8, labels_u8) -> bool:
        return (self.native_pipeline and image is None and not genparams and hasattr(seeds, "parts")
                and self.intensity_generator.meta_labels <= 4 and torch.is_tensor(segmentation) and segmentation.is_cuda
                and segmentation.dtype == torch.float32 and segmentation.is_contiguous()
                and (segmentation_u8 is None or labels_u8) and not any(a is not None for a in self.artifacts.values()))

    def _pipeline_keyed(self, segmentation, bank, key, scale01, labels_u8, out=None, seg_out=None, next_key=None):
        """One keyed sample: pointers + key -> ONE native call (draws, the draw kernel, the launch sequence).  Returns
        (image, labels, None, synth_params) or None when the sample is outside the fused kernels' domain.

        next_key: the key of the sample the caller will ask for NEXT on this stream (a batch, a stream of indices): its draw job
        then rides in this sample's floor(min) launch (fsg_keyed_sample_run's look-ahead) -- one launch less on the next sample's
        critical path, the same volumes.  If the next call is for another key or stream, the carried block is simply not used."""
        from .. import keyed

        shape = tuple(segmentation.shape)
        kc = self.keyed_context(shape)
        if not kc._tables_ready:
            kc.register_tables()
        twin = self._label_twin(segmentation)
        if labels_u8 and twin is None:
            return None
        ent = kc.subject(bank, segmentation, twin)
        dev = segmentation.device
        ws = self._workspace(shape, kc.rows_need)
        given = seg_out is not None
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=dev)
        if seg_out is None:
            seg_out = torch.empty(shape, dtype=torch.uint8 if labels_u8 else torch.float32, device=dev)
        # what the previous call carried for this one (see next_key): the parameter block of exactly this key, on this stream
        flags, block = 0, None
        stream_id = K._stream(dev).value
        pre = self.__dict__.setdefault("_pre", {}).pop(stream_id, None)  # one carried block per launch stream
        if pre is not None and pre[0] == key and pre[1] is kc:
            block, flags = pre[2], 1
        if block is None:
            block = torch.empty(kc.block_bytes, dtype=torch.uint8, device=dev)
        nblock = None
        if next_key is not None:
            next_key &= 0xFFFFFFFFFFFFFFFF
            nblock = torch.empty(kc.block_bytes, dtype=torch.uint8, device=dev)
            flags |= 4
        iv = kc.iv
        iv[0] = key if key < (1 << 63) else key - (1 << 64)
        iv[1] = out.data_ptr()
        if labels_u8:
            iv[2], iv[3] = 0, seg_out.data_ptr()
        else:
            iv[2], iv[3] = seg_out.data_ptr(), 0
        iv[4], iv[5], iv[6] = ent[2], ent[1], block.data_ptr()
        iv[7], iv[8], iv[9] = ws["ws0"].data_ptr(), ws["ws1"].data_ptr(), ws["low"].data_ptr()
        iv[10], iv[11], iv[12] = (ws["rows"].data_ptr() if ws["rows"] is not None else 0), ws["stride"], int(bool(scale01))
        tr = None
        if self.stage_traces is not None:
            tr = StageTrace()
            iv[13], iv[14], iv[15] = tr.slots()
        else:
            iv[13] = iv[14] = iv[15] = 0
        iv[16:80] = ent[0]
        events = None
        if self.blur_events is not None:
            self._blur_tick += 1
            if self._blur_tick % self.blur_events_every == 0:
                events = (kc.lib.fsg_event_create(), kc.lib.fsg_event_create())
        iv[80], iv[81] = events if events is not None else (0, 0)
        iv[82:86] = ent[3:7]  # the subject's code volume (0: four label volumes)
        iv[86] = flags
        if nblock is not None:
            iv[87], iv[88] = (next_key if next_key < (1 << 63) else next_key - (1 << 64)), nblock.data_ptr()
        else:
            iv[87] = iv[88] = 0
        d = _lib.KeyedDraws()
        import ctypes as C

        rc = kc.lib.fsg_keyed_sample_run(kc.handle, kc.ivp, len(kc.iv), C.byref(d), K._stream(dev))
        if events is not None:
            if rc == 0 and d.resample_active:
                self.blur_events.append((events[0], events[1], [(a_, int(d.blur_ntaps[a_]) // 2) for a_ in range(3) if d.blur_ntaps[a_]],
                                         tuple(int(v) for v in d.low_shape)))
            else:
                kc.lib.fsg_event_destroy(events[0])
                kc.lib.fsg_event_destroy(events[1])
        if rc in (_lib.E_ALIGN, _lib.E_TOOBIG):
            return None
        _lib.check(rc, "fsg_keyed_sample_run")
        if nblock is not None and d.rode:
            if len(self._pre) > 8:
                self._pre.clear()
            self._pre[stream_id] = (next_key, kc, nblock)
        if tr is not None:
            tr.meta = {"shape": shape, "low_shape": tuple(d.low_shape) if d.resample_active else None,
                       "blur_ntaps": list(d.blur_ntaps), "label_bytes": 2 if ent[3] else 4, "draw_carried": flags & 1}
            self.stage_traces.append(tr)
        if not d.deform_active:  # no warp ran: the labels pass through
            if given:
                seg_out.copy_(twin if labels_u8 else segmentation)
            else:
                seg_out = twin if labels_u8 else segmentation
        return out, seg_out, None, keyed.params_of(d, block)

    def _pipeline(self, image, segmentation, seeds, genparams, scale01: bool, segmentation_u8=None, labels_u8: bool = False,
                  key: int | None = None, next_key: int | None = None):
        """labels_u8: return the labels as uint8 (same values; written as such by the fused warp where the fused path runs,
        converted afterwards otherwise).  key (keyed mode): the sample's 64-bit key; None = the key announced by
        `sharding.seed_for_sample` / `announce_key`, else one drawn from numpy's global generator."""
        if self._is_keyed():
            from .. import sharding

            if key is None:
                key = sharding.take_key()
            if key is None:
                key = int(np.random.randint(0, 1 << 62)) << 1
            key &= 0xFFFFFFFFFFFFFFFF
            if self._keyed_applies(image, segmentation, seeds, genparams, segmentation_u8, labels_u8):
                got = self._pipeline_keyed(segmentation, seeds, key, scale01, labels_u8, next_key=next_key)
                if got is not None:
                    return got
            # outside the keyed path's domain: a "device"-mode sample of the global generators seeded from the key
            np.random.seed(key & 0xFFFFFFFF)
            torch.default_generator.manual_seed(key >> 1)
        if labels_u8:
            out, seg, img, params = self._pipeline_f(image, segmentation, seeds, genparams, scale01, segmentation_u8, True)
            return out, (seg if seg.dtype == torch.uint8 else seg.to(torch.uint8)), img, params
        return self._pipeline_f(image, segmentation, seeds, genparams, scale01, segmentation_u8, False)

    def _pipeline_f(self, image, segmentation, seeds, genparams, scale01: bool, segmentation_u8=None, labels_u8: bool = False):
        with _rng.use(self.rng):
            K._EPOCH[0] += 1
            arena = T.Arena()
            # The device block of THIS sample's parameters was allocated during the previous sample's host phase, i.e. before
            # that sample's kernels were enqueued: whatever owned the block before was last used ahead of them, so the upload
            # may be ordered behind the previous sample's blur alone (fsg_sample_plan::overlap).  The next sample's block
            # is allocated here, before this sample enqueues anything.
            dev = torch.device(self.device)
            nxt = self.__dict__.setdefault("_arena_next", {})
            akey = (dev.index, K._stream(dev).value)
            block = nxt.pop(akey, None) if _HEAD_OVERLAP != "0" else None
            early = block is not None
            if _HEAD_OVERLAP != "0":
                if block is None:
                    block = torch.empty(_ARENA_BLOCK, dtype=torch.uint8, device=dev)
                if len(nxt) > 8:
                    nxt.clear()
                nxt[akey] = torch.empty(_ARENA_BLOCK, dtype=torch.uint8, device=dev)
            c = self._prepare(image, segmentation, seeds, genparams, arena, segmentation_u8)
            c.arena_early = early
            fast = self._native_ok(c, labels_u8) and not _SLOW_PLAN
            if not (fast and block is not None and arena.stage(self.device, block)):
                arena.upload(self.device)
            if fast:
                try:
                    native = self._run_native_fast(c, scale01, labels_u8)
                finally:
                    if arena.pending is not None:  # the native call was not reached: upload on the launch stream now
                        arena.flush(False)
                if native is not None:
                    return native[0], native[1], None, self._synth_params(c, {})
            self._resolve(c)
            if self._native_ok(c, labels_u8):
                native = self._run_native(c, scale01)
                if native is not None:
                    return native[0], native[1], None, self._synth_params(c, {})
            return self._run_stagewise(c, scale01)

    def sample_batch(self, items, genparams_list=None, scale01: bool = False, streams: int = 1, lazy_items: int | None = None,
                     labels_u8: bool = False, keys=None):
        """B samples with one parameter upload and one native call.

        items: sequence of (image | None, segmentation, seeds) as for `sample`; the host draws are made sample by sample
        in this order, so the results are bit-identical to B consecutive `sample` calls under the same generator state
        (reference: B x FetalSynthGen.sample, generator/model.py:231-276).
        Returns (images (B,H,W,D) float32, labels (B,H,W,D) float32, [image_b | None], [synth_params_b]) -- one tensor per
        output, so a stager moves the batch with one copy.  streams > 1: consecutive samples are enqueued round-robin on
        that many side streams (their kernel tails overlap); the current stream waits for all of them before returning.
        Samples outside the fused path's domain fall back to the per-sample path, with the draws already made.
        lazy_items=B: `items` is an iterator of B entries consumed one at a time, each right before that sample's host
        draws (callers that re-seed the global generators per sample, e.g. PrefetchingStream).
        labels_u8: the labels tensor as uint8 (same values), written as such by the fused warp."""
        import ctypes as C


        if lazy_items is None:
            items = list(items)
            B = len(items)
        else:
            B = int(lazy_items)
        if keys is not None and self._is_keyed() and genparams_list is None:
            got = self._sample_batch_keyed(list(items), [int(k) for k in keys], scale01, streams, labels_u8)
            if got is not None:
                return got
            raise ValueError("keyed sample_batch: items outside the fused keyed path (need device-resident SeedBank subjects "
                             "of one shape, no image, no SR-artifact stages)")
        genparams_list = list(genparams_list) if genparams_list is not None else [{}] * B
        if len(genparams_list) != B:
            raise ValueError("genparams_list must have one entry per item")
        dev = torch.device(self.device)
        with _rng.use(self.rng):
            arena = T.Arena()
            ctxs = [self._prepare(img, seg, seeds, gp, arena) for (img, seg, seeds), gp in zip(items, genparams_list)]
            if len(ctxs) != B:
                raise ValueError(f"items yielded {len(ctxs)} entries, expected {B}")
            arena.upload(self.device)
            for c in ctxs:
                self._resolve(c)
            shapes = {c.shape for c in ctxs}
            fused = B > 0 and len(shapes) == 1 and all(self._native_ok(c) for c in ctxs)
            if fused:
                for c in ctxs:
                    self._native_operands(c)
                shape = ctxs[0].shape
                nstreams = max(1, min(int(streams), B))
                main = torch.cuda.current_stream(dev)
                side = self._side_streams(nstreams) if nstreams > 1 else [main]
                need = max(self._rows_needed(c) for c in ctxs)
                wss = []
                for q in range(nstreams):  # one scratch set per stream the batch runs on
                    with torch.cuda.stream(side[q]):
                        wss.append(self._workspace(shape, need))
                out_all = torch.empty((B, *shape), dtype=torch.float32, device=dev)
                seg_all = torch.empty((B, *shape), dtype=torch.uint8 if labels_u8 else torch.float32, device=dev)
                plans = (_lib.SamplePlan * B)()
                ok = True
                for b, c in enumerate(ctxs):
                    if c.spec is None:
                        seg_all[b].copy_(c.seg)  # no deformation: labels pass through (same stream as the upload)
                    ok = ok and self._fill_native_plan(plans[b], c, scale01, wss[b % nstreams], out_all[b], seg_all[b])
                if ok:
                    handles = (C.c_void_p * nstreams)()
                    if nstreams > 1:
                        fork = torch.cuda.Event()
                        fork.record(main)
                        for q in range(nstreams):
                            side[q].wait_event(fork)
                            handles[q] = side[q].cuda_stream
                    else:
                        handles[0] = K._stream(dev).value
                    rc = _lib.load().fsg_sample_run_batch(plans, B, handles, nstreams)
                    if nstreams > 1:
                        for q in range(nstreams):
                            join = torch.cuda.Event()
                            join.record(side[q])
                            main.wait_event(join)
                    if rc not in (_lib.E_ALIGN, _lib.E_TOOBIG):
                        _lib.check(rc, "fsg_sample_run_batch")
                        return out_all, seg_all, [None] * B, [self._synth_params(c, {}) for c in ctxs]
                    fused = False  # nothing usable was produced: per-sample path below (same plans, no new draws)
            outs = []
            for c in ctxs:
                native = self._run_native(c, scale01) if self._native_ok(c) else None
                outs.append((native[0], native[1], None, self._synth_params(c, {})) if native is not None
                            else self._run_stagewise(c, scale01))
        same = len({tuple(o[0].shape) for o in outs}) == 1 if outs else False
        images = torch.stack([o[0] for o in outs]) if same else [o[0] for o in outs]
        ldt = torch.uint8 if labels_u8 else torch.float32
        labels = torch.stack([o[1].to(ldt) for o in outs]) if same else [o[1].to(ldt) if labels_u8 else o[1] for o in outs]
        return images, labels, [o[2] for o in outs], [o[3] for o in outs]

    def _sample_batch_keyed(self, items, keys, scale01, streams, labels_u8):
        """B keyed samples written straight into one (B,H,W,D) tensor per output; sample b is `sample(..., key=keys[b])`."""
        B = len(items)
        if B == 0 or len(keys) != B:
            return None
        if not all(self._keyed_applies(img, seg, seeds, {}, None, labels_u8) for img, seg, seeds in items):
            return None
        shapes = {tuple(seg.shape) for _i, seg, _s in items}
        if len(shapes) != 1:
            return None
        shape = shapes.pop()
        dev = torch.device(self.device)
        out_all = torch.empty((B, *shape), dtype=torch.float32, device=dev)
        seg_all = torch.empty((B, *shape), dtype=torch.uint8 if labels_u8 else torch.float32, device=dev)
        nstreams = max(1, min(int(streams), B))
        main = torch.cuda.current_stream(dev)
        side = self._side_streams(nstreams) if nstreams > 1 else [main]
        if nstreams > 1:
            fork = torch.cuda.Event()
            fork.record(main)
            for q in range(nstreams):
                side[q].wait_event(fork)
        params = []
        for b, ((_img, seg, seeds), key) in enumerate(zip(items, keys)):
            with torch.cuda.stream(side[b % nstreams]):  # (the next sample of THIS stream: its draw job rides along)
                got = self._pipeline_keyed(seg, seeds, key & 0xFFFFFFFFFFFFFFFF, scale01, labels_u8, out=out_all[b],
                                           seg_out=seg_all[b], next_key=keys[b + nstreams] if b + nstreams < B else None)
            if got is None:
                return None
            params.append(got[3])
        if nstreams > 1:
            for q in range(nstreams):
                join = torch.cuda.Event()
                join.record(side[q])
                main.wait_event(join)
        return out_all, seg_all, [None] * B, params

    def _side_streams(self, n):
        cur = self.__dict__.setdefault("_batch_streams", [])
        while len(cur) < n:
            cur.append(torch.cuda.Stream(device=torch.device(self.device)))
        return cur[:n]

    def _run_stagewise(self, c, scale01):
        """Stage-by-stage launches of a prepared sample (images as intensity prior, SR-artifact stages, host label
        tensors, uint8 label output, configurations outside the fused kernels' domain)."""
        dev = self.device
        sd = self.spatial_deform
        image, segmentation, genparams = c.image, c.segmentation, c.genparams
        gmm_plan, dplan, bplan, rplan, nplan = c.gmm_plan, c.dplan, c.bplan, c.rplan, c.nplan
        mus, sigmas, gam, bias_dev, bias_tabs, spec = c.mus, c.sigmas, c.gam, c.bias_dev, c.bias_tabs, c.spec
        rs_tabs, back_tabs, has_art = c.rs_tabs, c.back_tabs, c.has_art
        if True:
            if gmm_plan is not None:
                f = gmm_plan.field
                z = f.device_tensor(dev) if f.host is not None else None
                if c.label_parts is not None:
                    output = K.gmm_sample_parts(c.label_parts, mus, sigmas, noise=z, seed=f.seed or 0,
                                                stream_id=f.stream_id)
                else:
                    labels = c.labels
                    if labels.dtype not in (torch.uint8, torch.int64):
                        labels = labels.long()
                    labels = labels.to(dev).contiguous()
                    output = K.gmm_sample(labels, mus, sigmas, noise=z, seed=f.seed or 0, stream_id=f.stream_id)
            else:
                output = self._intensity_prior(image)
            image = image.to(dev) if image is not None else None
            # one init launch for every min/max key of the sample: [min x,y,z | zoom min] [zoom max | unused x3]
            mm8 = K.new_minmax(dev, 4, 4)
            if dplan.active:
                image, segmentation, output = sd.run(dplan, image, segmentation, output, spec=spec,
                                                     mm6=K.coords_floormin(spec, mm8), gamma=gam, bias=bias_dev,
                                                     bias_tabs=bias_tabs, segmentation_u8=c.segmentation_u8)
            else:
                segmentation = segmentation.to(dev)
                if gam is not None:
                    output = K.gamma(output, gam)
                if bplan.active:
                    output = K.bias_mul(output, bias_dev, bias_tabs)

            fuse_scale = scale01 and not has_art
            f = nplan.field if nplan.active else None
            z = f.device_tensor(dev) if (f is not None and f.host is not None) else None
            if rplan.active:
                low = self.resampled.blur_resample(output.contiguous(), rplan.stds, rs_tabs,
                                                   noise_std=nplan.std32 if nplan.active else 0.0, noise=z,
                                                   seed=(f.seed if (f is not None and f.host is None) else None),
                                                   stream_id=f.stream_id if f is not None else 0)
                mm2 = K.zoom_minmax(low, back_tabs, mm=mm8[3:5])
                output = K.zoom_normalise(low, back_tabs, mm2, mode=1 if fuse_scale else 0)
            else:
                if nplan.active:
                    output = K.add_noise(output, nplan.std32, noise=z, seed=f.seed or 0, stream_id=f.stream_id)
                if fuse_scale:
                    output = K.scale(output, K.reduce_minmax(output), mode=1)
            output, artifacts = self._apply_artifacts(output, segmentation, genparams)
            if scale01 and has_art:
                output = K.scale(output.contiguous(), K.reduce_minmax(output.contiguous()), mode=1)

        return output, segmentation, image, self._synth_params(c, artifacts)
