"""Torch-tensor front-end of the C ABI: pointer/shape plumbing only, no arithmetic.

Every function validates device / dtype / contiguity on the host (a faulting kernel can take the whole
node down), launches on torch's current HIP stream and returns freshly allocated device tensors.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .tables import Arena, TAP_DTYPE

F32 = torch.float32


def _stream(ref=None):
    """Raw hipStream_t of torch's current stream on `ref`'s device (tensor, device or None).

    `torch.cuda.current_stream()` walks through availability checks (environment lookups) on every
    call; the raw getter is ~100x cheaper and this sits in front of every launch."""
    if isinstance(ref, torch.Tensor):
        idx = ref.device.index
    elif ref is None:
        idx = torch.cuda.current_device()
    else:
        idx = torch.device(ref).index
    if idx is None:
        idx = torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def _need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError(
                "fetalsyngen_amd kernels run on an MI355X (device='cuda:N') only; there is no CPU fallback"
            )
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _f32(t, name="tensor"):
    if t.dtype != F32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t


def _dims3(t):
    if t.dim() != 3:
        raise ValueError(f"expected a 3-D volume, got shape {tuple(t.shape)}")
    return int(t.shape[0]), int(t.shape[1]), int(t.shape[2])


_DEV_TABLES: dict = {}  # (device, id(host table)) -> (device bytes, host table kept alive)
_DEV_TABLES_MAX = 2048


_DEV_KEYS: dict = {}


def _device_key(device):
    """Stable string for a device argument (str / torch.device), memoised: this runs a dozen times per sample."""
    k = _DEV_KEYS.get(device)
    if k is None:
        k = _DEV_KEYS[device] = str(device)
    return k


def _device_table(tab, device):
    key = (_device_key(device), id(tab))
    hit = _DEV_TABLES.get(key)
    if hit is not None and hit[1] is tab:
        return hit[0]
    if tab.dtype != TAP_DTYPE:
        raise TypeError("tap table dtype")
    raw = np.ascontiguousarray(tab).view(np.uint8).reshape(-1)
    host = torch.empty(raw.size, dtype=torch.uint8, pin_memory=True)  # pinned: the copy below is async
    host.numpy()[:] = raw
    d = host.to(device, non_blocking=True)
    if len(_DEV_TABLES) >= _DEV_TABLES_MAX:
        _DEV_TABLES.pop(next(iter(_DEV_TABLES)))
    _DEV_TABLES[key] = (d, tab)
    return d


# Bumped by FetalSynthGen once per sample.  A DeviceTables object remembers the epoch it was built in: its upload was enqueued
# on the launch stream during THAT sample's host phase, so work that is ordered only behind the previous sample (the head of a
# sample on the library's side stream, fsg_sample_plan::overlap) may read it from the next sample on, not before.
_EPOCH = [0]


class DeviceTables:
    """Three per-axis tap tables resident on the device.

    Tables produced by the cached builders in `tables.py` are uploaded once per device and reused
    (the host array object is the cache key); ad-hoc tables are uploaded on first use."""

    def __init__(self, tabs, device, arena: Arena | None = None):
        self.lengths = tuple(len(t) for t in tabs)
        self.born = _EPOCH[0]
        self._dev = [_device_table(t, device) for t in tabs]
        self.ptrs = tuple(C.c_void_p(d.data_ptr()) for d in self._dev)  # the device copies never move
        self.ptrs_i = tuple(int(d.data_ptr()) for d in self._dev)        # the same as plain integers (flat plan arrays)


_DT_CACHE: dict = {}  # (id(host table list), device) -> (DeviceTables, the list kept alive)


def device_tables_for(tabs_list, device) -> DeviceTables:
    """DeviceTables for a table list with a stable identity (tables.zoom_tables_between memoises its lists): one dict hit
    per call instead of three table look-ups and a fresh pointer tuple."""
    key = (id(tabs_list), _device_key(device))
    hit = _DT_CACHE.get(key)
    if hit is not None and hit[1] is tabs_list:
        return hit[0]
    dt = DeviceTables(tabs_list, device)
    if len(_DT_CACHE) >= 4096:
        _DT_CACHE.pop(next(iter(_DT_CACHE)))
    _DT_CACHE[key] = (dt, tabs_list)
    return dt


def new_minmax(device, nmin=1, nmax=1):
    mm = torch.empty(nmin + nmax, dtype=torch.int32, device=device)
    _lib.check(_lib.load().fsg_minmax_init(_p(mm), nmin, nmax, _stream(mm)), "fsg_minmax_init")
    return mm


def key_to_float(key: int) -> float:
    return float(_lib.load().fsg_key_to_float(int(key)))


# ---- RNG / K1 ---------------------------------------------------------------------------------
def randn(shape, seed: int, stream_id: int, device) -> torch.Tensor:
    out = torch.empty(tuple(shape), dtype=F32, device=device)
    _need_gpu(out)
    _lib.check(_lib.load().fsg_randn_f32(_p(out), out.numel(), seed, stream_id, _stream(out)), "fsg_randn_f32")
    return out


def gmm_sample(labels, mus, sigmas, noise=None, seed=0, stream_id=0) -> torch.Tensor:
    _need_gpu(labels, mus, sigmas, noise)
    _f32(mus), _f32(sigmas)
    ntab = int(mus.numel())
    if sigmas.numel() != ntab or not (0 < ntab <= 256):
        raise ValueError("mus/sigmas must have the same length in 1..256")
    if noise is not None and (_f32(noise).numel() != labels.numel()):
        raise ValueError("noise must match labels")
    out = torch.empty(labels.shape, dtype=F32, device=labels.device)
    lib = _lib.load()
    if labels.dtype == torch.uint8:
        fn, name = lib.fsg_gmm_sample_u8, "fsg_gmm_sample_u8"
    elif labels.dtype == torch.int64:
        fn, name = lib.fsg_gmm_sample_i64, "fsg_gmm_sample_i64"
    else:
        raise TypeError("labels must be uint8 or int64")
    _lib.check(fn(_p(labels), labels.numel(), _p(mus), _p(sigmas), ntab, _p(noise), seed, stream_id, _p(out),
                  _stream(out)), name)
    return out


def gmm_sample_parts(parts, mus, sigmas, noise=None, seed=0, stream_id=0) -> torch.Tensor:
    """GMM draw from the per-meta-label seed volumes (uint8, disjoint supports) without summing them."""
    parts = [p for p in parts if p is not None]
    if not 1 <= len(parts) <= 4:
        raise ValueError("1..4 label volumes expected")
    _need_gpu(*parts, mus, sigmas, noise)
    for p in parts:
        if p.dtype != torch.uint8 or p.shape != parts[0].shape:
            raise TypeError("label parts must be uint8 volumes of one shape")
    _f32(mus), _f32(sigmas)
    ntab = int(mus.numel())
    if noise is not None and _f32(noise).numel() != parts[0].numel():
        raise ValueError("noise must match labels")
    out = torch.empty(parts[0].shape, dtype=F32, device=parts[0].device)
    ptrs = [_p(p) for p in parts] + [C.c_void_p(0)] * (4 - len(parts))
    _lib.check(_lib.load().fsg_gmm_sample_u8x4(*ptrs, out.numel(), _p(mus), _p(sigmas), ntab, _p(noise), seed,
                                               stream_id, _p(out), _stream(out)), "fsg_gmm_sample_u8x4")
    return out


def sample_head(parts, mus, sigmas, spec: "DeformSpec", bias=None, bias_tabs=None, noise=None, seed=0, stream_id=0,
                mm3=None):
    """Head of a sample as one launch (fsg_sample_head_f32): the GMM draw of `gmm_sample_parts`, the per-row coarse values of
    `DeformSpec.prepare_rows` and the first floor(min) pass of `coords_floormin`.  `mm3` must hold initialised keys
    (`new_minmax(dev, 3, 0)` when omitted); `coords_floormin_rest` completes them.  Returns (image, mm3); the rows are
    attached to `spec` as `prepare_rows` would."""
    parts = [p for p in parts if p is not None]
    if not 1 <= len(parts) <= 4:
        raise ValueError("1..4 label volumes expected")
    _need_gpu(*parts, mus, sigmas, noise)
    out = torch.empty(parts[0].shape, dtype=F32, device=parts[0].device)
    if mm3 is None:
        mm3 = new_minmax(out.device, 3, 0)
    need = 3 * int(spec.c.field_dims[2]) + (int(bias.shape[2]) if bias is not None else 0)
    stride = (max(need, 1) + 3) // 4 * 4
    rows = torch.empty(spec.shape[0] * spec.shape[1] * stride, dtype=F32, device=out.device)
    epi = _epilogue(None, bias, bias_tabs, spec.shape)
    spec.c.rows, spec.c.row_stride = None, 0
    ptrs = [_p(p) for p in parts] + [C.c_void_p(0)] * (4 - len(parts))
    _lib.check(_lib.load().fsg_sample_head_f32(*ptrs, out.numel(), _p(mus), _p(sigmas), int(mus.numel()), _p(noise), seed,
                                               stream_id, _p(out), C.byref(spec.c), C.byref(epi), _p(rows), stride,
                                               _p(mm3), _stream(out)), "fsg_sample_head_f32")
    spec.c.rows, spec.c.row_stride = rows.data_ptr(), stride
    spec._keep.append(rows)
    spec._rows_bias = int(bias.shape[2]) if bias is not None else 0
    return out, mm3


def coords_floormin_rest(spec: "DeformSpec", mm3) -> torch.Tensor:
    """Conditional full pass that completes the floor(min) keys after `sample_head` (fsg_coords_floormin_rest_f32)."""
    _lib.check(_lib.load().fsg_coords_floormin_rest_f32(C.byref(spec.c), _p(mm3), _stream(mm3)),
               "fsg_coords_floormin_rest_f32")
    return mm3


def label_stats(labels_u8, values, nlabels: int):
    """(count int64[nlabels], mean f64, var f64) per label -- wave-level reductions on the device."""
    _need_gpu(labels_u8, values)
    if labels_u8.dtype != torch.uint8 or labels_u8.numel() != _f32(values).numel():
        raise TypeError("labels uint8 and values float32 of equal size expected")
    dev = values.device
    cnt = torch.zeros(nlabels, dtype=torch.int64, device=dev)
    s1 = torch.zeros(nlabels, dtype=torch.float64, device=dev)
    s2 = torch.zeros(nlabels, dtype=torch.float64, device=dev)
    _lib.check(_lib.load().fsg_label_stats_u8(_p(labels_u8), _p(values), values.numel(), nlabels, _p(cnt), _p(s1),
                                              _p(s2), _stream(values)), "fsg_label_stats_u8")
    n = cnt.clamp(min=1).double()
    mean = s1 / n
    return cnt, mean, s2 / n - mean * mean


# ---- zoom family --------------------------------------------------------------------------------
def _zoom_args(src, tabs: DeviceTables, nch=1):
    _need_gpu(src)
    _f32(src)
    if nch == 1:
        sx, sy, sz = _dims3(src)
    else:
        if src.dim() != 4 or src.shape[3] != nch:
            raise ValueError("channel-last 4-D source expected")
        sx, sy, sz = (int(v) for v in src.shape[:3])
    return sx, sy, sz


def zoom3d(src, tabs: DeviceTables) -> torch.Tensor:
    nch = 1 if src.dim() == 3 else int(src.shape[3])
    if nch not in (1, 3):
        raise ValueError("1 or 3 channels supported")
    sx, sy, sz = _zoom_args(src, tabs, nch)
    dx, dy, dz = tabs.lengths
    shape = (dx, dy, dz) if nch == 1 else (dx, dy, dz, nch)
    dst = torch.empty(shape, dtype=F32, device=src.device)
    tx, ty, tz = tabs.ptrs
    _lib.check(_lib.load().fsg_zoom3d_f32(_p(src), sx, sy, sz, nch, tx, ty, tz, _p(dst), dx, dy, dz, _stream(src)),
               "fsg_zoom3d_f32")
    return dst


def resample_noise(src, tabs: DeviceTables, noise_std=0.0, noise=None, seed=None, stream_id=0) -> torch.Tensor:
    sx, sy, sz = _zoom_args(src, tabs)
    dx, dy, dz = tabs.lengths
    dst = torch.empty((dx, dy, dz), dtype=F32, device=src.device)
    mode = 0
    if noise is not None:
        _need_gpu(noise)
        if _f32(noise).numel() != dst.numel():
            raise ValueError("noise size")
        mode = 1
    elif seed is not None:
        mode = 2
    tx, ty, tz = tabs.ptrs
    _lib.check(_lib.load().fsg_resample_noise_f32(_p(src), sx, sy, sz, tx, ty, tz, _p(dst), dx, dy, dz, mode,
                                                  _p(noise), seed or 0, stream_id, float(noise_std), _stream(src)),
               "fsg_resample_noise_f32")
    return dst


def zoom_minmax(src, tabs: DeviceTables, mm=None) -> torch.Tensor:
    sx, sy, sz = _zoom_args(src, tabs)
    dx, dy, dz = tabs.lengths
    if mm is None:
        mm = new_minmax(src.device)
    tx, ty, tz = tabs.ptrs
    _lib.check(_lib.load().fsg_zoom3d_minmax_f32(_p(src), sx, sy, sz, tx, ty, tz, dx, dy, dz, _p(mm), _stream(src)),
               "fsg_zoom3d_minmax_f32")
    return mm


MM_SLOT_STRIDE = 16  # include/fsg_hip.h: FSG_MM_SLOT_STRIDE


def zoom_minmax_sharded(src, tabs: DeviceTables, nslots: int = 32) -> torch.Tensor:
    """K9 pass A with the two keys sharded over `nslots` slots (nslots x 16 int32; what fsg_sample_run uses): a workgroup
    updates slot (its index % nslots), `zoom_normalise` reduces the slots."""
    sx, sy, sz = _zoom_args(src, tabs)
    dx, dy, dz = tabs.lengths
    init = np.zeros((nslots, MM_SLOT_STRIDE), dtype=np.int32)
    init[:, 0], init[:, 1] = 0x7F800000, -2139095041  # key(+inf), key(-inf)
    slots = torch.from_numpy(init).to(src.device)
    tx, ty, tz = tabs.ptrs
    _lib.check(_lib.load().fsg_zoom3d_minmax_sharded_f32(_p(src), sx, sy, sz, tx, ty, tz, dx, dy, dz, _p(slots), nslots,
                                                         _stream(src)), "fsg_zoom3d_minmax_sharded_f32")
    return slots


def zoom_normalise(src, tabs: DeviceTables, mm, mode: int) -> torch.Tensor:
    """`mm`: the two keys of `zoom_minmax`, or the (nslots, 16) slots of `zoom_minmax_sharded`."""
    sx, sy, sz = _zoom_args(src, tabs)
    _need_gpu(mm)
    dx, dy, dz = tabs.lengths
    dst = torch.empty((dx, dy, dz), dtype=F32, device=src.device)
    tx, ty, tz = tabs.ptrs
    if mm.dim() == 2 and mm.shape[1] == MM_SLOT_STRIDE:
        _lib.check(_lib.load().fsg_zoom3d_normalise_sharded_f32(_p(src), sx, sy, sz, tx, ty, tz, _p(dst), dx, dy, dz,
                                                                _p(mm), int(mm.shape[0]), mode, _stream(src)),
                   "fsg_zoom3d_normalise_sharded_f32")
        return dst
    _lib.check(_lib.load().fsg_zoom3d_normalise_f32(_p(src), sx, sy, sz, tx, ty, tz, _p(dst), dx, dy, dz, _p(mm),
                                                    mode, _stream(src)), "fsg_zoom3d_normalise_f32")
    return dst


# ---- deformation ----------------------------------------------------------------------------------
class DeformSpec:
    """Host-side description of one spatial deformation + its device-resident small arrays."""

    def __init__(self, shape, A32, centre32, c2_32, flip, field_small=None, field_tabs=None, device=None):
        self.shape = tuple(int(v) for v in shape)
        self.device = device
        d = _lib.Deform()
        d.shape[:] = self.shape
        d.A[:] = np.asarray(A32, dtype=np.float32).reshape(-1).tolist()
        d.centre[:] = np.asarray(centre32, dtype=np.float32).tolist()
        d.c2[:] = np.asarray(c2_32, dtype=np.float32).tolist()
        d.flip = int(bool(flip))
        self._keep = []
        if field_small is not None:
            _need_gpu(field_small)
            _f32(field_small)
            if field_small.dim() != 4 or field_small.shape[3] != 3:
                raise ValueError("coarse field must be (s0,s1,s2,3)")
            if field_tabs.lengths != self.shape:
                raise ValueError("field tables must have the grid's lengths")
            d.field_dims[:] = [int(v) for v in field_small.shape[:3]]
            d.field = field_small.data_ptr()
            tx, ty, tz = field_tabs.ptrs
            d.tx, d.ty, d.tz = tx.value, ty.value, tz.value
            self._keep += [field_small, field_tabs]
        else:
            d.field_dims[:] = [0, 0, 0]
        d.rows, d.row_stride = None, 0
        self.c = d

    def prepare_rows(self, bias=None, bias_tabs=None):
        """Precompute the per-(x,y) coarse rows (displacement, and bias when it will be fused into the
        warp) into a workspace; later min/max and warp launches start each row with one coalesced load."""
        f2 = int(self.c.field_dims[2])
        b2 = int(bias.shape[2]) if bias is not None else 0
        need = 3 * f2 + b2
        if need == 0 or need > 512:
            return self
        stride = (need + 3) // 4 * 4
        rows = torch.empty(self.shape[0] * self.shape[1] * stride, dtype=F32, device=self.device)
        epi = _epilogue(None, bias, bias_tabs, self.shape)
        _lib.check(_lib.load().fsg_deform_rows_f32(C.byref(self.c), C.byref(epi), _p(rows), stride, _stream(rows)),
                   "fsg_deform_rows_f32")
        self.c.rows, self.c.row_stride = rows.data_ptr(), stride
        self._keep.append(rows)
        self._rows_bias = b2
        return self


def _epilogue(gamma, bias, bias_tabs, shape):
    epi = _lib.Epilogue()
    epi.gamma = float(np.float32(gamma)) if gamma is not None else 0.0
    if bias is not None:
        _need_gpu(bias)
        _f32(bias)
        if bias_tabs.lengths != tuple(shape):
            raise ValueError("bias tables must have the grid's lengths")
        epi.bias_dims[:] = _dims3(bias)
        epi.bias = bias.data_ptr()
        bx, by, bz = bias_tabs.ptrs
        epi.bx, epi.by, epi.bz = bx.value, by.value, bz.value
    return epi


def coords_minmax(spec: DeformSpec) -> torch.Tensor:
    mm6 = new_minmax(spec.device, 3, 3)
    _lib.check(_lib.load().fsg_coords_minmax_f32(C.byref(spec.c), _p(mm6), _stream(mm6)), "fsg_coords_minmax_f32")
    return mm6


def coords_floormin(spec: DeformSpec, mm3=None) -> torch.Tensor:
    """Keys whose floor is floor(min coordinate) per axis (faces first, full pass only if needed)."""
    if mm3 is None:
        mm3 = new_minmax(spec.device, 3, 0)
    rc = _lib.load().fsg_coords_floormin_f32(C.byref(spec.c), _p(mm3), _stream(mm3))
    if rc == _lib.E_TOOBIG:
        return coords_minmax(spec)
    _lib.check(rc, "fsg_coords_floormin_f32")
    return mm3


def coords(spec: DeformSpec, mm6):
    out = [torch.empty(spec.shape, dtype=F32, device=spec.device) for _ in range(3)]
    _lib.check(_lib.load().fsg_coords_f32(C.byref(spec.c), _p(mm6), _p(out[0]), _p(out[1]), _p(out[2]), _stream(mm6)),
               "fsg_coords_f32")
    return out


def warp(spec: DeformSpec, mm6, src_lin=None, src_nn=None, gamma=None, bias=None, bias_tabs=None, nn_out=None):
    """Fused warp: returns (out_lin | None, out_nn | None).

    `src_nn` float32 -> float32 labels (reference contract); uint8 -> uint8, or float32 when
    `nn_out=torch.float32` (uint8 device copy of a float32 segmentation, exact for 0..255)."""
    _need_gpu(mm6, src_lin, src_nn, bias)
    for s_ in (src_lin, src_nn):
        if s_ is not None and tuple(s_.shape) != spec.shape:
            raise ValueError(f"volume shape {tuple(s_.shape)} != grid {spec.shape}")
    out_lin = torch.empty_like(_f32(src_lin)) if src_lin is not None else None
    epi = _epilogue(gamma, bias, bias_tabs, spec.shape)
    if spec.c.rows and (int(bias.shape[2]) if bias is not None else 0) > getattr(spec, "_rows_bias", 0):
        raise ValueError("row workspace was prepared without this bias grid; call prepare_rows(bias, bias_tabs)")
    lib = _lib.load()
    args = lambda src, out: (C.byref(spec.c), _p(mm6), _p(src_lin), _p(out_lin), _p(src), _p(out), C.byref(epi),
                             _stream(mm6))
    if src_nn is None or src_nn.dtype == F32:
        out_nn = torch.empty_like(src_nn) if src_nn is not None else None
        _lib.check(lib.fsg_warp_f32(*args(src_nn, out_nn)), "fsg_warp_f32")
    elif src_nn.dtype == torch.uint8:
        if nn_out in (None, torch.uint8):
            out_nn = torch.empty_like(src_nn)
            _lib.check(lib.fsg_warp_f32_u8(*args(src_nn, out_nn)), "fsg_warp_f32_u8")
        elif nn_out == F32:
            out_nn = torch.empty(src_nn.shape, dtype=F32, device=src_nn.device)
            rc = lib.fsg_warp_f32_u8_to_f32(*args(src_nn, out_nn))
            if rc == _lib.E_ALIGN:  # outside the brick kernel's domain: float32 label volume, row kernel
                _lib.check(lib.fsg_warp_f32(*args(src_nn.to(F32), out_nn)), "fsg_warp_f32")
            else:
                _lib.check(rc, "fsg_warp_f32_u8_to_f32")
        else:
            raise TypeError("nn_out must be torch.uint8 or torch.float32")
    else:
        raise TypeError("nearest-neighbour volume must be float32 or uint8")
    return out_lin, out_nn


def interp3d(src, ii, jj, kk, mode: str, default_value=0.0) -> torch.Tensor:
    _need_gpu(src, ii, jj, kk)
    sx, sy, sz = _dims3(_f32(src))
    if not (ii.shape == jj.shape == kk.shape):
        raise ValueError("coordinate shapes differ")
    for c in (ii, jj, kk):
        _f32(c)
    if mode not in ("linear", "nearest"):
        raise Exception("mode must be linear or nearest")
    dst = torch.empty(ii.shape, dtype=F32, device=src.device)
    _lib.check(_lib.load().fsg_interp3d_f32(_p(src), sx, sy, sz, _p(ii), _p(jj), _p(kk), ii.numel(),
                                            1 if mode == "nearest" else 0, float(default_value), _p(dst), _stream(src)),
               "fsg_interp3d_f32")
    return dst


# ---- pointwise / blur / reductions ------------------------------------------------------------
def gamma(x, g: float) -> torch.Tensor:
    _need_gpu(x)
    out = torch.empty_like(_f32(x))
    _lib.check(_lib.load().fsg_gamma_f32(_p(x), x.numel(), float(np.float32(g)), _p(out), _stream(x)), "fsg_gamma_f32")
    return out


def cast_f16(x) -> torch.Tensor:
    """float32 -> float16 copy (round to nearest even) by fsg_cast_f32_to_f16: the optional half-precision image."""
    _need_gpu(x)
    x = _f32(x).contiguous()
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().fsg_cast_f32_to_f16(_p(x), x.numel(), _p(out), _stream(x)), "fsg_cast_f32_to_f16")
    return out


def bias_mul(x, bias, bias_tabs: DeviceTables) -> torch.Tensor:
    _need_gpu(x, bias)
    nx, ny, nz = _dims3(_f32(x))
    b0, b1, b2 = _dims3(_f32(bias))
    if bias_tabs.lengths != (nx, ny, nz):
        raise ValueError("bias tables must have the volume's lengths")
    out = torch.empty_like(x)
    bx, by, bz = bias_tabs.ptrs
    _lib.check(_lib.load().fsg_bias_mul_f32(_p(x), nx, ny, nz, _p(bias), b0, b1, b2, bx, by, bz, _p(out), _stream(x)),
               "fsg_bias_mul_f32")
    return out


def add_noise(x, noise_std: float, noise=None, seed=0, stream_id=0) -> torch.Tensor:
    _need_gpu(x, noise)
    out = torch.empty_like(_f32(x))
    if noise is not None and _f32(noise).numel() != x.numel():
        raise ValueError("noise size")
    _lib.check(_lib.load().fsg_add_noise_f32(_p(x), x.numel(), _p(noise), seed, stream_id, float(noise_std), _p(out),
                                             _stream(x)), "fsg_add_noise_f32")
    return out


def blur_axis(x, axis: int, taps: np.ndarray, force_generic=False) -> torch.Tensor:
    _need_gpu(x)
    nx, ny, nz = _dims3(_f32(x))
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    if taps.ndim != 1 or len(taps) % 2 == 0:
        raise ValueError("odd number of taps expected")
    out = torch.empty_like(x)
    lib = _lib.load()
    rc = _lib.E_ALIGN
    if not force_generic and len(taps) <= 129:
        rc = lib.fsg_blur_axis_taps_host_f32(_p(x), _p(out), nx, ny, nz, axis,
                                             taps.ctypes.data_as(C.POINTER(C.c_float)), len(taps), _stream(x))
    if rc == _lib.E_ALIGN:  # shape not covered by the tuned kernels -> generic kernel (still HIP)
        tdev = torch.from_numpy(taps).to(x.device)
        rc = lib.fsg_blur_axis_f32(_p(x), _p(out), nx, ny, nz, axis, _p(tdev), len(taps), _stream(x))
    _lib.check(rc, "fsg_blur_axis")
    return out


def blur_yz(x, taps_y: np.ndarray, taps_z: np.ndarray):
    """Axis-1 then axis-2 blur in one launch (fsg_blur_yz_taps_host_f32); None when the shape / radii are outside the
    fused kernel's domain (the caller then runs the two single-axis passes)."""
    _need_gpu(x)
    nx, ny, nz = _dims3(_f32(x))
    ty = np.ascontiguousarray(taps_y, dtype=np.float32)
    tz = np.ascontiguousarray(taps_z, dtype=np.float32)
    out = torch.empty_like(x)
    fp = C.POINTER(C.c_float)
    rc = _lib.load().fsg_blur_yz_taps_host_f32(_p(x), _p(out), nx, ny, nz, ty.ctypes.data_as(fp), len(ty),
                                               tz.ctypes.data_as(fp), len(tz), _stream(x))
    if rc == _lib.E_ALIGN:
        return None
    _lib.check(rc, "fsg_blur_yz_taps_host_f32")
    return out


def blur_resample(x, tabs: DeviceTables, taps, noise_std=0.0, noise=None, seed=None, stream_id=0, keep_mid=False):
    """K6 + K7 (+ K8) as the fused pair of launches (fsg_blur_resample_x_f32, fsg_blur_resample_yz_noise_f32): `taps` = the
    three per-axis Gaussian tap arrays.  None when the configuration is outside the fused kernels' domain."""
    _need_gpu(x, noise)
    n0, n1, n2 = _dims3(_f32(x))
    m0, m1, m2 = tabs.lengths
    tp = [np.ascontiguousarray(t, dtype=np.float32) for t in taps]
    lib = _lib.load()
    if not lib.fsg_blur_resample_supported(n0, n1, n2, m0, m1, m2, len(tp[0]), len(tp[1]), len(tp[2])):
        return None
    fp = C.POINTER(C.c_float)
    mid = torch.empty((m0, n1, n2), dtype=F32, device=x.device)
    out = torch.empty((m0, m1, m2), dtype=F32, device=x.device)
    tx, ty, tz = tabs.ptrs
    _lib.check(lib.fsg_blur_resample_x_f32(_p(x), n0, n1, n2, tx, m0, tp[0].ctypes.data_as(fp), len(tp[0]), _p(mid), _stream(x)),
               "fsg_blur_resample_x_f32")
    mode = 0
    if noise is not None:
        if _f32(noise).numel() != out.numel():
            raise ValueError("noise size")
        mode = 1
    elif seed is not None:
        mode = 2
    _lib.check(lib.fsg_blur_resample_yz_noise_f32(_p(mid), m0, n1, n2, ty, tz, m1, m2, tp[1].ctypes.data_as(fp), len(tp[1]),
                                                  tp[2].ctypes.data_as(fp), len(tp[2]), mode, _p(noise), seed or 0, stream_id,
                                                  float(noise_std), _p(out), _stream(x)), "fsg_blur_resample_yz_noise_f32")
    return (out, mid) if keep_mid else out


def reduce_minmax(x) -> torch.Tensor:
    _need_gpu(x)
    mm = new_minmax(x.device)
    _lib.check(_lib.load().fsg_reduce_minmax_f32(_p(_f32(x)), x.numel(), _p(mm), _stream(x)), "fsg_reduce_minmax_f32")
    return mm


def scale(x, mm, mode: int) -> torch.Tensor:
    _need_gpu(x, mm)
    out = torch.empty_like(_f32(x))
    _lib.check(_lib.load().fsg_scale_f32(_p(x), x.numel(), _p(mm), mode, _p(out), _stream(x)), "fsg_scale_f32")
    return out


# ---- SR-artifact slice-stack simulation (fsg_slice_acq.hip) -------------------------------------------
SA_MODES = {"linear": 0, "nearest_psf": 1, "torch": 2}


def _sa_mode(semantics: str, interp_psf: bool) -> int:
    if semantics == "torch":
        return SA_MODES["torch"]
    if semantics != "cuda":
        raise ValueError(f"semantics must be 'cuda' or 'torch', got {semantics!r}")
    return SA_MODES["nearest_psf"] if interp_psf else SA_MODES["linear"]


def _sa_mask(m, shape, name):
    if m is None or m.numel() == 0:
        return None
    _need_gpu(m)
    if m.dtype not in (torch.bool, torch.uint8):
        raise TypeError(f"{name} must be bool/uint8, got {m.dtype}")
    if tuple(m.shape[-len(shape):]) != tuple(shape) or m.numel() != int(np.prod(shape)):
        raise ValueError(f"{name} shape {tuple(m.shape)} does not match {tuple(shape)}")
    return m


def _sa_common(transforms, psf):
    _need_gpu(transforms, psf)
    _f32(transforms, "transforms"), _f32(psf, "psf")
    if transforms.dim() != 3 or tuple(transforms.shape[1:]) != (3, 4):
        raise ValueError(f"transforms must be (n,3,4), got {tuple(transforms.shape)}")
    if psf.dim() != 3:
        raise ValueError(f"psf must be 3-D, got {tuple(psf.shape)}")
    return int(transforms.shape[0]), tuple(int(v) for v in psf.shape)


def slice_acq_forward(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight=False,
                      interp_psf=False, semantics="cuda"):
    """(n,3,4) transforms x (D,H,W) volume -> (n,h,w) slices [, weights]  (fsg_slice_acq_forward_f32)."""
    n, (pd, ph, pw) = _sa_common(transforms, psf)
    _need_gpu(vol)
    D, H, W = _dims3(_f32(vol, "vol"))
    h, w = int(slice_shape[0]), int(slice_shape[1])
    vm, sm = _sa_mask(vol_mask, (D, H, W), "vol_mask"), _sa_mask(slices_mask, (n, h, w), "slices_mask")
    out = torch.empty((n, h, w), dtype=F32, device=vol.device)
    wgt = torch.empty_like(out) if need_weight else None
    rc = _lib.load().fsg_slice_acq_forward_f32(_p(transforms), _p(vol), _p(vm), _p(psf), pd, ph, pw, _p(sm), _p(out),
                                               _p(wgt), D, H, W, n, h, w, float(res_slice),
                                               _sa_mode(semantics, interp_psf), _stream(vol))
    _lib.check(rc, "fsg_slice_acq_forward_f32")
    return (out, wgt) if need_weight else out


def slice_acq_adjoint(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf=False,
                      equalize=False, semantics="cuda", return_weight=False, slice_ids=None):
    """(n,h,w) slices -> (D,H,W) volume  (fsg_slice_acq_adjoint_f32 + fsg_equalize_f32).
    slice_ids (host integer tensor, len(transforms)): use slices[slice_ids[z]] for transform z (subset without a copy)."""
    n, (pd, ph, pw) = _sa_common(transforms, psf)
    _need_gpu(slices)
    _f32(slices, "slices")
    ns = int(slices.shape[0])
    if slices.dim() != 3 or (slice_ids is None and ns != n):
        raise ValueError(f"slices must be (n,h,w) with n={n}, got {tuple(slices.shape)}")
    if slice_ids is not None:
        # host tensor on purpose: the ids index device memory, so they are range-checked here before a kernel sees them
        if slice_ids.is_cuda or slice_ids.numel() != n:
            raise ValueError("slice_ids must be a host integer tensor with one entry per transform")
        ids = slice_ids.to(torch.int64)
        if n and (int(ids.min()) < 0 or int(ids.max()) >= ns):
            raise IndexError(f"slice_ids out of range for {ns} slices")
        slice_ids = _upload(ids.to(torch.int32).contiguous(), slices.device)
    h, w = int(slices.shape[1]), int(slices.shape[2])
    D, H, W = (int(v) for v in vol_shape)
    vm, sm = _sa_mask(vol_mask, (D, H, W), "vol_mask"), _sa_mask(slices_mask, (ns, h, w), "slices_mask")
    vol = torch.empty((D, H, W), dtype=F32, device=slices.device)
    wgt = torch.empty_like(vol) if (equalize or return_weight) else None
    mode = _sa_mode(semantics, interp_psf)
    lib, st = _lib.load(), _stream(slices)
    rc = lib.fsg_slice_acq_adjoint_f32(_p(transforms), _p(psf), pd, ph, pw, _p(slices), _p(sm), _p(slice_ids), _p(vm),
                                       _p(vol), _p(wgt), D, H, W, n, h, w, float(res_slice), mode, st)
    _lib.check(rc, "fsg_slice_acq_adjoint_f32")
    torch_sem = mode == SA_MODES["torch"]
    if equalize or (torch_sem and vm is not None):
        rc = lib.fsg_equalize_f32(_p(vol), _p(wgt if equalize else None), _p(vm if torch_sem else None),
                                  1e-2 if torch_sem else 0.0, vol.numel(), st)
        _lib.check(rc, "fsg_equalize_f32")
    return (vol, wgt) if return_weight else vol


# ---- SR-artifact volumetric helpers (fsg_artifacts.hip) ----------------------------------------------------
def _upload(host: torch.Tensor, device):
    """Small host tensor -> device through a pinned staging copy (async on the current stream)."""
    pin = torch.empty(host.shape, dtype=host.dtype, pin_memory=True)
    pin.copy_(host)
    return pin.to(device, non_blocking=True)


def mog3d(shape, centers, sigmas, device) -> torch.Tensor:
    """clamp(sum of k anisotropic Gaussians, 0, 1) on a (D,H,W) grid; centers/sigmas (k,3) in the (x0,y0,z0) order
    `mog_3d_tensor` unpacks (x pairs with the LAST axis).  Host arrays or device tensors."""
    D, H, W = (int(v) for v in shape)

    def dev(a):
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.to(F32).contiguous()
        return _upload(torch.as_tensor(np.asarray(a, dtype=np.float32)).reshape(-1, 3), device)

    c, s = dev(centers), dev(sigmas)
    k = int(c.shape[0])
    if tuple(c.shape) != (k, 3) or tuple(s.shape) != (k, 3) or k == 0:
        raise ValueError("centers and sigmas must be (k,3) with k >= 1")
    out = torch.empty((D, H, W), dtype=F32, device=device)
    tab = torch.empty(k * 3 * max(D, H, W), dtype=F32, device=device)
    _lib.check(_lib.load().fsg_mog3d_f32(_p(c), _p(s), k, D, H, W, _p(tab), _p(out), _stream(out)), "fsg_mog3d_f32")
    return out


class PerlinPlan:
    """Device-resident lattices of one fractal-noise draw: per octave the gradient lattice and the per-axis
    linspace coordinates (built on the host in the reference's arithmetic, uploaded in one copy)."""

    def __init__(self, shape, octaves, device):
        # octaves: list of (grad (r0+1,r1+1,r2+1,3) fp32, [lin0, lin1, lin2] fp32, (r0,r1,r2), amplitude)
        self.shape = tuple(int(v) for v in shape)
        self.n = len(octaves)
        if not 1 <= self.n <= 8:
            raise ValueError("1..8 octaves")
        flat, offs = [], []
        pos = 0
        for g, lins, r, _amp in octaves:
            if tuple(g.shape) != (r[0] + 1, r[1] + 1, r[2] + 1, 3) or [len(v) for v in lins] != list(self.shape):
                raise ValueError("perlin octave: lattice / axis table shapes do not match")
            a = g.reshape(-1).to(F32)
            b = torch.cat([v.to(F32) for v in lins])
            offs.append((pos, pos + a.numel()))
            flat += [a, b]
            pos += a.numel() + b.numel()
        self.buf = _upload(torch.cat(flat), device)
        base = self.buf.data_ptr()
        self.grads = (C.c_void_p * self.n)(*[base + 4 * o[0] for o in offs])
        self.lins = (C.c_void_p * self.n)(*[base + 4 * o[1] for o in offs])
        self.res = (C.c_int32 * (3 * self.n))(*[int(v) for o in octaves for v in o[2]])
        self.amps = (C.c_float * self.n)(*[float(o[3]) for o in octaves])


def perlin_fractal(plan: PerlinPlan, mm=None):
    """Raw fractal noise volume + its (min,max) keys."""
    dev = plan.buf.device
    out = torch.empty(plan.shape, dtype=F32, device=dev)
    if mm is None:
        mm = new_minmax(dev, 1, 1)
    n0, n1, n2 = plan.shape
    rc = _lib.load().fsg_perlin_fractal_f32(plan.grads, plan.lins, plan.res, plan.amps, plan.n, n0, n1, n2, _p(out), _p(mm),
                                            _stream(out))
    _lib.check(rc, "fsg_perlin_fractal_f32")
    return out, mm


def blend(a, b, w, w_mm=None, increase=0.0, seg=None, noise_std=None, b_mm=None, a_mm=None, want_weight=False,
          want_out=True):
    """out = (1-w) a + w b (see fsg_blend_f32).  w_mm given: w is raw Perlin noise; noise_std given: b is the
    multi-scale noise field of StructNoise (needs b_mm, a_mm)."""
    _need_gpu(a, b, w, seg, w_mm, b_mm, a_mm)
    n = int(w.numel())
    for t_ in (a, b, seg):
        if t_ is not None and (t_.numel() != n or t_.dtype != F32):
            raise ValueError("blend operands must be float32 volumes of one size")
    out = torch.empty_like(w) if want_out else None
    w_out = torch.empty_like(w) if want_weight else None
    rc = _lib.load().fsg_blend_f32(_p(a), _p(b), _p(w), n, 1 if w_mm is not None else 0, _p(w_mm), float(increase), _p(seg),
                                   1 if noise_std is not None else 0, _p(b_mm), _p(a_mm),
                                   float(noise_std if noise_std is not None else 0.0), _p(out), _p(w_out), _stream(w))
    _lib.check(rc, "fsg_blend_f32")
    return (out, w_out) if want_weight else out


def slice_noise_(slices, threshold, sigma, noise1=None, noise2=None, seed=0, stream_id=0):
    _need_gpu(slices, noise1, noise2)
    _f32(slices, "slices")
    rc = _lib.load().fsg_slice_noise_f32(_p(slices), slices.numel(), float(threshold), float(sigma), _p(noise1), _p(noise2),
                                         int(seed), int(stream_id), _stream(slices))
    _lib.check(rc, "fsg_slice_noise_f32")
    return slices


def slice_void_(slices, slice_ids, params, ylin, xlin):
    """slices (n,h,w) modified in place on rows slice_ids (int32 device); params (nvoid,7) device fp32."""
    _need_gpu(slices, slice_ids, params, ylin, xlin)
    n, h, w = (int(v) for v in slices.shape)
    nv = int(slice_ids.numel())
    if slice_ids.dtype != torch.int32 or tuple(params.shape) != (nv, 7) or ylin.numel() != h or xlin.numel() != w:
        raise ValueError("slice_void_: bad argument shapes")
    rc = _lib.load().fsg_slice_void_f32(_p(slices), h, w, _p(slice_ids), _p(params), nv, _p(ylin), _p(xlin), _stream(slices))
    _lib.check(rc, "fsg_slice_void_f32")
    return slices


def slice_sums(slices) -> torch.Tensor:
    _need_gpu(slices)
    _f32(slices, "slices")
    n = int(slices.shape[0])
    out = torch.empty(n, dtype=F32, device=slices.device)
    rc = _lib.load().fsg_slice_sums_f32(_p(slices), n, slices.numel() // n, _p(out), _stream(slices))
    _lib.check(rc, "fsg_slice_sums_f32")
    return out


NZ_BUCKET = 4096
_NZ_MODES = {">": 0, "==": 1, "!=": 2}


def nonzero_ranks(vol, op=">", value=0.0):
    """Number of voxels of `vol` (float32 / bool / uint8, any shape) satisfying `vol op value`, and a function mapping
    ranks (raster order among those voxels, int64 host tensor) to their coordinates, (k, vol.dim()) int64 on the host.
    One host sync for the count, one per selection."""
    _need_gpu(vol)
    lib, st = _lib.load(), _stream(vol)
    u8 = vol.dtype in (torch.bool, torch.uint8)
    if not u8:
        _f32(vol, "vol")
    n, mode = int(vol.numel()), _NZ_MODES[op]
    nb = (n + NZ_BUCKET - 1) // NZ_BUCKET
    counts = torch.empty(nb, dtype=torch.int32, device=vol.device)
    cnt_fn = lib.fsg_nonzero_count_u8 if u8 else lib.fsg_nonzero_count_f32
    sel_fn = lib.fsg_nonzero_select_u8 if u8 else lib.fsg_nonzero_select_f32
    _lib.check(cnt_fn(_p(vol), n, mode, float(value), _p(counts), st), "fsg_nonzero_count")
    ends = np.cumsum(counts.cpu().numpy().astype(np.int64))
    total = int(ends[-1]) if nb else 0
    shape = tuple(vol.shape)

    def select(ranks, flat_device=False):
        r = np.asarray(ranks, dtype=np.int64).reshape(-1)
        if r.size == 0:
            return torch.zeros((0,) if flat_device else (0, len(shape)), dtype=torch.int64,
                               device=vol.device if flat_device else "cpu")
        if r.min() < 0 or r.max() >= total:
            raise IndexError("rank out of range")
        b = np.searchsorted(ends, r, side="right")
        local = r - (ends[b] - counts_host[b])
        req = torch.from_numpy(np.stack([b, local]).astype(np.int32))
        d = _upload(req, vol.device)
        out = torch.empty(r.size, dtype=torch.int64, device=vol.device)
        _lib.check(sel_fn(_p(vol), n, mode, float(value), _p(d[0]), _p(d[1]), int(r.size), _p(out), _stream(vol)),
                   "fsg_nonzero_select")
        if flat_device:
            return out
        flat = out.cpu().numpy()
        if (flat < 0).any():
            raise RuntimeError("nonzero_select: volume changed between count and select")
        return torch.from_numpy(np.stack(np.unravel_index(flat, shape), -1).astype(np.int64))

    counts_host = np.diff(np.concatenate([[0], ends]))
    return total, select


def box_sum3d(v, k: int) -> torch.Tensor:
    """Zero-padded k x k x k box sum (the reference convolves with a ones kernel, padding k//2): three passes of the blur kernels
    with unit taps (exact for the small integers of a binary mask)."""
    ones = np.ones(int(k), np.float32)
    for axis in range(3):
        v = blur_axis(v.contiguous(), axis, ones)
    return v


def compact_values(values, pred, op=">", value=0.0):
    """values[pred op value] in raster order (device tensor) -- the boolean-mask gather, without the mask."""
    _need_gpu(values, pred)
    _f32(values, "values"), _f32(pred, "pred")
    n, mode = int(pred.numel()), _NZ_MODES[op]
    if values.numel() != n:
        raise ValueError("values and pred must have one size")
    lib = _lib.load()
    nb = (n + NZ_BUCKET - 1) // NZ_BUCKET
    counts = torch.empty(nb, dtype=torch.int32, device=pred.device)
    _lib.check(lib.fsg_nonzero_count_f32(_p(pred), n, mode, float(value), _p(counts), _stream(pred)), "fsg_nonzero_count")
    c = counts.cpu().numpy().astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(c)[:-1]]) if nb else np.zeros(0, np.int64)
    total = int(c.sum())
    out = torch.empty(total, dtype=F32, device=pred.device)
    if total:
        d = _upload(torch.from_numpy(offs.astype(np.int64)), pred.device)
        _lib.check(lib.fsg_compact_f32(_p(values), _p(pred), n, mode, float(value), _p(d), _p(out), _stream(pred)),
                   "fsg_compact_f32")
    return out


_EWISE = {"add": 0, "gt": 1, "eq": 2, "mul": 3, "mul_gt": 4, "max": 5, "sub_gt": 6, "le": 7}


def _ewise(op, a, b=None, value=0.0):
    _need_gpu(a, b)
    _f32(a, "a")
    if b is not None and (_f32(b, "b").numel() != a.numel()):
        raise ValueError("operands must have one size")
    out = torch.empty_like(a)
    _lib.check(_lib.load().fsg_ewise_f32(_p(a), _p(b), a.numel(), _EWISE[op], float(value), _p(out), _stream(a)),
               "fsg_ewise_f32")
    return out


def axpy(a, b):
    """a + b."""
    return _ewise("add", a, b)


def threshold(a, value=0.0):
    """(a > value) as float32 0/1."""
    return _ewise("gt", a, None, value)


def equals(a, value):
    return _ewise("eq", a, None, value)


def mul(a, b):
    return _ewise("mul", a, b)


def mask_mul(a, b, value=0.0):
    """a * (b > value)."""
    return _ewise("mul_gt", a, b, value)


def maximum(a, b):
    return _ewise("max", a, b)


def sub_gt(a, b, value=0.0):
    """((a - b) > value) as float32 0/1."""
    return _ewise("sub_gt", a, b, value)


def less_equal(a, value):
    return _ewise("le", a, None, value)


def distance_to_mask(mask, radius: int, metric: str) -> torch.Tensor:
    """Capped distance transform of a float 0/1 mask: squared Euclidean ("euclid2", exact where <= radius^2) or
    city block ("l1", exact where <= radius); three axis passes of fsg_dist_pass_f32."""
    _need_gpu(mask)
    n0, n1, n2 = _dims3(_f32(mask, "mask"))
    m = {"euclid2": 0, "l1": 1}[metric]
    lib, st = _lib.load(), _stream(mask)
    src, first = mask, 1
    for axis in (2, 1, 0):
        dst = torch.empty_like(mask)
        _lib.check(lib.fsg_dist_pass_f32(_p(src), _p(dst), n0, n1, n2, axis, int(radius), m, first, st), "fsg_dist_pass_f32")
        src, first = dst, 0
    return src


def boundary_mask(image, mask, mask_modif, mog, dist, n_dilate: int, want_mask=False):
    _need_gpu(image, mask, mask_modif, mog, dist)
    n = int(mask.numel())
    for t_ in (image, mask, mask_modif, mog, dist):
        if t_ is not None and (_f32(t_).numel() != n):
            raise ValueError("boundary_mask operands must have one size")
    out = torch.empty_like(mask) if image is not None else None
    mo = torch.empty_like(mask) if want_mask else None
    rc = _lib.load().fsg_boundary_mask_f32(_p(image), _p(mask), _p(mask_modif), _p(mog), _p(dist), int(n_dilate), n, _p(out),
                                           _p(mo), _stream(mask))
    _lib.check(rc, "fsg_boundary_mask_f32")
    return (out, mo) if want_mask else out


def bernoulli_keep(a, p: float, seed: int, stream_id: int = 0):
    _need_gpu(a)
    out = torch.empty_like(_f32(a))
    _lib.check(_lib.load().fsg_bernoulli_keep_f32(_p(a), a.numel(), float(p), int(seed), int(stream_id), _p(out), _stream(a)),
               "fsg_bernoulli_keep_f32")
    return out


def scatter_ones(shape, flat_idx, device):
    """Zero float volume of `shape` with ones at the flat (int64, device) indices."""
    _need_gpu(flat_idx)
    out = torch.zeros(tuple(shape), dtype=F32, device=device)
    if flat_idx.numel():
        _lib.check(_lib.load().fsg_scatter_const_f32(_p(out), out.numel(), _p(flat_idx.contiguous()), int(flat_idx.numel()),
                                                     1.0, _stream(out)), "fsg_scatter_const_f32")
    return out
