"""Host-side parameter preparation: per-axis sample tables, Gaussian taps, parameter arena.

These are the few-hundred-byte to few-KB inputs of the kernels.  They are computed on the host with
the same torch / numpy calls the reference uses for the same quantities, so the sample positions and
weights the kernels consume are bit-identical to the reference's:

  * zoom tables     -- utils/generation.py:315-363 (`torch.arange(..., dtype=float32)[:n]`, clamp, floor)
  * resample tables -- generator/augmentation/synthseg.py:84-102 (float64 `np.arange`, cast to fp32) +
                       the validity test / split of utils/generation.py:228-256
  * Gaussian taps   -- utils/generation.py:74-81
"""
from __future__ import annotations

from functools import lru_cache

import os
import threading

import numpy as np
import torch

from . import _lib

TAP_DTYPE = np.dtype([("lo", "<i4"), ("hi", "<i4"), ("w_lo", "<f4"), ("w_hi", "<f4")])


def _pack(lo, hi, w_lo, w_hi) -> np.ndarray:
    t = np.empty(len(lo), dtype=TAP_DTYPE)
    t["lo"], t["hi"], t["w_lo"], t["w_hi"] = lo, hi, w_lo, w_hi
    return t


@lru_cache(maxsize=1024)
def zoom_table(n_src: int, factor: float, n_dst: int) -> np.ndarray:
    """One axis of the separable linear zoom: n_dst samples of an n_src-long axis.
    Cached: the same (read-only) array object is returned for the same arguments, which also lets the
    device-side copy be reused (kernels.DeviceTables)."""
    delta = (1.0 - factor) / (2.0 * factor)
    pos = torch.arange(delta, delta + n_dst / factor, 1 / factor, dtype=torch.float32)[:n_dst]
    pos = pos.clamp_(min=0).clamp_(max=n_src - 1)
    lo = torch.floor(pos).to(torch.int32)
    hi = torch.clamp(lo + 1, max=n_src - 1)
    w_hi = pos - lo
    w_lo = 1 - w_hi
    tab = _pack(lo.numpy(), hi.numpy(), w_lo.numpy(), w_hi.numpy())
    tab.setflags(write=False)
    return tab


def zoom_tables(src_shape, factor):
    """(tables for x,y,z, destination shape) for `zoom(src, factor)`."""
    factor = np.asarray(factor, dtype=np.float64)
    new = np.round(np.asarray(src_shape[:3]) * factor).astype(int)
    tabs = [zoom_table(int(src_shape[a]), float(factor[a]), int(new[a])) for a in range(3)]
    return tabs, tuple(int(v) for v in new)


@lru_cache(maxsize=4096)
def zoom_tables_between(src_shape: tuple, dst_shape: tuple, inverse_of_down: bool = False):
    """`zoom_tables` for a factor that is a function of the two shapes alone, memoised on the shapes (three calls per sample):
    factor = dst / src (coarse grid -> volume: the deformation field, the bias field), or, with `inverse_of_down`,
    1 / (src / dst) (RandResample's zoom-back: the reciprocal of the down-sampling factors new_size / size)."""
    src = np.asarray(src_shape)
    dst = np.asarray(dst_shape)
    factor = 1 / (src / dst) if inverse_of_down else dst / src
    return zoom_tables(src_shape, factor)


def position_table(pos64: np.ndarray, n_src: int) -> np.ndarray:
    """Table for explicit float64 sample positions along one axis (axis-aligned trilinear gather):
    positions are rounded to fp32 first; outside (0, n_src-1] is marked lo = -1."""
    p = torch.tensor(np.asarray(pos64), dtype=torch.float32)
    ok = (p > 0) & (p <= n_src - 1)
    lo = torch.floor(p).to(torch.int32)
    hi = torch.clamp(lo + 1, max=n_src - 1)
    w_hi = p - lo
    w_lo = 1 - w_hi
    lo = torch.where(ok, lo, torch.full_like(lo, -1))
    hi = torch.where(ok, hi, torch.zeros_like(hi))
    return _pack(lo.numpy(), hi.numpy(), w_lo.numpy(), w_hi.numpy())


_LOG5 = np.log(5)


def resample_plan(in_shape, resolution, spacing, u_std: float):
    """Blur sigmas, low-res size, factors and per-axis tables of RandResample."""
    spacing = np.asarray(spacing, dtype=np.float64)
    resolution = np.asarray(resolution, dtype=np.float64)
    size = np.asarray(in_shape)
    stds = (0.85 + 0.3 * u_std) * _LOG5 / np.pi * spacing / resolution  # reference's association order
    stds[spacing <= resolution] = 0.0
    new_size = (size * resolution / spacing).astype(int)
    factors = new_size / size
    ns, sz = new_size.tolist(), size.tolist()
    tabs = [_resample_axis_table(ns[a], sz[a]) for a in range(3)]
    return stds, tuple(ns), factors, tabs


@lru_cache(maxsize=1024)
def _resample_axis_table(n_new: int, n_src: int) -> np.ndarray:
    """Sample positions of RandResample along one axis depend only on (new size, size)."""
    factor = np.float64(n_new) / np.float64(n_src)  # == (new_size / size)[a]
    delta = (1.0 - factor) / (2.0 * factor)
    pos = np.arange(delta, delta + n_new / factor, 1 / factor)[:n_new]
    tab = position_table(pos, n_src)
    tab.setflags(write=False)
    return tab


@lru_cache(maxsize=64)
def _tap_positions(half: int) -> torch.Tensor:
    return torch.linspace(-half, half, 2 * half + 1, dtype=torch.float32)


@lru_cache(maxsize=256)
def gaussian_taps(sigma: float) -> np.ndarray:
    t = _tap_positions(int(np.ceil(3 * sigma)))
    g = torch.exp(-((t / sigma) ** 2) / 2)
    return (g / g.sum()).numpy()


_PIN = None


def _can_pin() -> bool:
    global _PIN
    if _PIN is None:  # torch.cuda.is_available() is slow (environment lookups): ask once
        _PIN = bool(torch.cuda.is_available())
    return _PIN


class _StagingRing:
    """Pinned host slots for the parameter arenas of consecutive samples.  A slot is rewritten by the host only after an
    event recorded behind its copy kernel has completed (the host runs ahead of the GPU by design).  An event record is a
    barrier packet in the launch queue (~5.5 us of bubble), so ONE event covers a group of GROUP consecutive slots: it is
    recorded behind the copy of the group's last slot, and a slot of that group is reused (a full lap later) only once it
    has completed."""

    SLOT = 1 << 16
    SLOTS = 32
    GROUP = 8

    def __init__(self, device):
        self.device = torch.device(device)
        self.buf = torch.empty((self.SLOTS, self.SLOT), dtype=torch.uint8, pin_memory=True)
        self.events = [None] * (self.SLOTS // self.GROUP)
        self.next = 0
        self.lock = threading.Lock()

    def raw_stream(self):
        idx = self.device.index
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)

    def acquire(self):
        with self.lock:
            slot = self.next
            self.next = (slot + 1) % self.SLOTS
        ev = self.events[slot // self.GROUP]  # recorded behind this group's last copy of the previous lap
        if ev is not None:
            ev.synchronize()
        return slot, self.buf[slot]

    def release(self, slot):
        if slot % self.GROUP != self.GROUP - 1:
            return
        g = slot // self.GROUP
        ev = self.events[g]
        if ev is None:
            ev = self.events[g] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))


_RINGS: dict = {}


def _staging_ring(device):
    if os.environ.get("FSG_ARENA_MEMCPY", "0") == "1":
        return None
    dev = torch.device(device)
    idx = torch.cuda.current_device() if dev.index is None else dev.index
    # one ring per launch stream: a group's event must sit behind every copy of the group
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    ring = _RINGS.get(key)
    if ring is None:
        ring = _RINGS[key] = _StagingRing(dev)  # kept for the life of the process: copies may still be reading it
    return ring


class Arena:
    """Packs many small host arrays into ONE pinned buffer and uploads them with one async copy.

    Pointers handed to the C ABI are `base + offset`.  The device tensor must stay alive until the
    kernels that read it have been enqueued on the same stream (stream-ordered allocator semantics)."""

    ALIGN = 256

    def __init__(self):
        self._items = []
        self._size = 0

    def add(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        off = self._size
        self._items.append((off, arr))
        self._size = (off + arr.nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        return off

    def upload(self, device) -> torch.Tensor:
        size = max(self._size, self.ALIGN)
        ring = _staging_ring(device) if (_can_pin() and size <= _StagingRing.SLOT) else None
        if ring is not None:
            # copy *kernel* on the launch stream from a device-visible pinned slot: the sample stays in one hardware queue
            # (a memcpy command between the previous sample's kernels and this one's costs ~10 us of queue hand-over)
            slot, host = ring.acquire()
            hv = host.numpy()
            for off, arr in self._items:
                hv[off : off + arr.nbytes] = arr.view(np.uint8).reshape(-1)
            n = (size + 15) // 16 * 16
            self.dev = torch.empty(n, dtype=torch.uint8, device=device)
            _lib.check(_lib.load().fsg_copy_bytes(self.dev.data_ptr(), host.data_ptr(), n, ring.raw_stream()), "fsg_copy_bytes")
            ring.release(slot)
        else:
            host = torch.empty(size, dtype=torch.uint8, pin_memory=_can_pin())
            hv = host.numpy()
            for off, arr in self._items:
                hv[off : off + arr.nbytes] = arr.view(np.uint8).reshape(-1)
            self.dev = host.to(device, non_blocking=True)
        self.base = self.dev.data_ptr()
        self.__dict__.pop("_f32", None)
        return self.dev

    def stage(self, device, dev_tensor) -> bool:
        """Deferred form of `upload`: the items go into a pinned slot and `dev_tensor` (uint8, device, at least the arena's
        size) becomes the arena's device block, but NO copy is enqueued -- the native call that consumes the arena does it
        (fsg_sample_plan::arena_host / arena_dev / arena_bytes), possibly on its side stream.  `flush` must follow.
        False (nothing staged) when the arena does not fit a slot or pinned memory is unavailable."""
        size = max(self._size, self.ALIGN)
        n = (size + 15) // 16 * 16
        if dev_tensor is None or dev_tensor.numel() < n or n > _StagingRing.SLOT or not _can_pin():
            return False
        ring = _staging_ring(device)
        if ring is None:
            return False
        slot, host = ring.acquire()
        hv = host.numpy()
        for off, arr in self._items:
            hv[off : off + arr.nbytes] = arr.view(np.uint8).reshape(-1)
        self.dev = dev_tensor
        self.base = dev_tensor.data_ptr()
        self.__dict__.pop("_f32", None)
        self.pending = (ring, slot, host.data_ptr(), n)
        return True

    pending = None

    def flush(self, copied: bool):
        """After `stage`: `copied` = a native call has enqueued the copy (and ordered the current stream behind it);
        otherwise it is enqueued here, on the current stream.  Then the pinned slot is handed back to the ring."""
        ring, slot, host_ptr, n = self.pending
        self.pending = None
        if not copied:
            _lib.check(_lib.load().fsg_copy_bytes(self.dev.data_ptr(), host_ptr, n, ring.raw_stream()), "fsg_copy_bytes")
        ring.release(slot)

    def ptr(self, off: int) -> int:
        return self.base + off

    def f32(self, off: int, shape) -> torch.Tensor:
        """float32 view of an uploaded item (offsets are multiples of ALIGN)."""
        f = self.__dict__.get("_f32")
        if f is None:
            f = self._f32 = self.dev.view(torch.float32)
        n = 1
        for v in shape:
            n *= int(v)
        return f[off >> 2 : (off >> 2) + n].view(tuple(shape))
