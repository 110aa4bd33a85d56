"""Keyed mode (`rng="keyed"`): host side of `fsg_keyed_*` (include/fsg_hip.h, csrc/fsg_keyed.hip).

The reference's generator draws ~30 scalars and three small tensors per sample from numpy's / torch's global generators,
one interpreter round trip each; replaying that tape costs the host ~230 us per 256^3 sample -- as much
as the GPU needs for the whole sample.  In keyed mode a sample is a pure function of its 64-bit key
(`sharding.sample_key(base_seed, index)`): every draw comes from Philox4x32-10 under that key, the scalars in C inside
ONE native call, the small tensors on the device.  Python hands over pointers and the key, and turns the exported draws
(`fsg_keyed_draws`) into the reference's `synth_params` dictionary.

Distributions and the arithmetic from draw to parameter are the reference's; the random numbers are not (by design -- use
`rng="reference"` for same-seed parity with a CPU run of the reference).
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from . import kernels as K
from . import seedcodes
from . import tables as T


def config_of(gen, shape) -> _lib.KeyedConfig:
    """fsg_keyed_config of a FetalSynthGen (its stage objects hold the reference's YAML keys)."""
    ig, sd, bf, rs, nz, gm = (gen.intensity_generator, gen.spatial_deform, gen.biasfield, gen.resampled, gen.noise, gen.gamma)
    c = _lib.KeyedConfig()
    c.shape[:] = [int(v) for v in shape]
    c.size[:] = [int(v) for v in sd.size]
    c.resolution[:] = [float(v) for v in gen.resolution]
    c.min_subclusters, c.max_subclusters, c.meta_labels = int(ig.min_subclusters), int(ig.max_subclusters), int(ig.meta_labels)
    c.nlabels = max(ig.seed_labels) + 1
    c.n_seed_labels = len(ig.seed_labels)
    c.tie_classes = int(ig.generation_classes != ig.seed_labels)
    if c.nlabels > 256 or c.n_seed_labels > 256:
        raise ValueError("keyed mode takes label values < 256")
    for j, (a, b) in enumerate(zip(ig.seed_labels, ig.generation_classes)):
        c.seed_labels[j], c.generation_classes[j] = int(a), int(b)
    c.deform_prob, c.flip_prb = float(sd.prob), float(sd.flip_prb)
    c.max_rotation, c.max_shear, c.max_scaling = float(sd.max_rotation), float(sd.max_shear), float(sd.max_scaling)
    c.nonlinear = int(bool(sd.nonlinear_transform))
    c.nonlin_scale_min, c.nonlin_scale_max, c.nonlin_std_max = float(sd.nonlin_scale_min), float(sd.nonlin_scale_max), float(sd.nonlin_std_max)
    c.gamma_prob, c.gamma_std = float(gm.prob), float(gm.gamma_std)
    c.bias_prob, c.bf_scale_min, c.bf_scale_max = float(bf.prob), float(bf.scale_min), float(bf.scale_max)
    c.bf_std_min, c.bf_std_max = float(bf.std_min), float(bf.std_max)
    c.resample_prob, c.min_resolution, c.max_resolution = float(rs.prob), float(rs.min_resolution), float(rs.max_resolution)
    c.noise_prob, c.noise_std_min, c.noise_std_max = float(nz.prob), float(nz.std_min), float(nz.std_max)
    return c


def config_dict(cfg: _lib.KeyedConfig) -> dict:
    """The configuration as plain Python values (what tests hand to the oracle's restatement)."""
    out = {}
    for name, _t in cfg._fields_:
        v = getattr(cfg, name)
        out[name] = list(v) if hasattr(v, "__len__") else v
    out["seed_labels"] = out["seed_labels"][: cfg.n_seed_labels]
    out["generation_classes"] = out["generation_classes"][: cfg.n_seed_labels]
    return out


class KeyedContext:
    """One `fsg_keyed_ctx` (host-only object) for a (generator configuration, volume shape) pair."""

    def __init__(self, gen, shape):
        self.lib = _lib.load()
        self.shape = tuple(int(v) for v in shape)
        self.cfg = config_of(gen, self.shape)
        self.device = gen.device
        h = C.c_void_p()
        _lib.check(self.lib.fsg_keyed_create(C.byref(self.cfg), C.byref(h)), "fsg_keyed_create")
        self.handle = h
        self.block_bytes = int(self.lib.fsg_keyed_block_bytes(h))
        self.iv = np.zeros(_lib.KEYED_I["COUNT"], dtype=np.int64)
        self.ivp = self.iv.ctypes.data
        self._subjects = {}
        self.use_codes = True  # the subject's seed volumes as one uint16 code volume (seedcodes.py, built on first use)
        self._tables_ready = False
        self._keep = []  # device tables registered with the context
        # rows of the per-(x,y) coarse workspace the largest grids need (3 * field_dims[2] + bias_dims[2])
        f2 = int(np.round(self.cfg.nonlin_scale_max * self.shape[2])) if self.cfg.nonlinear else 0
        b2 = max(int(np.round(self.cfg.bf_scale_max * self.shape[2])), 1)
        self.rows_need = 3 * f2 + b2

    def close(self):
        if self.handle:
            self.lib.fsg_keyed_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables: built by the same cached builders as the other modes, registered by device pointer ----------------------
    def _register(self, kind, axis, n, tab):
        d = K._device_table(tab, self.device)
        self._keep.append(d)
        _lib.check(self.lib.fsg_keyed_set_table(self.handle, kind, axis, int(n), C.c_void_p(d.data_ptr())), "fsg_keyed_set_table")

    def register_tables(self):
        """Every tap table a sample of this configuration can ask for (bounded: the low-res size takes at most
        size * (1 - min / max resolution) values per axis, the coarse grids a handful)."""
        if self._tables_ready:
            return
        c = self.cfg
        for a in range(3):
            size = self.shape[a]
            lo = int(size * c.resolution[a] / max(c.max_resolution, c.resolution[a]))
            hi = int(size * c.resolution[a] / c.min_resolution)
            for m in range(max(lo - 1, 1), min(max(hi, lo) + 1, 4 * size) + 1):
                self._register(_lib.KT_RESAMPLE, a, m, T._resample_axis_table(m, size))
                f = np.float64(m) / np.float64(size)
                self._register(_lib.KT_BACK, a, m, T.zoom_table(m, float(1 / f), int(np.round(m * (1 / f)))))
            if c.nonlinear:
                for s_ in range(max(int(np.floor(c.nonlin_scale_min * size)) - 1, 1), int(np.ceil(c.nonlin_scale_max * size)) + 2):
                    self._register(_lib.KT_FIELD, a, s_, T.zoom_table(s_, float(np.float64(size) / np.float64(s_)), size))
            for s_ in range(max(int(np.floor(c.bf_scale_min * size)) - 1, 1), int(np.ceil(c.bf_scale_max * size)) + 2):
                self._register(_lib.KT_BIAS, a, s_, T.zoom_table(s_, float(np.float64(size) / np.float64(s_)), size))
        self._tables_ready = True

    # ---- per-subject pointer block --------------------------------------------------------------------------------------
    def subject(self, bank, seg, twin):
        """int64 pointers of one subject's label volumes (bank slots, float32 segmentation, its uint8 twin), validated once
        per (bank, segmentation) object pair -- the C side only sees addresses."""
        key = (id(bank), id(seg))
        hit = self._subjects.get(key)
        if hit is not None and hit[0]() is bank and hit[1]() is seg and hit[2] == seg._version:
            ent = hit[3]
            if twin is not None and ent[1] == 0:
                ent[1] = twin.data_ptr()
            self._codes(bank, ent)  # (built once per bank object; a rewrite of a seed volume through torch rebuilds it)
            return ent
        c, shape = self.cfg, self.shape
        dev = torch.device(self.device)
        if tuple(seg.shape) != shape or seg.dtype != torch.float32 or not seg.is_cuda or not seg.is_contiguous():
            raise ValueError(f"segmentation: expected a contiguous float32 tensor of shape {shape} on {dev}")
        ptrs = np.zeros(64, dtype=np.int64)
        vol = bank.vol
        for n in range(c.min_subclusters, c.max_subclusters + 1):
            for m in range(1, c.meta_labels + 1):
                part = vol[n][m]
                off_dev = part.device.type != dev.type or (dev.index is not None and part.device.index != dev.index)
                if tuple(part.shape) != shape or part.dtype != torch.uint8 or not part.is_contiguous() or off_dev:
                    raise ValueError(f"seed volume ({n}, {m}): expected a contiguous uint8 tensor of shape {shape} on {dev}, "
                                     f"got {part.dtype} {tuple(part.shape)} on {part.device}")
                ptrs[4 * (n - c.min_subclusters) + (m - 1)] = part.data_ptr()
        ent = [ptrs, 0 if twin is None else twin.data_ptr(), seg.data_ptr(), 0, 0, 0, 0]  # .. codes, tuples, ntuples, stride
        self._codes(bank, ent)  # ~1 ms once per bank object (one pass over its volumes), little next to loading the subject
        if len(self._subjects) > 4096:
            self._subjects.clear()
        self._subjects[key] = (weakref.ref(bank), weakref.ref(seg), seg._version, ent)
        return ent

    def _codes(self, bank, ent):
        """ent[3:7] = the subject's code volume (seedcodes.build), kept ON the bank object so that it lives and dies with it.
        A seed volume rewritten through torch bumps its `_version`: the codes are rebuilt; a rewrite through a raw pointer needs
        `FetalSynthGen.invalidate_label_twins()` (which drops them)."""
        if not self.use_codes:
            ent[3:7] = [0, 0, 0, 0]
            return
        c = self.cfg
        vol = bank.vol
        parts = [vol[n][m] for n in range(c.min_subclusters, c.max_subclusters + 1) for m in range(1, c.meta_labels + 1)]
        ver = sum(p._version for p in parts)
        have = getattr(bank, "_seed_codes", None)
        if have is None or have[0] != ver or have[1] != self.shape:
            cols = []
            for n in range(c.min_subclusters, c.max_subclusters + 1):
                for m in range(1, 5):
                    cols.append(vol[n][m] if m <= c.meta_labels else None)
            stride = len(cols) + 1
            zero = None
            dense = []
            for col in cols:
                if col is None:
                    zero = torch.zeros_like(parts[0]) if zero is None else zero
                    col = zero
                dense.append(col)
            built = seedcodes.build_device(dense, stride) if (parts[0].numel() % 8 == 0 and parts[0].numel() <= (1 << 30)) else None
            have = (ver, self.shape, built, stride)
            bank._seed_codes = have
        built = have[2]
        if built is None:
            ent[3:7] = [0, 0, 0, 0]
        else:
            ent[3:7] = [built[0].data_ptr(), built[1].data_ptr(), int(built[1].shape[0]), int(have[3])]

    def draws(self, key: int) -> _lib.KeyedDraws:
        d = _lib.KeyedDraws()
        _lib.check(self.lib.fsg_keyed_draw(self.handle, C.c_uint64(key & 0xFFFFFFFFFFFFFFFF), C.byref(d)), "fsg_keyed_draw")
        return d


def params_of(d: _lib.KeyedDraws, block: torch.Tensor | None) -> dict:
    """The reference's `synth_params` dictionary (generator/model.py:231-276: selected_seeds, seed_intensities,
    deform_params, gamma_params, bf_params, resample_params, noise_params, artifacts) from exported draws.  `block`: the
    sample's device parameter block (mus / sigmas are views of it, device tensors as in the reference)."""
    m2s = {m + 1: int(d.subclusters[m]) for m in range(4) if d.subclusters[m]}
    si = {}
    if block is not None:
        f = block.view(torch.float32)
        si = {"mus": f[d.off_mus >> 2:(d.off_mus >> 2) + d.ntab], "sigmas": f[d.off_sigmas >> 2:(d.off_sigmas >> 2) + d.ntab]}
    if d.deform_active:
        nr = {}
        if d.nonlinear:
            nr = {"nonlin_scale": np.array([d.nonlin_scale]), "nonlin_std": d.nonlin_std, "size_F_small": list(d.field_dims)}
        dp = {"affine": {"rotations": np.array(d.rotations), "shears": np.array(d.shears), "scalings": np.array(d.scalings)},
              "non_rigid": nr, "flip": bool(d.flip)}
    else:
        dp = {"affine": None, "non_rigid": None, "flip": False}
    if d.bias_active:
        bp = {"bf_scale": np.array([d.bf_scale]), "bf_std": np.array([d.bf_std]), "bf_size": list(d.bias_dims)}
    else:
        bp = {"bf_scale": None, "bf_std": None, "bf_size": None}
    return {
        "selected_seeds": {"mlabel2subclusters": m2s},
        "seed_intensities": si,
        "deform_params": dp,
        "gamma_params": {"gamma": d.gamma if d.gamma_active else None},
        "bf_params": bp,
        "resample_params": {"spacing": [d.spacing] * 3 if d.resample_active else None},
        "noise_params": {"noise_std": float(d.noise_std32) if d.noise_active else None},
        "artifacts": {},
        "key": int(d.key),
    }
