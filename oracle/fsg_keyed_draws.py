"""ORACLE side of keyed mode -- numpy restatement of what `csrc/fsg_keyed.hip` draws for a sample key.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as oracle/fsg_oracle.py: only tests/, smoke() and bench.py's
cpu_baseline leg may import it).

Keyed mode has no counterpart in the reference (which draws from numpy's / torch's global generators);
what it shares with the reference is the ARITHMETIC from a uniform / normal draw to a parameter.  That arithmetic is
restated here from the reference's lines (cited per block, paths relative to /root/reference/fetalsyngen/), on top of an
independent numpy Philox4x32-10, so that tests can hold the C draws to it:

  * scalars (`host_draws`): bit-exact up to libm (cos / sin / exp / log of numpy vs glibc: <= 2 ulp, the tests say so);
  * small device tensors (`device_normals`, `device_uniforms`, `gmm_tables`): the integer stream is exact; the normals go
    through the GPU's v_log / v_sqrt / v_sin / v_cos approximations, so they agree to ~1e-6 relative, not bitwise.

Parity status: the Philox core is pinned by the published Random123 known-answer vectors (tests/test_keyed_draws.py); the
parameter arithmetic by the same golden fixtures as oracle/fsg_oracle.py (it is the same arithmetic).
"""
from __future__ import annotations

import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11; Random123), vectorised over the counter words (uint64 arrays holding 32-bit
    values).  Returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & MASK), np.uint64(k1 & MASK)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        h0, l0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        h1, l1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0 = (k0 + np.uint64(W0)) & np.uint64(MASK)
        k1 = (k1 + np.uint64(W1)) & np.uint64(MASK)
    return c0, c1, c2, c3


def _block(key: int, stream: int, blk):
    blk = np.asarray(blk, dtype=np.uint64)
    return philox4x32_10(blk & MASK, blk >> np.uint64(32), stream & MASK, stream >> 32, key & MASK, key >> 32)


def slot_u(key: int, slot: int) -> float:
    """Host scalar slot: uniform double in [0,1) from 53 bits of words (x, y) of counter (slot, 0, 0, 0)."""
    x, y, _z, _w = (int(v) for v in _block(key, 0, slot))
    return float(((x << 32 | y) >> 11) * (1.0 / 9007199254740992.0))


def slot_n(key: int, slot: int) -> float:
    x, y, z, w = (int(v) for v in _block(key, 0, slot))
    u1 = (((x << 32 | y) >> 11) + 1) * (1.0 / 9007199254740992.0)
    u2 = ((z << 32 | w) >> 11) * (1.0 / 9007199254740992.0)
    return float(np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925286766559 * u2))


S = dict(SUB0=0, DEFORM=4, FLIP=5, ROT=6, SHEAR=9, SCALE=12, SHIFT=15, NL_SCALE=18, NL_STD=19, GAMMA_GATE=20, GAMMA=21,
         BIAS_GATE=22, BF_SCALE=23, BF_STD=24, RES_GATE=25, SPACING=26, RES_STD=27, NOISE_GATE=28, NOISE_STD=29)


def affine_matrix(rot, shear, scale) -> np.ndarray:
    """utils/generation.py:39-71, float64."""
    cx, cy, cz = np.cos(rot[0]), np.cos(rot[1]), np.cos(rot[2])
    sx, sy, sz = np.sin(rot[0]), np.sin(rot[1]), np.sin(rot[2])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    shx = np.array([[1, 0, 0], [shear[1], 1, 0], [shear[2], 0, 1]])
    shy = np.array([[1, shear[0], 0], [0, 1, 0], [0, shear[2], 1]])
    shz = np.array([[1, 0, shear[0]], [0, 1, shear[1]], [0, 0, 1]])
    return (shx @ shy @ shz @ rx @ ry @ rz) * np.asarray(scale, dtype=np.float64).reshape(3, 1)


def host_draws(cfg: dict, key: int) -> dict:
    """Every scalar of sample `key` under configuration `cfg` (keys as fsg_keyed_config in include/fsg_hip.h)."""
    u = lambda name, off=0: slot_u(key, S[name] + off)  # noqa: E731
    shape, size, res = np.array(cfg["shape"]), np.array(cfg["size"]), np.array(cfg["resolution"], dtype=np.float64)
    d = {"key": key}
    lo, hi = cfg["min_subclusters"], cfg["max_subclusters"]
    # rand_gmm.py:82-85: one integer in [lo, hi] per meta label
    d["subclusters"] = [min(lo + int(u("SUB0", m) * (hi - lo + 1)), hi) for m in range(cfg["meta_labels"])]
    # affine_nonrigid.py:140-145, :248-263, :271-290, :303-317
    d["deform_active"] = u("DEFORM") < cfg["deform_prob"]
    if d["deform_active"]:
        d["flip"] = u("FLIP") < cfg["flip_prb"]
        mr, ms, mc = cfg["max_rotation"], cfg["max_shear"], cfg["max_scaling"]
        d["rotations"] = np.array([(2 * mr * u("ROT", a) - mr) / 180.0 * np.pi for a in range(3)])
        d["shears"] = np.array([2 * ms * u("SHEAR", a) - ms for a in range(3)])
        d["scalings"] = np.array([1 + (2 * mc * u("SCALE", a) - mc) for a in range(3)])
        d["A"] = affine_matrix(d["rotations"], d["shears"], d["scalings"]).astype(np.float32)
        centre = ((shape - 1) / 2).astype(np.float32).astype(np.float64)
        room = np.maximum((shape - size).astype(np.float32) / np.float32(2), np.float32(0)).astype(np.float64)
        d["c2"] = centre + (2 * (room * np.array([u("SHIFT", a) for a in range(3)])) - room)
        d["nonlinear"] = bool(cfg["nonlinear"])
        if d["nonlinear"]:
            d["nonlin_scale"] = cfg["nonlin_scale_min"] + u("NL_SCALE") * (cfg["nonlin_scale_max"] - cfg["nonlin_scale_min"])
            d["field_dims"] = np.round(d["nonlin_scale"] * shape).astype(int).tolist()
            d["nonlin_std"] = cfg["nonlin_std_max"] * u("NL_STD")
    # synthseg.py:263-268
    d["gamma_active"] = u("GAMMA_GATE") < cfg["gamma_prob"]
    if d["gamma_active"]:
        d["gamma"] = float(np.exp(cfg["gamma_std"] * slot_n(key, S["GAMMA"])))
    # synthseg.py:157-170
    d["bias_active"] = u("BIAS_GATE") < cfg["bias_prob"]
    if d["bias_active"]:
        d["bf_scale"] = cfg["bf_scale_min"] + u("BF_SCALE") * (cfg["bf_scale_max"] - cfg["bf_scale_min"])
        d["bias_dims"] = np.maximum(np.round(d["bf_scale"] * shape).astype(int), 1).tolist()
        d["bf_std"] = cfg["bf_std_min"] + (cfg["bf_std_max"] - cfg["bf_std_min"]) * u("BF_STD")
    # synthseg.py:63-84
    d["resample_active"] = u("RES_GATE") < cfg["resample_prob"]
    d["low_shape"] = shape.tolist()
    if d["resample_active"]:
        d["spacing"] = cfg["min_resolution"] + (cfg["max_resolution"] - cfg["min_resolution"]) * u("SPACING")
        d["u_std"] = u("RES_STD")
        spacing = np.array([1.0, 1.0, 1.0]) * d["spacing"]
        stds = (0.85 + 0.3 * d["u_std"]) * np.log(5) / np.pi * spacing / res
        stds[spacing <= res] = 0.0
        d["stds"] = stds
        d["low_shape"] = (shape * res / spacing).astype(int).tolist()
    # synthseg.py:218-223
    d["noise_active"] = u("NOISE_GATE") < cfg["noise_prob"]
    if d["noise_active"]:
        d["noise_std"] = cfg["noise_std_min"] + (cfg["noise_std_max"] - cfg["noise_std_min"]) * u("NOISE_STD")
    return d


# ---- device streams (csrc/fsg_common.h: fsg_randn4; csrc/fsg_keyed.hip: keyed_uniform) -------------------------------------
def device_uniforms(key: int, stream: int, n: int) -> np.ndarray:
    """Element e = word e % 4 of block e // 4, top 24 bits, [0,1) float32 (torch.rand's grid)."""
    e = np.arange(n, dtype=np.uint64)
    words = np.stack(_block(key, stream, e >> np.uint64(2)), axis=-1)
    w = words[np.arange(n), (e & np.uint64(3)).astype(np.int64)]
    return ((w >> np.uint64(8)).astype(np.float32) * np.float32(5.9604644775390625e-08)).astype(np.float32)


def device_normals(key: int, stream: int, n: int) -> np.ndarray:
    """Box-Muller as fsg_randn4 writes it: (x, z) -> radii through -2 ln u with u in (0,1], (y, w) -> angles in
    revolutions; element order (r0 cos, r0 sin, r1 cos, r1 sin) per block.  float32 arithmetic, numpy's libm."""
    nb = (n + 3) // 4
    x, y, z, w = _block(key, stream, np.arange(nb, dtype=np.uint64))
    s = np.float32(5.9604644775390625e-08)
    u0 = ((x >> np.uint64(8)) + np.uint64(1)).astype(np.float32) * s
    u1 = ((z >> np.uint64(8)) + np.uint64(1)).astype(np.float32) * s
    t0 = (y >> np.uint64(8)).astype(np.float32) * s
    t1 = (w >> np.uint64(8)).astype(np.float32) * s
    r0 = np.sqrt(np.float32(-2.0) * np.log(u0, dtype=np.float32), dtype=np.float32)
    r1 = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
    two_pi = np.float32(6.2831853071795864)
    out = np.stack([r0 * np.cos(two_pi * t0, dtype=np.float32), r0 * np.sin(two_pi * t0, dtype=np.float32),
                    r1 * np.cos(two_pi * t1, dtype=np.float32), r1 * np.sin(two_pi * t1, dtype=np.float32)], axis=-1)
    return out.reshape(-1)[:n].astype(np.float32)


def gmm_tables(cfg: dict, key: int):
    """rand_gmm.py:120-145 on the keyed streams: mus = 25 + 200 U, sigmas = 5 + 20 U (stream 5), class-tied means
    mus[seed_labels] = clamp(mus[generation_classes] + 25 z, 0, 225) with z from stream 6 (right-hand side first)."""
    nl = cfg["nlabels"]
    uu = device_uniforms(key, 5, 2 * nl)
    mus = np.float32(25) + np.float32(200) * uu[:nl]
    sigmas = np.float32(5) + np.float32(20) * uu[nl:]
    if cfg["tie_classes"]:
        sl, gc = np.asarray(cfg["seed_labels"]), np.asarray(cfg["generation_classes"])
        z = device_normals(key, 6, len(sl))
        tied = mus[gc] + np.float32(25) * z
        mus[sl] = np.minimum(np.maximum(tied, np.float32(0)), np.float32(225))
    return mus, sigmas
