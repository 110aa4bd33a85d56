"""ORACLE (part 2) -- CPU restatement of the SR-artifact slice-stack simulation.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as oracle/fsg_oracle.py: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it).

The reference has TWO implementations of slice acquisition with different arithmetic
(paths relative to /root/reference/fetalsyngen/generator/artifacts/svort/):

  * `semantics="torch"`  -- the CPU fallback `slice_acquisition_torch` / `slice_acquisition_adjoint_torch`
    (slice_acquisition/slice_acq.py:266-546): sparse coefficient matrix, nearest voxel (round half to
    even), strict inside test `0 < p < n-1`, raw PSF values, normalisation where weight > 1e-2.
    Parity status: PINNED by tests/golden/slice_acq.npz (captured from the reference on CPU).
  * `semantics="cuda"`   -- the CUDA kernels (slice_acquisition/slice_acq_cuda_kernel.cu:17-171 forward,
    :472-670 adjoint, :672-693 equalize), what the reference runs on a GPU: `interp_psf=False` samples
    the volume trilinearly per PSF tap, `interp_psf=True` snaps to the nearest voxel (round half away
    from zero) and re-interpolates the PSF at that voxel; inside test `0 <= p < n-1`; per-pixel weight
    normalisation; the adjoint skips pixels whose weight is < 0.5.
    Parity status: UNPINNED -- CUDA cannot run in the build container, the reference holds no fixture
    for it.  Anchors: (a) with a 1x1x1 PSF the linear mode is F.grid_sample(align_corners=True) away
    from the border, which the golden `fwd_delta_nomask` (reference torch path) pins; (b) forward and
    adjoint are restated from the same loops and checked to be mutually adjoint.

Host algebra (RigidTransform, axis-angle conversions, PSF, interleaving) restated from
transform/transform.py, transform/transform_convert.py:24-161 and data/utils.py.
"""
from __future__ import annotations

from math import log, sqrt

import numpy as np
import torch

F32 = np.float32


# --------------------------------------------------------------------------------------
# PSF and scan order (data/utils.py)
# --------------------------------------------------------------------------------------
GAUSSIAN_FWHM = 1 / (2 * sqrt(2 * log(2)))
SINC_FWHM = 1.206709128803223 * GAUSSIAN_FWHM


def get_psf(r_max=None, res_ratio=(1, 1, 3), threshold=1e-4) -> torch.Tensor:
    """Gaussian PSF cropped to its support and normalised (data/utils.py:64-109, psf_type="gaussian")."""
    sx, sy, sz = SINC_FWHM * res_ratio[0], SINC_FWHM * res_ratio[1], GAUSSIAN_FWHM * res_ratio[2]
    if r_max is None:
        r_max = max(max(int(2 * r + 1) for r in (sx, sy, sz)), 4)
    x = torch.linspace(-r_max, r_max, 2 * r_max + 1, dtype=torch.float32)
    gz, gy, gx = torch.meshgrid(x, x, x, indexing="ij")
    psf = torch.exp(-0.5 * (gx**2 / sx**2 + gy**2 / sy**2 + gz**2 / sz**2))
    psf[psf.abs() < threshold] = 0
    rx = int(torch.nonzero(psf.sum((0, 1)) > 0)[0, 0])
    ry = int(torch.nonzero(psf.sum((0, 2)) > 0)[0, 0])
    rz = int(torch.nonzero(psf.sum((1, 2)) > 0)[0, 0])
    psf = psf[rz : 2 * r_max + 1 - rz, ry : 2 * r_max + 1 - ry, rx : 2 * r_max + 1 - rx].contiguous()
    return psf / psf.sum()


def interleave_index(N, n_i):
    """Acquisition order of an interleaved stack (data/utils.py:19-29)."""
    idx = [None] * N
    t = 0
    for i in range(n_i):
        for j in range(i, N, n_i):
            idx[j] = t
            t += 1
    return idx


# --------------------------------------------------------------------------------------
# rigid transform algebra (transform/transform_convert.py:24-161, transform/transform.py)
# --------------------------------------------------------------------------------------
EPS = 1e-6


def axisangle2mat(ax: torch.Tensor) -> torch.Tensor:
    """(n,6) [rotvec | t] -> (n,3,4), Rodrigues; first-order form below theta^2 <= 1e-6 (:24-85)."""
    n = ax.shape[0]
    ang, tr = ax[:, :3], ax[:, 3:]
    th2 = torch.sum(ang**2, dim=1)
    mat = torch.eye(3, 4, dtype=torch.float32).unsqueeze(0).repeat(n, 1, 1)
    m = th2 > EPS
    th = torch.sqrt(th2[m])
    u = ang[m] / th.unsqueeze(1)
    s, c = torch.sin(th), torch.cos(th)
    o = 1 - c
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    mat[m, 0, 0] = c + x * x * o
    mat[m, 0, 1] = x * y * o - z * s
    mat[m, 0, 2] = y * s + x * z * o
    mat[m, 1, 0] = z * s + x * y * o
    mat[m, 1, 1] = c + y * y * o
    mat[m, 1, 2] = -x * s + y * z * o
    mat[m, 2, 0] = -y * s + x * z * o
    mat[m, 2, 1] = x * s + y * z * o
    mat[m, 2, 2] = c + z * z * o
    a = ang[~m]
    mat[~m, 0, 1], mat[~m, 0, 2] = -a[:, 2], a[:, 1]
    mat[~m, 1, 0], mat[~m, 1, 2] = a[:, 2], -a[:, 0]
    mat[~m, 2, 0], mat[~m, 2, 1] = -a[:, 1], a[:, 0]
    mat[:, :, 3] = tr
    return mat


def mat2axisangle(mat: torch.Tensor) -> torch.Tensor:
    """(n,3,4) -> (n,6) through the 4-branch quaternion extraction (:88-161)."""
    A = mat[:, :3, :3]
    tr = A.diagonal(dim1=1, dim2=2).sum(dim=1)
    w, x, y, z = (torch.zeros_like(tr) for _ in range(4))
    d2 = A[:, 2, 2] < EPS
    d01 = A[:, 0, 0] > A[:, 1, 1]
    d0n1 = A[:, 0, 0] < -A[:, 1, 1]
    s = 2.0 * torch.sqrt(tr + 1.0)
    i = (~d2) & (~d0n1)
    w[i], x[i] = 0.25 * s[i], (A[i, 2, 1] - A[i, 1, 2]) / s[i]
    y[i], z[i] = (A[i, 0, 2] - A[i, 2, 0]) / s[i], (A[i, 1, 0] - A[i, 0, 1]) / s[i]
    s = 2.0 * torch.sqrt(A[:, 0, 0] - A[:, 1, 1] - A[:, 2, 2] + 1.0)
    i = d2 & d01
    w[i], x[i] = (A[i, 2, 1] - A[i, 1, 2]) / s[i], 0.25 * s[i]
    y[i], z[i] = (A[i, 0, 1] + A[i, 1, 0]) / s[i], (A[i, 0, 2] + A[i, 2, 0]) / s[i]
    s = 2.0 * torch.sqrt(A[:, 1, 1] - A[:, 0, 0] - A[:, 2, 2] + 1.0)
    i = d2 & (~d01)
    w[i], x[i] = (A[i, 0, 2] - A[i, 2, 0]) / s[i], (A[i, 0, 1] + A[i, 1, 0]) / s[i]
    y[i], z[i] = 0.25 * s[i], (A[i, 1, 2] + A[i, 2, 1]) / s[i]
    s = 2.0 * torch.sqrt(A[:, 2, 2] - A[:, 0, 0] - A[:, 1, 1] + 1.0)
    i = (~d2) & d0n1
    w[i], x[i] = (A[i, 1, 0] - A[i, 0, 1]) / s[i], (A[i, 0, 2] + A[i, 2, 0]) / s[i]
    y[i], z[i] = (A[i, 1, 2] + A[i, 2, 1]) / s[i], 0.25 * s[i]
    neg = w < 0
    w[neg], x[neg], y[neg], z[neg] = -w[neg], -x[neg], -y[neg], -z[neg]
    na = torch.sqrt(x**2 + y**2 + z**2)
    th = 2 * torch.atan2(na, w)
    f = torch.where(na > EPS, th / na, 2.0 / w)
    out = torch.zeros((A.shape[0], 6), dtype=A.dtype)
    out[:, 0], out[:, 1], out[:, 2] = x * f, y * f, z * f
    out[:, 3:] = mat[:, :3, 3]
    return out


def compose(mat1: torch.Tensor, mat2: torch.Tensor) -> torch.Tensor:
    """trans-first composition `self.compose(other)` (transform.py:60-70)."""
    R1, t1, R2, t2 = mat1[:, :, :3], mat1[:, :, 3:], mat2[:, :, :3], mat2[:, :, 3:]
    return torch.cat((torch.matmul(R1, R2), t2 + torch.matmul(R2.transpose(-2, -1), t1)), -1)


def mat_last2first(mat):
    R, t = mat[:, :, :3], mat[:, :, 3:]
    return torch.cat([R, torch.matmul(R.transpose(-2, -1), t)], -1)


# --------------------------------------------------------------------------------------
# slice acquisition, torch-fallback semantics (slice_acq.py:266-546) -- PINNED
# --------------------------------------------------------------------------------------
def _taps_torch(psf):
    """offsets (x,y,z) of the PSF taps with value > 0, raster (k,j,i) order, and their values (:276-277)."""
    p = np.asarray(psf, dtype=F32)
    kji = np.argwhere(p > 0)
    shape = np.array(p.shape, dtype=F32)
    off = ((kji.astype(F32) - (shape - 1) / 2) * F32(1.0))[:, ::-1]
    return np.ascontiguousarray(off), p[p > 0]


def _rot(R, v):
    """R @ v per row, fp32, products summed left to right (what the 3x3 matmul does per element)."""
    return np.stack([R[i, 0] * v[:, 0] + R[i, 1] * v[:, 1] + R[i, 2] * v[:, 2] for i in range(3)], -1).astype(F32)


def _coef_torch(tr, vol_shape, slice_shape, smask, psf, res):
    """Per masked pixel: voxel ids (-1 = outside) and PSF values of its taps (:272-310)."""
    tr = np.asarray(tr, dtype=F32)
    R, T = tr[:, :3], tr[:, 3]
    h, w = slice_shape
    off, pv = _taps_torch(psf)
    jj, ii = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    sel = np.ones((h, w), bool) if smask is None else np.asarray(smask, bool)
    y = ((jj[sel].astype(F32) - F32((h - 1) / 2)) * F32(res)).astype(F32)
    x = ((ii[sel].astype(F32) - F32((w - 1) / 2)) * F32(res)).astype(F32)
    sxyz = np.stack([x, y, np.zeros_like(x)], -1)
    sxyz = _rot(R, (sxyz + T).astype(F32))
    pxyz = _rot(R, ((off - T).astype(F32) + T).astype(F32))
    shift = ((np.array(vol_shape[::-1], dtype=F32) - 1) / F32(2.0)).astype(F32)
    pos = ((shift + pxyz[None]).astype(F32) + sxyz[:, None]).astype(F32)  # (npix, ntap, 3)
    inside = np.all((pos > 0) & (pos < shift * 2), -1)
    r = np.rint(pos).astype(np.int64)
    vid = r[..., 0] + r[..., 1] * vol_shape[2] + r[..., 2] * (vol_shape[1] * vol_shape[2])
    vid[~inside] = -1
    return sel, vid, pv


def slice_acq_forward_torch(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight):
    """`slice_acquisition_torch` (:369-429) for a PSF with more than one tap or need_weight=True."""
    vol = np.asarray(vol, dtype=F32)
    if vol_mask is not None:
        vol = (vol * np.asarray(vol_mask).astype(F32)).astype(F32)
    n = len(transforms)
    h, w = slice_shape
    out = np.zeros((n, h, w), F32)
    wgt = np.zeros((n, h, w), F32)
    flat = vol.reshape(-1)
    for s in range(n):
        sel, vid, pv = _coef_torch(transforms[s], vol.shape, slice_shape, None if slices_mask is None else slices_mask[s],
                                   psf, res_slice)
        ok = vid >= 0
        val = np.where(ok, flat[np.clip(vid, 0, None)] * pv[None], 0).astype(F32)
        out[s][sel] = val.sum(1, dtype=F32)
        wgt[s][sel] = np.where(ok, pv[None], 0).astype(F32).sum(1, dtype=F32)
    m = wgt > 1e-2
    out[m] = out[m] / wgt[m]
    if slices_mask is not None:
        out = out * np.asarray(slices_mask).astype(F32)
    return (out, wgt) if need_weight else out


def slice_acq_adjoint_torch(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, equalize):
    """`slice_acquisition_adjoint_torch` (:483-546)."""
    slices = np.asarray(slices, dtype=F32)
    if slices_mask is not None:
        slices = slices * np.asarray(slices_mask).astype(F32)
    nvox = int(np.prod(vol_shape))
    vol = np.zeros(nvox, np.float64)
    wgt = np.zeros(nvox, np.float64)
    for s in range(len(transforms)):
        sel, vid, pv = _coef_torch(transforms[s], vol_shape, slices.shape[-2:], None if slices_mask is None else slices_mask[s],
                                   psf, res_slice)
        ok = vid >= 0
        sv = slices[s][sel]
        np.add.at(vol, vid[ok], (sv[:, None] * pv[None])[ok])
        np.add.at(wgt, vid[ok], np.broadcast_to(pv[None], vid.shape)[ok])
    vol, wgt = vol.astype(F32), wgt.astype(F32)
    if equalize:
        m = wgt > 1e-2
        vol[m] = vol[m] / wgt[m]
    vol = vol.reshape(vol_shape)
    if vol_mask is not None:
        vol = vol * np.asarray(vol_mask).astype(F32)
    return vol


# --------------------------------------------------------------------------------------
# slice acquisition, CUDA-kernel semantics (slice_acq_cuda_kernel.cu) -- UNPINNED restatement
# --------------------------------------------------------------------------------------
def _cuda_geometry(transforms, vol_shape, slice_shape, res):
    tr = np.asarray(transforms, dtype=F32).reshape(-1, 3, 4)
    D, H, W = vol_shape
    h, w = slice_shape
    n = len(tr)
    ix = np.arange(w, dtype=np.float64)[None, None, :]
    iy = np.arange(h, dtype=np.float64)[None, :, None]
    res64 = np.float64(F32(res))
    t = tr.astype(np.float64)
    _x = np.broadcast_to(((ix - (w - 1) / 2.0) * res64 + t[:, 0, 3, None, None]).astype(F32), (n, h, w))
    _y = np.broadcast_to(((iy - (h - 1) / 2.0) * res64 + t[:, 1, 3, None, None]).astype(F32), (n, h, w))
    _z = np.broadcast_to(tr[:, 2, 3, None, None], (n, h, w))
    R = tr[:, :, :3]
    r = lambda i, j: R[:, i, j][:, None, None]  # noqa: E731
    xc = (r(0, 0) * _x + r(0, 1) * _y + r(0, 2) * _z).astype(F32)
    yc = (r(1, 0) * _x + r(1, 1) * _y + r(1, 2) * _z).astype(F32)
    zc = (r(2, 0) * _x + r(2, 1) * _y + r(2, 2) * _z).astype(F32)
    xc = (xc.astype(np.float64) + (W - 1) / 2.0).astype(F32)
    yc = (yc.astype(np.float64) + (H - 1) / 2.0).astype(F32)
    zc = (zc.astype(np.float64) + (D - 1) / 2.0).astype(F32)
    return R, xc, yc, zc


def _round_away(x):
    """CUDA `round` on float: half away from zero."""
    return (np.sign(x) * np.floor(np.abs(x) + F32(0.5))).astype(F32)


def _psf_at(psf, R, dx, dy, dz):
    """interp_psf branch (:79-101): PSF re-sampled trilinearly at the snapped voxel's offset in slice axes.
    Returns (value, valid)."""
    dp, hp, wp = psf.shape
    r = lambda i, j: R[:, i, j][:, None, None]  # noqa: E731
    xp = ((r(0, 0) * dx + r(1, 0) * dy + r(2, 0) * dz).astype(F32).astype(np.float64) + (wp - 1) / 2.0).astype(F32)
    yp = ((r(0, 1) * dx + r(1, 1) * dy + r(2, 1) * dz).astype(F32).astype(np.float64) + (hp - 1) / 2.0).astype(F32)
    zp = ((r(0, 2) * dx + r(1, 2) * dy + r(2, 2) * dz).astype(F32).astype(np.float64) + (dp - 1) / 2.0).astype(F32)
    ok = ~((xp < 0) | (yp < 0) | (zp < 0) | (xp >= wp - 1) | (yp >= hp - 1) | (zp >= dp - 1))
    xf, yf, zf = np.floor(xp), np.floor(yp), np.floor(zp)
    wx, wy, wz = (xp - xf).astype(F32), (yp - yf).astype(F32), (zp - zf).astype(F32)
    xi = np.clip(xf.astype(np.int64), 0, max(wp - 2, 0))
    yi = np.clip(yf.astype(np.int64), 0, max(hp - 2, 0))
    zi = np.clip(zf.astype(np.int64), 0, max(dp - 2, 0))
    pf = np.pad(psf, ((0, 1), (0, 1), (0, 1)))
    one = F32(1)
    g = lambda a, b, c: pf[zi + c, yi + b, xi + a]  # noqa: E731
    v = np.zeros(xp.shape, F32)
    v = v + (one - wx) * (one - wy) * (one - wz) * g(0, 0, 0)
    v = v + wx * (one - wy) * (one - wz) * g(1, 0, 0)
    v = v + (one - wx) * wy * (one - wz) * g(0, 1, 0)
    v = v + (one - wx) * (one - wy) * wz * g(0, 0, 1)
    v = v + wx * wy * (one - wz) * g(1, 1, 0)
    v = v + wx * (one - wy) * wz * g(1, 0, 1)
    v = v + (one - wx) * wy * wz * g(0, 1, 1)
    v = v + wx * wy * wz * g(1, 1, 1)
    return v.astype(F32), ok


def _cuda_taps(psf):
    dp, hp, wp = psf.shape
    for iz in range(-(dp // 2), (dp + 1) // 2):
        for iy in range(-(hp // 2), (hp + 1) // 2):
            for ix in range(-(wp // 2), (wp + 1) // 2):
                pv = psf[iz + dp // 2, iy + hp // 2, ix + wp // 2]
                if pv != 0:
                    yield ix, iy, iz, F32(pv)


def _tap_pos(R, xc, yc, zc, ix, iy, iz, vol_shape):
    D, H, W = vol_shape
    r = lambda i, j: R[:, i, j][:, None, None]  # noqa: E731
    x = (((xc + r(0, 0) * F32(ix)).astype(F32) + r(0, 1) * F32(iy)).astype(F32) + r(0, 2) * F32(iz)).astype(F32)
    y = (((yc + r(1, 0) * F32(ix)).astype(F32) + r(1, 1) * F32(iy)).astype(F32) + r(1, 2) * F32(iz)).astype(F32)
    z = (((zc + r(2, 0) * F32(ix)).astype(F32) + r(2, 1) * F32(iy)).astype(F32) + r(2, 2) * F32(iz)).astype(F32)
    ok = ~((x < 0) | (y < 0) | (z < 0) | (x >= W - 1) | (y >= H - 1) | (z >= D - 1))
    return x, y, z, ok


_CORNERS = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1))  # (dx,dy,dz), kernel order


def _corner_w(wx, wy, wz, c):
    one = F32(1)
    a = wx if c[0] else (one - wx)
    b = wy if c[1] else (one - wy)
    d = wz if c[2] else (one - wz)
    return (a * b).astype(F32) * d


def slice_acq_forward_cuda(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf):
    """`slice_acquisition_forward_cuda_kernel` (:17-171) + host wrapper (:954-991)."""
    vol = np.asarray(vol, dtype=F32)
    psf = np.asarray(psf, dtype=F32)
    D, H, W = vol.shape
    R, xc, yc, zc = _cuda_geometry(transforms, vol.shape, slice_shape, res_slice)
    val = np.zeros(xc.shape, F32)
    wgt = np.zeros(xc.shape, F32)
    vm = None if vol_mask is None else np.asarray(vol_mask, bool)
    for ix, iy, iz, pv in _cuda_taps(psf):
        x, y, z, ok = _tap_pos(R, xc, yc, zc, ix, iy, iz, vol.shape)
        if interp_psf:
            xr, yr, zr = _round_away(x), _round_away(y), _round_away(z)
            xi = np.clip(xr.astype(np.int64), 0, W - 1)
            yi = np.clip(yr.astype(np.int64), 0, H - 1)
            zi = np.clip(zr.astype(np.int64), 0, D - 1)
            if vm is not None:
                ok = ok & vm[zi, yi, xi]
            pval, okp = _psf_at(psf, R, (xr - xc).astype(F32), (yr - yc).astype(F32), (zr - zc).astype(F32))
            ok = ok & okp
            val = np.where(ok, (val + pval * vol[zi, yi, xi]).astype(F32), val)
            wgt = np.where(ok, (wgt + pval).astype(F32), wgt)
        else:
            xf, yf, zf = np.floor(x), np.floor(y), np.floor(z)
            wx, wy, wz = (x - xf).astype(F32), (y - yf).astype(F32), (z - zf).astype(F32)
            xi = np.clip(xf.astype(np.int64), 0, W - 2)
            yi = np.clip(yf.astype(np.int64), 0, H - 2)
            zi = np.clip(zf.astype(np.int64), 0, D - 2)
            for c in _CORNERS:
                okc = ok if vm is None else (ok & vm[zi + c[2], yi + c[1], xi + c[0]])
                pw = (_corner_w(wx, wy, wz, c) * pv).astype(F32)
                val = np.where(okc, (val + (pw * vol[zi + c[2], yi + c[1], xi + c[0]]).astype(F32)).astype(F32), val)
                wgt = np.where(okc, (wgt + pw).astype(F32), wgt)
    sm = np.ones(xc.shape, bool) if slices_mask is None else np.asarray(slices_mask, bool)
    good = (wgt > 0) & sm
    out = np.where(good, val / np.where(good, wgt, 1), 0).astype(F32)
    return (out, np.where(good, wgt, 0).astype(F32)) if need_weight else out


def slice_acq_adjoint_cuda(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf, equalize):
    """`slice_acquisition_adjoint_forward_cuda_kernel` (:472-670) + `equalize_cuda_kernel` (:672-693, is_grad=false).
    Scatter sums are accumulated in float64 here (the kernel uses fp32 atomics in nondeterministic order)."""
    slices = np.asarray(slices, dtype=F32)
    psf = np.asarray(psf, dtype=F32)
    D, H, W = vol_shape
    R, xc, yc, zc = _cuda_geometry(transforms, vol_shape, slices.shape[-2:], res_slice)
    sm = np.ones(xc.shape, bool) if slices_mask is None else np.asarray(slices_mask, bool)
    vm = None if vol_mask is None else np.asarray(vol_mask, bool)
    # pass 1 (:515-558): pixel weight (vol_mask NOT consulted here)
    wgt = np.zeros(xc.shape, F32)
    cache = []
    for ix, iy, iz, pv in _cuda_taps(psf):
        x, y, z, ok = _tap_pos(R, xc, yc, zc, ix, iy, iz, vol_shape)
        if interp_psf:
            xr, yr, zr = _round_away(x), _round_away(y), _round_away(z)
            pval, okp = _psf_at(psf, R, (xr - xc).astype(F32), (yr - yc).astype(F32), (zr - zc).astype(F32))
            ok = ok & okp
            cache.append((xr, yr, zr, pval, ok))
        else:
            pval = np.full(xc.shape, pv, F32)
            cache.append((x, y, z, pval, ok))
        wgt = np.where(ok, (wgt + pval).astype(F32), wgt)
    live = sm & ~(wgt < 0.5)
    vol = np.zeros(D * H * W, np.float64)
    vw = np.zeros(D * H * W, np.float64)
    wsafe = np.where(live, wgt, 1).astype(F32)
    for a, b, c_, pval, ok in cache:
        ok = ok & live
        pn = (pval / wsafe).astype(F32)
        if interp_psf:
            xi = np.clip(a.astype(np.int64), 0, W - 1)
            yi = np.clip(b.astype(np.int64), 0, H - 1)
            zi = np.clip(c_.astype(np.int64), 0, D - 1)
            iv = zi * (H * W) + yi * W + xi
            if vm is not None:
                ok = ok & vm[zi, yi, xi]
            np.add.at(vol, iv[ok], (pn * slices).astype(F32)[ok])
            np.add.at(vw, iv[ok], pn[ok])
        else:
            xf, yf, zf = np.floor(a), np.floor(b), np.floor(c_)
            wx, wy, wz = (a - xf).astype(F32), (b - yf).astype(F32), (c_ - zf).astype(F32)
            xi = np.clip(xf.astype(np.int64), 0, W - 2)
            yi = np.clip(yf.astype(np.int64), 0, H - 2)
            zi = np.clip(zf.astype(np.int64), 0, D - 2)
            for c in _CORNERS:
                okc = ok if vm is None else (ok & vm[zi + c[2], yi + c[1], xi + c[0]])
                pw = (_corner_w(wx, wy, wz, c) * pn).astype(F32)
                iv = (zi + c[2]) * (H * W) + (yi + c[1]) * W + (xi + c[0])
                np.add.at(vol, iv[okc], (pw * slices).astype(F32)[okc])
                np.add.at(vw, iv[okc], pw[okc])
    vol, vw = vol.astype(F32), vw.astype(F32)
    if equalize:
        m = vw > 0
        vol[m] = vol[m] / vw[m]
    return vol.reshape(vol_shape), vw.reshape(vol_shape)


# --------------------------------------------------------------------------------------
# volumetric helpers of the artifact stages (generator/artifacts/utils.py) -- PINNED by tests/golden/sr_units.npz
# --------------------------------------------------------------------------------------
def mog3d(shape, centers, sigmas):
    """clamp(sum_g exp(-d_g^2/2), 0, 1); a centre is unpacked (x0,y0,z0) with x the LAST axis (utils.py:125-160)."""
    D, H, W = shape
    z, y, x = torch.meshgrid(torch.arange(D).float(), torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    c = np.asarray(centers, dtype=np.float32).reshape(-1, 3)
    s = np.broadcast_to(np.asarray(sigmas, dtype=np.float32).reshape(len(c), -1), (len(c), 3))
    out = torch.zeros((D, H, W))
    for (x0, y0, z0), (sx, sy, sz) in zip(c, s):
        d = ((x - float(x0)) / float(sx)) ** 2 + ((y - float(y0)) / float(sy)) ** 2 + ((z - float(z0)) / float(sz)) ** 2
        out += torch.exp(-d / 2)
    return torch.clamp(out, 0, 1)


def perlin_lattice(res):
    """rand(theta), rand(phi) from the torch global generator -> unit gradients, tileable wrap (utils.py:266-283)."""
    r0, r1, r2 = res
    th = 2 * torch.pi * torch.rand(r0 + 1, r1 + 1, r2 + 1)
    ph = 2 * torch.pi * torch.rand(r0 + 1, r1 + 1, r2 + 1)
    g = torch.stack((torch.sin(ph) * torch.cos(th), torch.sin(ph) * torch.sin(th), torch.cos(ph)), -1)
    g[-1] = g[0]
    g[:, -1] = g[:, 0]
    g[:, :, -1] = g[:, :, 0]
    return g


def perlin_octave(shape, res, g):
    """One octave on `shape` with lattice `g` (utils.py:255-327)."""
    lin = [torch.linspace(0, res[i], shape[i]) for i in range(3)]
    grid = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)
    cell = grid.floor().long()
    loc = grid - cell
    rr = torch.tensor(res)

    def corner(dx, dy, dz):
        i0 = (cell[..., 0] + dx).clamp(max=res[0])
        i1 = (cell[..., 1] + dy).clamp(max=res[1])
        i2 = (cell[..., 2] + dz).clamp(max=res[2])
        d = loc - torch.tensor([float(dx), float(dy), float(dz)])
        return (g[i0, i1, i2] * d).sum(-1)

    t = loc * loc * loc * (loc * (loc * 6 - 15) + 10)
    n00 = corner(0, 0, 0) * (1 - t[..., 0]) + t[..., 0] * corner(1, 0, 0)
    n10 = corner(0, 1, 0) * (1 - t[..., 0]) + t[..., 0] * corner(1, 1, 0)
    n01 = corner(0, 0, 1) * (1 - t[..., 0]) + t[..., 0] * corner(1, 0, 1)
    n11 = corner(0, 1, 1) * (1 - t[..., 0]) + t[..., 0] * corner(1, 1, 1)
    n0 = n00 * (1 - t[..., 1]) + t[..., 1] * n10
    n1 = n01 * (1 - t[..., 1]) + t[..., 1] * n11
    return n0 * (1 - t[..., 2]) + t[..., 2] * n1


def fractal_noise(shape, res, octaves, persistence, lacunarity, increase, lattices=None):
    """generate_fractal_noise_3d (utils.py:330-388) minus its wall-clock re-seed of numpy; lattices drawn from the
    torch global generator in octave order unless given.  Returns (normalised [0,1] field, raw field)."""
    noise = torch.zeros(shape)
    f, a = 1, 1.0
    for q in range(octaves):
        r = (f * res, f * res, f * res)
        g = lattices[q] if lattices is not None else perlin_lattice(r)
        noise += a * perlin_octave(shape, r, g)
        f *= lacunarity
        a *= persistence
    out = (noise + increase - noise.min()) / (noise.max() - noise.min())
    return torch.clamp(out, 0, 1), noise


def rician(slices, thr, sigma, z1, z2):
    """Scanner.add_noise (simulate_reco.py:247-255) with dense noise fields z1, z2 (used where slices > thr)."""
    s = torch.as_tensor(slices).clone()
    m = s > thr
    s[m] = torch.sqrt((s[m] + torch.as_tensor(z1)[m] * sigma) ** 2 + (torch.as_tensor(z2)[m] * sigma) ** 2)
    return s


def box_mean3(v):
    """PSFReconstructor.smooth_volume (simulate_reco.py:584-595)."""
    import torch.nn.functional as TF

    return TF.conv3d(torch.as_tensor(v)[None, None], torch.ones(1, 1, 3, 3, 3) / 27, padding=1)[0, 0]
