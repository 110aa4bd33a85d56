"""Rigid-transform algebra of the slice-stack simulation: mirror of
`fetalsyngen.generator.artifacts.svort.transform` (reference transform/transform.py:14-128, :134-199,
:359-390) and of the CPU conversions in transform/transform_convert.py:24-161.

A stack has at most a few hundred slices, so this is host work: every transform lives in a (n,3,4) or (n,6)
fp32 tensor on the CPU, conversions are evaluated element-wise in the reference's operation order (so they
agree with the reference's CPU path bit for bit), and only the final (n,3,4) matrices are uploaded, once
per launch, by the slice-acquisition wrappers.  The reference launches tiny CUDA kernels for these
conversions (transform_convert_cuda_kernel.cu); at n <= 250 a host loop-free evaluation is cheaper than the
launches.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation

EPS = 1e-6


def axisangle2mat(axisangle: torch.Tensor) -> torch.Tensor:
    """(n,6) [rotation vector | translation] -> (n,3,4).  Rodrigues' formula for theta^2 > 1e-6, the
    first-order matrix I + [a]x below (transform_convert.py:24-85)."""
    a = axisangle.detach().to("cpu", torch.float32)
    ang, t = a[:, :3], a[:, 3:]
    th2 = (ang**2).sum(1)
    big = th2 > EPS
    th = torch.sqrt(torch.where(big, th2, torch.ones_like(th2)))
    u = ang / th[:, None]
    s, c = torch.sin(th), torch.cos(th)
    o = 1 - c
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    rows = [
        (c + x * x * o, x * y * o - z * s, y * s + x * z * o),
        (z * s + x * y * o, c + y * y * o, -x * s + y * z * o),
        (-y * s + x * z * o, x * s + y * z * o, c + z * z * o),
    ]
    one, ax, ay, az = torch.ones_like(th2), ang[:, 0], ang[:, 1], ang[:, 2]
    small = [(one, -az, ay), (az, one, -ax), (-ay, ax, one)]
    out = torch.empty((a.shape[0], 3, 4), dtype=torch.float32)
    for i in range(3):
        for j in range(3):
            out[:, i, j] = torch.where(big, rows[i][j], small[i][j])
    out[:, :, 3] = t
    return out


def mat2axisangle(mat: torch.Tensor) -> torch.Tensor:
    """(n,3,4) -> (n,6) via the four-branch quaternion extraction (transform_convert.py:88-161)."""
    m = mat.detach().to("cpu", torch.float32)
    r = lambda i, j: m[:, i, j]  # noqa: E731
    low22 = r(2, 2) < EPS
    d0_gt_d1 = r(0, 0) > r(1, 1)
    d0_lt_nd1 = r(0, 0) < -r(1, 1)

    def quat(s, w, x, y, z):
        return torch.stack((w(s), x(s), y(s), z(s)), 0)

    s1 = 2.0 * torch.sqrt(r(0, 0) + r(1, 1) + r(2, 2) + 1.0)
    s2 = 2.0 * torch.sqrt(r(0, 0) - r(1, 1) - r(2, 2) + 1.0)
    s3 = 2.0 * torch.sqrt(r(1, 1) - r(0, 0) - r(2, 2) + 1.0)
    s4 = 2.0 * torch.sqrt(r(2, 2) - r(0, 0) - r(1, 1) + 1.0)
    q1 = torch.stack((0.25 * s1, (r(2, 1) - r(1, 2)) / s1, (r(0, 2) - r(2, 0)) / s1, (r(1, 0) - r(0, 1)) / s1), 0)
    q2 = torch.stack(((r(2, 1) - r(1, 2)) / s2, 0.25 * s2, (r(0, 1) + r(1, 0)) / s2, (r(0, 2) + r(2, 0)) / s2), 0)
    q3 = torch.stack(((r(0, 2) - r(2, 0)) / s3, (r(0, 1) + r(1, 0)) / s3, 0.25 * s3, (r(1, 2) + r(2, 1)) / s3), 0)
    q4 = torch.stack(((r(1, 0) - r(0, 1)) / s4, (r(0, 2) + r(2, 0)) / s4, (r(1, 2) + r(2, 1)) / s4, 0.25 * s4), 0)
    q = torch.where((~low22) & (~d0_lt_nd1), q1, torch.zeros_like(q1))
    q = torch.where(low22 & d0_gt_d1, q2, q)
    q = torch.where(low22 & (~d0_gt_d1), q3, q)
    q = torch.where((~low22) & d0_lt_nd1, q4, q)
    q = torch.where(q[0] < 0, -q, q)
    w, x, y, z = q[0], q[1], q[2], q[3]
    na = torch.sqrt(x**2 + y**2 + z**2)
    f = torch.where(na > EPS, 2 * torch.atan2(na, w) / na, 2.0 / w)
    out = torch.empty((m.shape[0], 6), dtype=torch.float32)
    out[:, 0], out[:, 1], out[:, 2] = x * f, y * f, z * f
    out[:, 3:] = m[:, :, 3]
    return out


def _to_trans_first(mat):  # t' = R^T t
    R, t = mat[:, :, :3], mat[:, :, 3:]
    return torch.cat((R, R.transpose(-2, -1) @ t), -1)


def _to_trans_last(mat):  # t' = R t
    R, t = mat[:, :, :3], mat[:, :, 3:]
    return torch.cat((R, R @ t), -1)


class RigidTransform:
    """n rigid transforms, stored as given (axis-angle (n,6) or matrix (n,3,4)), with the reference's
    `trans_first` convention flag (transform.py:14-128).  Data stays on the host."""

    def __init__(self, data, trans_first=True, device=None):
        data = data.detach().to("cpu")
        self.trans_first = trans_first
        self._axisangle = self._matrix = None
        if data.shape[1] == 6:
            self._axisangle = data
        elif data.shape[1] == 3:
            self._matrix = data
        else:
            raise Exception("Unknown format for rigid transform!")

    def _data(self):
        return self._axisangle if self._axisangle is not None else self._matrix

    def matrix(self, trans_first=True):
        mat = self._matrix if self._matrix is not None else axisangle2mat(self._axisangle)
        if self.trans_first and not trans_first:
            mat = _to_trans_last(mat)
        elif not self.trans_first and trans_first:
            mat = _to_trans_first(mat)
        return mat

    def axisangle(self, trans_first=True):
        ax = self._axisangle if self._axisangle is not None else mat2axisangle(self._matrix)
        if self.trans_first and not trans_first:
            ax = mat2axisangle(_to_trans_last(axisangle2mat(ax)))
        elif not self.trans_first and trans_first:
            ax = mat2axisangle(_to_trans_first(axisangle2mat(ax)))
        return ax

    def inv(self):
        m = self.matrix(True)
        R, t = m[:, :, :3], m[:, :, 3:]
        return RigidTransform(torch.cat((R.transpose(-2, -1), -(R @ t)), -1), True)

    def compose(self, other):
        a, b = self.matrix(True), other.matrix(True)
        R1, t1, R2, t2 = a[:, :, :3], a[:, :, 3:], b[:, :, :3], b[:, :, 3:]
        return RigidTransform(torch.cat((R1 @ R2, t2 + R2.transpose(-2, -1) @ t1), -1), True)

    def __getitem__(self, idx):
        d = self._data()[idx]
        if d.dim() < self._data().dim():
            d = d.unsqueeze(0)
        return RigidTransform(d, self.trans_first)

    def detach(self):
        return RigidTransform(self._data().detach(), self.trans_first)

    @property
    def device(self):
        return self._data().device

    def dtype(self):
        return self._data().dtype

    def __len__(self):
        return self._data().shape[0]

    @staticmethod
    def cat(transforms):
        return RigidTransform(torch.cat([t.matrix(True) for t in transforms], 0), True)

    def mean(self, trans_first=True, simple_mean=True):
        if not simple_mean:
            raise NotImplementedError("average_rotation (transform.py:301-336) is not on the generator's path")
        return RigidTransform(self.axisangle(trans_first).mean(0, keepdim=True), trans_first)


def mat_update_resolution(mat, res_from, res_to):
    """Rescale the translation column (transform.py:162-167)."""
    assert mat.dim() == 3
    fac = torch.ones_like(mat[:1, :1])
    fac[..., 3] = res_from / res_to
    return mat * fac


def ax_update_resolution(ax, res_from, res_to):
    assert ax.dim() == 2
    fac = torch.ones_like(ax[:1])
    fac[:, 3:] = res_from / res_to
    return ax * fac


def random_angle(n, restricted, device=None):
    """n uniformly distributed rotations as rotation vectors (transform.py:178-188).  numpy global draws:
    rand(n), rand(n), rand(n)."""
    a = 2 * np.pi * np.random.rand(n)
    b = np.arccos(2 * np.random.rand(n) - 1)
    c = np.pi * np.random.rand(n) if restricted else np.pi * (2 * np.random.rand(n) - 1)
    rv = Rotation.from_euler("ZXZ", np.stack([a, b, c], -1)).as_rotvec()
    return torch.from_numpy(rv).to(torch.float32)


def random_init_stack_transforms(n_slice, gap, restricted, txy, device=None):
    """One random orientation for the whole stack, slices `gap` apart along its normal, optional common
    in-plane shift (transform.py:359-369).  numpy draws: random_angle(1), then uniform x2 if txy."""
    angle = random_angle(1, restricted).expand(n_slice, -1)
    tz = (torch.arange(0, n_slice, dtype=torch.float32) - (n_slice - 1) / 2.0) * gap
    if txy:
        tx = torch.ones_like(tz) * np.random.uniform(-txy, txy)
        ty = torch.ones_like(tz) * np.random.uniform(-txy, txy)
    else:
        tx = ty = torch.zeros_like(tz)
    return RigidTransform(torch.cat((angle, torch.stack((tx, ty, tz), -1)), -1), True)


def init_zero_transform(n, device=None):
    return RigidTransform(torch.zeros((n, 6), dtype=torch.float32))


def reset_transform(transform):
    """Forget orientation and in-plane shift, centre the slice positions (transform.py:386-390)."""
    ax = transform.axisangle().clone()
    ax[:, :-1] = 0
    ax[:, -1] -= ax[:, -1].mean()
    return RigidTransform(ax)


def init_stack_transform(n_slice: int, gap: float, device=None) -> RigidTransform:
    """Axis-aligned stack: identity orientation, slices `gap` apart, centred (transform.py:372-378)."""
    ax = torch.zeros((n_slice, 6), dtype=torch.float32)
    ax[:, -1] = (torch.arange(n_slice, dtype=torch.float32) - (n_slice - 1) / 2.0) * gap
    return RigidTransform(ax, trans_first=True)


def mat_transform_points(mat: torch.Tensor, x: torch.Tensor, trans_first: bool) -> torch.Tensor:
    """Apply (*,3,4) transforms to (*,3) points: R (x + t) when trans_first, else R x + t (transform.py:393-404)."""
    R, T = mat[..., :-1], mat[..., -1:]
    x = x[..., None]
    x = torch.matmul(R, x + T) if trans_first else torch.matmul(R, x) + T
    return x[..., 0]


def transform_points(transform: RigidTransform, x: torch.Tensor) -> torch.Tensor:
    """Points (N,3) through N transforms, or (*,3) through one (transform.py:407-414)."""
    assert x.ndim == 2 and x.shape[-1] == 3
    return mat_transform_points(transform.matrix(transform.trans_first), x, transform.trans_first)
