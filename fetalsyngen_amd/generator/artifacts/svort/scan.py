"""Scan geometry helpers: mirror of `fetalsyngen.generator.artifacts.svort.data` (reference
data/utils.py:19-109 `interleave_index`, `resolution2sigma`, `get_PSF`; data/fetal_motion.py:13-48
`get_trajectory`, `sample_motion`).

Trajectories: the reference ships its fetal-motion trajectories as `svort/data/traj.npy`, a *pickled* object
array (numpy refuses it with allow_pickle=False), so it is not loaded here.  `get_trajectory()` serves a
deterministic synthetic bank of the same structure instead -- lists of `(traj, T, dT)` with `traj(t)` ->
(len(t), 3) (Euler xyz angles in radians / translations in mm), `T` samples long, `dT` seconds per sample --
and `set_trajectory_bank()` lets a user install their own.
"""
from __future__ import annotations

from math import log, sqrt

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from .rigid import RigidTransform

GAUSSIAN_FWHM = 1 / (2 * sqrt(2 * log(2)))
SINC_FWHM = 1.206709128803223 * GAUSSIAN_FWHM


def interleave_index(N, n_i):
    """idx[j] = acquisition time rank of slice j when the stack is acquired in n_i interleaved passes."""
    idx = [None] * N
    t = 0
    for first in range(n_i):
        for j in range(first, N, n_i):
            idx[j] = t
            t += 1
    return idx


def resolution2sigma(rx, ry=None, rz=None, /, isotropic=False):
    """PSF standard deviations from the acquisition resolution (sinc-like in plane, Gaussian through plane)."""
    fx, fy, fz = (GAUSSIAN_FWHM,) * 3 if isotropic else (SINC_FWHM, SINC_FWHM, GAUSSIAN_FWHM)
    assert not ((ry is None) ^ (rz is None))
    if ry is not None:
        return fx * rx, fy * ry, fz * rz
    if isinstance(rx, (float, int)):
        return fx * rx if isotropic else (fx * rx, fy * rx, fz * rx)
    if isinstance(rx, torch.Tensor):
        if isotropic:
            return fx * rx
        assert rx.shape[-1] == 3
        return rx * torch.tensor([fx, fy, fz], dtype=rx.dtype, device=rx.device)
    if isinstance(rx, (list, tuple)):
        assert len(rx) == 3
        return resolution2sigma(rx[0], rx[1], rx[2], isotropic=isotropic)
    raise Exception(str(type(rx)))


def get_PSF(r_max=None, res_ratio=(1, 1, 3), threshold=1e-4, device=torch.device("cpu"), psf_type="gaussian"):
    """Point-spread function of one slice on the voxel grid, cropped to its support, unit sum
    (data/utils.py:64-109).  A few hundred taps: evaluated on the host, moved to `device` at the end."""
    sx, sy, sz = resolution2sigma(res_ratio, isotropic=False)
    if r_max is None:
        r_max = max(max(int(2 * r + 1) for r in (sx, sy, sz)), 4)
    x = torch.linspace(-r_max, r_max, 2 * r_max + 1, dtype=torch.float32)
    gz, gy, gx = torch.meshgrid(x, x, x, indexing="ij")
    if psf_type == "gaussian":
        psf = torch.exp(-0.5 * (gx**2 / sx**2 + gy**2 / sy**2 + gz**2 / sz**2))
    elif psf_type == "sinc":
        psf = torch.sinc(torch.sqrt((gx / res_ratio[0]) ** 2 + (gy / res_ratio[1]) ** 2)) ** 2 * torch.exp(
            -0.5 * gz**2 / sz**2)
    else:
        raise TypeError(f"Unknown PSF type: <{psf_type}>!")
    psf[psf.abs() < threshold] = 0
    lo = [int(torch.nonzero(psf.sum(dims) > 0)[0, 0]) for dims in ((1, 2), (0, 2), (0, 1))]  # z, y, x
    n = 2 * r_max + 1
    psf = psf[lo[0] : n - lo[0], lo[1] : n - lo[1], lo[2] : n - lo[2]].contiguous()
    return (psf / psf.sum()).to(device)


# ---- motion trajectories ------------------------------------------------------------------------------
class SmoothTrajectory:
    """traj(t): a sum of a few low-frequency sinusoids per channel, t in samples; (len(t), 3)."""

    def __init__(self, rng, T, dT, amplitude, n_modes=6):
        periods_s = np.exp(rng.uniform(np.log(8.0), np.log(240.0), (n_modes, 3)))
        self.omega = 2 * np.pi * dT / periods_s
        self.phase = rng.uniform(0, 2 * np.pi, (n_modes, 3))
        self.amp = amplitude * rng.dirichlet(np.ones(n_modes), 3).T * rng.uniform(0.2, 1.0, 3)

    def __call__(self, t):
        t = np.asarray(t, dtype=np.float64).reshape(-1, 1, 1)
        return (self.amp * np.sin(self.omega * t + self.phase)).sum(1)


_bank = None


def synthetic_trajectory_bank(n=32, T=4000, dT=0.25, seed=20220519):
    """(rotations, translations): n smooth trajectories each, up to ~0.15 rad / ~4 mm excursions."""
    rng = np.random.default_rng(seed)
    rot = [(SmoothTrajectory(rng, T, dT, 0.15), T, dT) for _ in range(n)]
    trans = [(SmoothTrajectory(rng, T, dT, 4.0), T, dT) for _ in range(n)]
    return rot, trans


def set_trajectory_bank(rot, trans):
    global _bank
    _bank = (rot, trans)


def get_trajectory():
    global _bank
    if _bank is None:
        _bank = synthetic_trajectory_bank()
    return _bank


def sample_motion(ts, device=None, rand=True):
    """Per-slice motion relative to the first slice of the stack, sampled from one rotation and one
    translation trajectory (data/fetal_motion.py:24-48; same numpy draw order)."""
    trajs_rot, trajs_trans = get_trajectory()
    traj, T, dT = trajs_rot[np.random.choice(len(trajs_rot))]
    t0 = np.random.uniform(0, T - ts[-1] / dT) if rand else 0
    R = traj(t0 + ts / dT)
    if rand:
        R = R[:, np.random.permutation(3)]
        R = R * (2 * (np.random.rand(1, 3) < 0.5) - 1)
    R = Rotation.from_euler("xyz", R).as_matrix()
    traj, T, dT = trajs_trans[np.random.choice(len(trajs_trans))]
    t0 = np.random.uniform(0, T - ts[-1] / dT) if rand else 0
    trans = traj(t0 + ts / dT)
    if rand:
        trans = trans[:, np.random.permutation(3)]
        trans = trans * (2 * (np.random.rand(1, 3) < 0.5) - 1)
    R = torch.tensor(R, dtype=torch.float32)
    trans = torch.tensor(trans, dtype=torch.float32)
    R = torch.matmul(R, R[0].transpose(-2, -1))
    trans = trans - trans[0]
    return RigidTransform(torch.cat((R, trans.unsqueeze(-1)), -1), trans_first=False)


def random_stack(n_slice, gap, max_angle=0.3):
    """(n,3,4) matrices of one mildly tilted stack (benchmark / test helper, numpy global RNG)."""
    ax = torch.zeros((n_slice, 6), dtype=torch.float32)
    ax[:, :3] = torch.from_numpy(np.random.uniform(-max_angle, max_angle, 3).astype(np.float32))
    ax[:, 5] = (torch.arange(n_slice, dtype=torch.float32) - (n_slice - 1) / 2.0) * gap
    return RigidTransform(ax).matrix()
