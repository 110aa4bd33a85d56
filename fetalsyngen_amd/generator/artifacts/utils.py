"""Mirror of `fetalsyngen.generator.artifacts.utils` (reference generator/artifacts/utils.py:10-388):
the parameter dataclasses of the SR-artifact stages and the volume helpers they share, on MI355X kernels.

    ScannerParams / StructNoiseMergeParams / ReconMergeParams / ReconParams   (ref :10-78)   same fields
    make_gaussian_kernel, gaussian_blur_3d                                    (ref :81-122)  fsg_blur_axis_*
    mog_3d_tensor                                                             (ref :125-160) fsg_mog3d_f32
    apply_kernel, erode, dilate                                               (ref :163-210) fsg_box3d_*
    generate_perlin_noise_3d, generate_fractal_noise_3d                       (ref :224-388) fsg_perlin_fractal_f32

The Perlin lattices (a few thousand gradients) and every per-axis coordinate table are built on the host in
the reference's arithmetic; volumes are only ever touched by HIP kernels.  No CPU execution path.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass

import numpy as np
import torch

from ... import kernels as K
from ... import tables as T


@dataclass
class ScannerParams:
    resolution_slice_fac_min: float
    resolution_slice_fac_max: float
    resolution_slice_max: int
    slice_thickness_min: float
    slice_thickness_max: float
    gap_min: float
    gap_max: float
    min_num_stack: int
    max_num_stack: int
    max_num_slices: int
    noise_sigma_min: float
    noise_sigma_max: float
    TR_min: float
    TR_max: float
    prob_void: float
    prob_gamma: float
    gamma_std: float
    slice_size: int
    restrict_transform: bool
    txy: float
    resolution_recon: float = None
    slice_noise_threshold: float = 0.1


@dataclass
class StructNoiseMergeParams:
    merge_type: str
    gauss_nloc_min: int = None
    gauss_nloc_max: int = None
    gauss_sigma_mu: float = None
    gauss_sigma_std: float = None
    perlin_res_list: list[int] = None
    perlin_octaves_list: list[int] = None
    perlin_persistence: float = None
    perlin_lacunarity: int = None
    perlin_increase_size: float = None


@dataclass
class ReconMergeParams:
    merge_type: str
    gauss_ngaussians_min: int = None
    gauss_ngaussians_max: int = None
    perlin_res_list: list[int] = None
    perlin_octaves_list: list[int] = None
    perlin_persistence: float = None
    perlin_lacunarity: int = None
    perlin_increase_size: float = None


@dataclass
class ReconParams:
    prob_misreg_slice: float
    slices_misreg_ratio: float
    prob_misreg_stack: float
    txy: float
    prob_smooth: float
    prob_rm_slices: float
    rm_slices_min: float
    rm_slices_max: float
    prob_merge: float
    merge_params: ReconMergeParams


# ---- blur ---------------------------------------------------------------------------------------------
def make_gaussian_kernel(sigma, device):
    return torch.from_numpy(T.gaussian_taps(float(sigma))).to(device)


def gaussian_blur_3d(input, stds, device=None):
    """Zero-padded separable Gaussian blur, axes 0,1,2 in turn (ref :93-122)."""
    out = input
    for axis in range(3):
        if stds[axis] > 0:
            out = K.blur_axis(out.contiguous(), axis, T.gaussian_taps(float(stds[axis])))
    return torch.squeeze(out)


# ---- mixture of Gaussians -------------------------------------------------------------------------------
def _scalar(v) -> float:
    if isinstance(v, torch.Tensor):
        return float(v.reshape(-1)[0].item())
    return float(v)


def mog_params(centers, sigmas):
    """(k,3) float32 arrays in the (x0,y0,z0) / (sigma_x,sigma_y,sigma_z) order mog_3d_tensor unpacks (ref :146-154):
    a scalar sigma serves all centres, a per-centre scalar all three axes."""
    k = len(centers)
    if not isinstance(sigmas, (list, np.ndarray)):
        sigmas = [sigmas] * k
    c = np.empty((k, 3), np.float32)
    s = np.empty((k, 3), np.float32)
    for g, (cen, sig) in enumerate(zip(centers, sigmas)):
        c[g] = [_scalar(v) for v in cen]
        s[g] = [_scalar(v) for v in sig] if isinstance(sig, (list, np.ndarray)) else [_scalar(sig)] * 3
    return c, s


def mog_3d_tensor(shape, centers, sigmas, device):
    """Sum of Gaussian blobs clamped to [0,1] on a (D,H,W) grid (ref :125-160).  One fused kernel: per-axis
    squared-distance tables, then a single pass over the volume (the reference makes one full-volume pass and
    three full-size coordinate grids per blob)."""
    c, s = mog_params(centers, sigmas)
    if len(c) == 0:
        return torch.zeros(tuple(int(v) for v in shape), dtype=torch.float32, device=device)
    return K.mog3d(shape, c, s, device)


# ---- binary morphology ------------------------------------------------------------------------------------
def apply_kernel(im, kernel_size=3):
    """Zero-padded box sum, (1,1,D,H,W) like the reference's conv3d with a ones kernel (ref :163-171)."""
    v = im.reshape(im.shape[-3:]).float().contiguous()
    return K.box_sum3d(v, int(kernel_size)).view(1, 1, *v.shape)


def erode(mask, kernel_size=3):
    return (apply_kernel(mask, kernel_size) == kernel_size**3).int().squeeze(0).squeeze(0)


def dilate(mask, kernel_size=3):
    return (apply_kernel(mask, kernel_size) > 0).int().squeeze(0).squeeze(0)


# ---- Perlin noise -------------------------------------------------------------------------------------------
def perlin_interpolant(t):
    return t * t * t * (t * (t * 6 - 15) + 10)


def perlin_lattice(res, tileable=(True, True, True)):
    """Unit gradient per lattice node, (r0+1,r1+1,r2+1,3); torch global generator: rand(theta), rand(phi)
    (ref :266-283)."""
    r0, r1, r2 = (int(v) for v in res)
    theta = 2 * torch.pi * torch.rand(r0 + 1, r1 + 1, r2 + 1)
    phi = 2 * torch.pi * torch.rand(r0 + 1, r1 + 1, r2 + 1)
    g = torch.stack((torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)), dim=-1)
    if tileable[0]:
        g[-1, :, :] = g[0, :, :]
    if tileable[1]:
        g[:, -1, :] = g[:, 0, :]
    if tileable[2]:
        g[:, :, -1] = g[:, :, 0]
    return g


def _octave(shape, res, tileable, amplitude):
    lins = [torch.linspace(0, int(res[i]), int(shape[i])) for i in range(3)]
    return perlin_lattice(res, tileable), lins, tuple(int(v) for v in res), float(amplitude)


def generate_perlin_noise_3d(shape, res, tileable=(True, True, True), interpolant=perlin_interpolant, device=None):
    """One octave of Perlin noise (ref :224-327)."""
    if interpolant is not perlin_interpolant:
        raise NotImplementedError("only the quintic Perlin interpolant is built into the kernel")
    plan = K.PerlinPlan(shape, [_octave(shape, res, tileable, 1.0)], device)
    return K.perlin_fractal(plan)[0]


RESEED_NUMPY_FROM_CLOCK = True  # the reference re-seeds numpy's global generator from the wall clock (ref :365-367)


def fractal_noise_plan(shape, res, octaves=1, persistence=0.5, lacunarity=2, tileable=(True, True, True), device=None):
    """Host side of generate_fractal_noise_3d: the clock re-seed and the per-octave lattices (ref :365-384)."""
    if RESEED_NUMPY_FROM_CLOCK:
        seed = int(time.time())
        os.environ["PYTHONHASHSEED"] = str(seed)
        np.random.seed(seed)
    octs, frequency, amplitude = [], 1, 1
    for _ in range(int(octaves)):
        octs.append(_octave(shape, (frequency * res[0], frequency * res[1], frequency * res[2]), tileable, amplitude))
        frequency *= lacunarity
        amplitude *= persistence
    return K.PerlinPlan(shape, octs, device)


def generate_fractal_noise_3d(shape, res, octaves=1, persistence=0.5, lacunarity=2, tileable=(True, True, True),
                              interpolant=perlin_interpolant, increase=0.0, device=None):
    """Fractal noise normalised to [0,1] (ref :330-388): raw octave sum + min/max in one kernel, normalisation in
    a second (callers that only blend with it use `fractal_noise_plan` + `K.blend` and never materialise it)."""
    if interpolant is not perlin_interpolant:
        raise NotImplementedError("only the quintic Perlin interpolant is built into the kernel")
    if device is None:
        device = "cuda"
    plan = fractal_noise_plan(shape, res, octaves, persistence, lacunarity, tileable, device)
    raw, mm = K.perlin_fractal(plan)
    return K.blend(None, None, raw, w_mm=mm, increase=increase, want_weight=True, want_out=False)[1]
