"""Hand-derivable known-answer cases for the CUDA arithmetic of the slice acquisition (the DEFAULT mode of
`svort.slice_acq`, `semantics="cuda"`), so that the default has an anchor that is not the oracle's own loops.

Source of the closed forms: reference `svort/slice_acquisition/slice_acq_cuda_kernel.cu` -- forward :17-171, adjoint
:472-670, equalize :672-693.  The reference holds no fixture for this arithmetic and CUDA cannot run in the build
container, so each case below is small enough that the kernel's sum can be written out by hand:

  geometry   one slice, identity rotation, res_slice = 1; volume 9^3, slice 5x5  =>  pixel (iy, ix) sits at volume
             position (z, y, x) = (4 + tz, iy + 2 + ty, ix + 2 + tx)   [(:45-55): _x = (ix - (w-1)/2) res + tx, centre +(W-1)/2]
  PSF        3x3x3, taps p[a, b, c] at offsets (a-1, b-1, c-1)          [(:60-62): iz_p from -d_p/2 to (d_p+1)/2 - 1]

  linear, integer position   : wx = wy = wz = 0, so only the base corner of each tap counts (:118-131):
                               slice = sum_abc p[abc] vol[z+a-1, y+b-1, x+c-1] / sum_abc p[abc]
  linear, tx = 0.5           : wx = 0.5: every tap reads 0.5 vol[.., x] + 0.5 vol[.., x+1]
  interp_psf, integer pos.   : nearest voxel = the tap's own voxel; the PSF is re-interpolated at
                               x_psf = (x_round - x_center) + (w_p-1)/2 = c, and taps with x_psf >= w_p - 1 are SKIPPED (:84),
                               so only taps a, b, c in {0, 1} count, with their own values (weights exactly 0/1)
  interp_psf, tx = 0.5       : C round() is half AWAY from zero (:73): x = k + 0.5 -> k + 1, x_psf = c + 0.5, so tap c reads
                               0.5 p[.., c] + 0.5 p[.., c+1] for c in {0, 1} and is skipped for c = 2
  adjoint (linear, integer)  : weight = sum p (>= 0.5 or the pixel is dropped, :565), every tap adds p/weight * s to its voxel
                               and p/weight to the weight volume (:620-626); equalize divides where the weight is > 0 (:672-693)

The same cases are asserted on the CPU oracle (`-m "not gpu"`) and on the HIP kernels (`-m gpu`).
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import fsg_oracle_sr as S

VS, SS = (9, 9, 9), (5, 5)


def _case(seed=0):
    rs = np.random.RandomState(seed)
    vol = (rs.rand(*VS) * 100).astype(np.float32)
    psf = (0.2 + rs.rand(3, 3, 3)).astype(np.float32)
    slices = (rs.rand(1, *SS) * 50).astype(np.float32)
    return vol, psf, slices


def _transform(tx=0.0, ty=0.0, tz=0.0):
    t = np.zeros((1, 3, 4), dtype=np.float32)
    t[0, :, :3] = np.eye(3)
    t[0, :, 3] = (tx, ty, tz)
    return t


def _expected_forward_linear(vol, psf, tx):
    """slice[iy, ix] by the closed form above, float64 (tolerance covers fp32 summation)."""
    out = np.zeros(SS)
    wsum = np.zeros(SS)
    fx = 0.5 if tx == 0.5 else 0.0
    for iy, ix in itertools.product(range(SS[0]), range(SS[1])):
        acc = w = 0.0
        for a, b, c in itertools.product(range(3), repeat=3):
            z, y, x = 4 + a - 1, iy + 2 + b - 1, ix + 2 + c - 1
            p = float(psf[a, b, c])
            acc += p * ((1 - fx) * vol[z, y, x] + fx * vol[z, y, x + 1])
            w += p
        out[iy, ix], wsum[iy, ix] = acc / w, w
    return out, wsum


def _expected_forward_nn(vol, psf, tx):
    out = np.zeros(SS)
    wsum = np.zeros(SS)
    for iy, ix in itertools.product(range(SS[0]), range(SS[1])):
        acc = w = 0.0
        for a, b, c in itertools.product(range(3), repeat=3):
            if a == 2 or b == 2 or c == 2:
                continue  # x_psf >= w_p - 1 (and the like): skipped by the kernel
            z, y = 4 + a - 1, iy + 2 + b - 1
            if tx == 0.5:   # x = ix + 2.5 + c - 1 rounds up, x_psf = c + 0.5
                x = ix + 2 + c
                p = 0.5 * float(psf[a, b, c]) + 0.5 * float(psf[a, b, c + 1])
            else:
                x = ix + 2 + c - 1
                p = float(psf[a, b, c])
            acc += p * vol[z, y, x]
            w += p
        out[iy, ix], wsum[iy, ix] = acc / w, w
    return out, wsum


def _expected_adjoint_linear(psf, slices):
    vol = np.zeros(VS)
    wgt = np.zeros(VS)
    tot = float(psf.astype(np.float64).sum())
    for iy, ix in itertools.product(range(SS[0]), range(SS[1])):
        for a, b, c in itertools.product(range(3), repeat=3):
            z, y, x = 4 + a - 1, iy + 2 + b - 1, ix + 2 + c - 1
            vol[z, y, x] += float(psf[a, b, c]) / tot * float(slices[0, iy, ix])
            wgt[z, y, x] += float(psf[a, b, c]) / tot
    return vol, wgt


# ---- the oracle (CPU) ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tx", [0.0, 0.5])
def test_oracle_forward_known_answers(tx):
    vol, psf, _ = _case()
    for interp_psf, expected in ((False, _expected_forward_linear), (True, _expected_forward_nn)):
        s, w = S.slice_acq_forward_cuda(_transform(tx), vol, None, None, psf, SS, 1.0, True, interp_psf)
        es, ew = expected(vol, psf, tx)
        np.testing.assert_allclose(s[0], es, rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(w[0], ew, rtol=2e-6)


def test_oracle_adjoint_known_answer():
    _, psf, slices = _case()
    ev, ew = _expected_adjoint_linear(psf, slices)
    v, w = S.slice_acq_adjoint_cuda(_transform(), psf, slices, None, None, VS, 1.0, False, equalize=False)
    np.testing.assert_allclose(v, ev, rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(w, ew, rtol=2e-6, atol=1e-7)
    v, _w = S.slice_acq_adjoint_cuda(_transform(), psf, slices, None, None, VS, 1.0, False, equalize=True)
    np.testing.assert_allclose(v, np.where(ew > 0, ev / np.where(ew > 0, ew, 1), 0.0), rtol=5e-6, atol=1e-5)


def test_oracle_pixels_leaving_the_volume_are_dropped():
    # tz = +5.5 puts the lowest tap plane at z = 8.5 >= D - 1: every tap fails the inside test (:66), pixels keep their 0 and weight 0
    vol, psf, _ = _case()
    s, w = S.slice_acq_forward_cuda(_transform(tz=5.5), vol, None, None, psf, SS, 1.0, True, False)
    assert not s.any() and not w.any()
    # tz = +3: taps a = 2 reach z = 8 = D - 1 and are excluded, the rest is the closed form over a in {0, 1}
    s, w = S.slice_acq_forward_cuda(_transform(tz=3.0), vol, None, None, psf, SS, 1.0, True, False)
    acc = wt = 0.0
    for a, b, c in itertools.product(range(2), range(3), range(3)):
        acc += float(psf[a, b, c]) * vol[7 + a - 1, 2 + 2 + b - 1, 1 + 2 + c - 1]
        wt += float(psf[a, b, c])
    np.testing.assert_allclose(s[0, 2, 1], acc / wt, rtol=2e-6)
    np.testing.assert_allclose(w[0, 2, 1], wt, rtol=2e-6)


# ---- the HIP kernels (default mode: CUDA arithmetic) ----------------------------------------------------------------
def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [False, True])
@pytest.mark.parametrize("tx", [0.0, 0.5])
def test_hip_forward_known_answers(tx, precise):
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import kernels as K

    vol, psf, _ = _case()
    prev = _lib.load().fsg_set_tuning(2 if precise else 0)  # FSG_TUNE_PRECISE_MATH: the .cu's operation order
    try:
        for interp_psf, expected in ((False, _expected_forward_linear), (True, _expected_forward_nn)):
            s, w = K.slice_acq_forward(_dev(_transform(tx)), _dev(vol), None, None, _dev(psf), SS, 1.0, need_weight=True,
                                       interp_psf=interp_psf)
            es, ew = expected(vol, psf, tx)
            np.testing.assert_allclose(s.cpu().numpy()[0], es, rtol=5e-6, atol=1e-4)
            np.testing.assert_allclose(w.cpu().numpy()[0], ew, rtol=5e-6)
    finally:
        _lib.load().fsg_set_tuning(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("direct", [False, True])
def test_hip_adjoint_known_answer(direct):
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import kernels as K

    _, psf, slices = _case()
    ev, ew = _expected_adjoint_linear(psf, slices)
    prev = _lib.load().fsg_set_tuning(128 if direct else 0)  # FSG_TUNE_SA_DIRECT: direct atomics instead of the LDS pre-sum
    try:
        v, w = K.slice_acq_adjoint(_dev(_transform()), _dev(psf), _dev(slices), None, None, VS, 1.0, equalize=False,
                                   return_weight=True)
        np.testing.assert_allclose(v.cpu().numpy(), ev, rtol=5e-6, atol=1e-5)
        np.testing.assert_allclose(w.cpu().numpy(), ew, rtol=5e-6, atol=1e-7)
        v = K.slice_acq_adjoint(_dev(_transform()), _dev(psf), _dev(slices), None, None, VS, 1.0, equalize=True)
        np.testing.assert_allclose(v.cpu().numpy(), np.where(ew > 0, ev / np.where(ew > 0, ew, 1), 0.0), rtol=1e-5, atol=1e-5)
    finally:
        _lib.load().fsg_set_tuning(prev)


@pytest.mark.gpu
def test_hip_pixels_leaving_the_volume_are_dropped():
    from fetalsyngen_amd import kernels as K

    vol, psf, _ = _case()
    s, w = K.slice_acq_forward(_dev(_transform(tz=5.5)), _dev(vol), None, None, _dev(psf), SS, 1.0, need_weight=True)
    assert not s.cpu().numpy().any() and not w.cpu().numpy().any()
