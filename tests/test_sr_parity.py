"""GPU parity: slice acquisition / adjoint (fsg_slice_acq.hip, through the C ABI) against the oracle and the
golden vectors captured from the reference's torch fallback.

Tolerances (volume intensities 0..100 here):
  * `semantics="torch"`: same taps and voxels as the reference's fallback; fp32 sums in tap order (forward)
    or atomic order (adjoint) instead of the sparse mv's: atol 2e-4 forward, 5e-4 adjoint;
  * `semantics="cuda"`: the oracle restates the CUDA kernel's loops in the same operation order: atol 1e-4 forward
    under FSG_TUNE_PRECISE_MATH (same order on the GPU), + rtol 1e-5 for the default pipelined forward (lerp blend,
    pre-summed tap offsets); 5e-4 adjoint (fp32 scatter order).
    (CUDA semantics are "parity unpinned": see oracle/fsg_oracle_sr.py.)
"""
import numpy as np
import pytest
import torch

from oracle import fsg_oracle_sr as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VS, SS, RES = (20, 24, 28), (14, 18), 1.3


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (and libfsg_hip.so); there is no fallback to skip to")
    from fetalsyngen_amd import kernels

    return kernels


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(x):
    return x.detach().cpu().numpy()


@pytest.mark.parametrize("pk", ["aniso", "iso"])
@pytest.mark.parametrize("mk", ["nomask", "masks"])
def test_torch_semantics_vs_reference_golden(K, golden, pk, mk):
    g = golden("slice_acq")
    vm = g["vol_mask"] if mk == "masks" else None
    sm = g["slices_mask"] if mk == "masks" else None
    tr, psf = dev(g["transforms"]), dev(g[f"psf_{pk}"])
    s, w = K.slice_acq_forward(tr, dev(g["vol"]), dev(vm), dev(sm), psf, SS, RES, need_weight=True, semantics="torch")
    np.testing.assert_allclose(host(s), g[f"fwd_{pk}_{mk}"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(host(w), g[f"fwdw_{pk}_{mk}"], rtol=0, atol=1e-6)
    for eq in (0, 1):
        v = K.slice_acq_adjoint(tr, psf, dev(g[f"fwd_{pk}_{mk}"]), dev(sm), dev(vm), VS, RES, equalize=bool(eq),
                                semantics="torch")
        np.testing.assert_allclose(host(v), g[f"adj_{pk}_{mk}_eq{eq}"], rtol=0, atol=5e-4)


def test_delta_psf_linear_is_reference_grid_sample(K, golden):
    g = golden("slice_acq")
    s, w = K.slice_acq_forward(dev(g["transforms"]), dev(g["vol"]), None, None, dev(g["psf_delta"]), SS, RES,
                               need_weight=True, interp_psf=False)
    s, w = host(s), host(w)
    inside = w > 0
    assert inside.mean() > 0.3
    np.testing.assert_allclose(s[inside], g["fwd_delta_nomask"][inside], rtol=0, atol=2e-4)
    assert np.all(s[~inside] == 0)


@pytest.mark.parametrize("interp_psf", [False, True])
@pytest.mark.parametrize("mk", ["nomask", "masks"])
@pytest.mark.parametrize("pk", ["aniso", "iso"])
def test_cuda_semantics_vs_oracle(K, golden, interp_psf, mk, pk):
    g = golden("slice_acq")
    vm = g["vol_mask"] if mk == "masks" else None
    sm = g["slices_mask"] if mk == "masks" else None
    tr, psf = g["transforms"], g[f"psf_{pk}"]
    es, ew = S.slice_acq_forward_cuda(tr, g["vol"], vm, sm, psf, SS, RES, True, interp_psf)
    s, w = K.slice_acq_forward(dev(tr), dev(g["vol"]), dev(vm), dev(sm), dev(psf), SS, RES, need_weight=True,
                               interp_psf=interp_psf)
    # default build: the unmasked linear forward blends with lerps and sums the PSF value as the tap's weight (fp32
    # rounding apart from the source's order); FSG_TUNE_PRECISE_MATH runs the source's operation order
    np.testing.assert_allclose(host(w), ew, rtol=0, atol=5e-6)
    np.testing.assert_allclose(host(s), es, rtol=1e-5, atol=1e-4)
    from fetalsyngen_amd import _lib
    prev = _lib.load().fsg_set_tuning(2)
    try:
        sp, wp = K.slice_acq_forward(dev(tr), dev(g["vol"]), dev(vm), dev(sm), dev(psf), SS, RES, need_weight=True,
                                     interp_psf=interp_psf)
    finally:
        _lib.load().fsg_set_tuning(prev)
    np.testing.assert_allclose(host(wp), ew, rtol=0, atol=1e-6)
    np.testing.assert_allclose(host(sp), es, rtol=0, atol=1e-4)
    s1 = K.slice_acq_forward(dev(tr), dev(g["vol"]), dev(vm), dev(sm), dev(psf), SS, RES, interp_psf=interp_psf)
    assert torch.equal(s1, s)
    for eq in (False, True):
        ev, evw = S.slice_acq_adjoint_cuda(tr, psf, es, sm, vm, VS, RES, interp_psf, eq)
        v, vw = K.slice_acq_adjoint(dev(tr), dev(psf), dev(es), dev(sm), dev(vm), VS, RES, interp_psf=interp_psf,
                                    equalize=eq, return_weight=True)
        np.testing.assert_allclose(host(vw), evw, rtol=0, atol=2e-5)
        np.testing.assert_allclose(host(v), ev, rtol=0, atol=5e-4)


def _random_rigid(rng, n, max_rot, max_t):
    ax = np.concatenate([rng.uniform(-max_rot, max_rot, (n, 3)), rng.uniform(-max_t, max_t, (n, 3))], 1).astype(np.float32)
    return S.axisangle2mat(torch.from_numpy(ax)).numpy()


@pytest.mark.parametrize("interp_psf", [False, True])
def test_fullsize_properties(K, interp_psf):
    """BASELINE config 4 scale (384^3 volume, anisotropic PSF of a 3 mm slice at 0.5 mm): size-independent
    properties -- partition of unity, <A x, y> = <x, A^T y>, equalised adjoint of constant slices == 1."""
    rng = np.random.default_rng(7)
    n, ss, vs = 24, 320, (384, 384, 384)
    psf = S.get_psf(res_ratio=(1.6, 1.6, 6.0)).numpy()
    assert psf.size <= 4096
    tr = _random_rigid(rng, n, 0.6, 40.0)
    tr[:, 2, 3] += np.linspace(-60, 60, n, dtype=np.float32)
    x = torch.rand(vs, device=DEV)
    ax, w = K.slice_acq_forward(dev(tr), x, None, None, dev(psf), (ss, ss), 1.6, need_weight=True, interp_psf=interp_psf)
    one = K.slice_acq_forward(dev(tr), torch.ones(vs, device=DEV), None, None, dev(psf), (ss, ss), 1.6, interp_psf=interp_psf)
    seen = w > 0
    assert 0.2 < seen.float().mean().item() < 1.0
    assert torch.all((one[seen] - 1).abs() < 1e-5) and torch.all(one[~seen] == 0)
    y = torch.rand_like(ax) * (w >= 0.5)
    aty = K.slice_acq_adjoint(dev(tr), dev(psf), y, None, None, vs, 1.6, interp_psf=interp_psf)
    lhs, rhs = (ax.double() * y.double()).sum().item(), (aty.double() * x.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * abs(lhs)
    v, vw = K.slice_acq_adjoint(dev(tr), dev(psf), (w >= 0.5).float(), None, None, vs, 1.6, interp_psf=interp_psf,
                                equalize=True, return_weight=True)
    hit = vw > 1e-3
    assert hit.float().mean().item() > 0.05
    assert torch.all((v[hit] - 1).abs() < 1e-3) and torch.all(v[vw == 0] == 0)


def test_bad_arguments(K):
    tr = torch.zeros(2, 3, 4, device=DEV)
    vol = torch.zeros(8, 8, 8, device=DEV)
    with pytest.raises(Exception):
        K.slice_acq_forward(tr, vol, None, None, torch.ones(65, 1, 1, device=DEV), (4, 4), 1.0)  # PSF axis too long
    with pytest.raises(ValueError):
        K.slice_acq_forward(tr[:, :2], vol, None, None, torch.ones(1, 1, 1, device=DEV), (4, 4), 1.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.slice_acq_forward(tr.cpu(), vol.cpu(), None, None, torch.ones(1, 1, 1), (4, 4), 1.0)


def test_adjoint_slice_subset_without_copy(K, golden):
    """slice_ids picks the slices a transform applies to (PSFReconstructor.kept_slices_idx, simulate_reco.py:711-769):
    same result as gathering the kept slices first."""
    g = golden("slice_acq")
    tr, psf, s = dev(g["transforms"]), dev(g["psf_aniso"]), dev(g["fwd_aniso_nomask"])
    keep = torch.tensor([3, 0, 4], dtype=torch.int64)
    a = K.slice_acq_adjoint(tr[keep.to(DEV)].contiguous(), psf, s[keep.to(DEV)].contiguous(), None, None, VS, RES,
                            interp_psf=True, equalize=True)
    b = K.slice_acq_adjoint(tr[keep.to(DEV)].contiguous(), psf, s, None, None, VS, RES, interp_psf=True, equalize=True,
                            slice_ids=keep)
    np.testing.assert_allclose(host(a), host(b), rtol=0, atol=1e-4)
    with pytest.raises(IndexError):
        K.slice_acq_adjoint(tr[:3].contiguous(), psf, s, None, None, VS, RES, slice_ids=torch.tensor([0, 5, 1]))


@pytest.mark.parametrize("tuning", [(4096, 0, 20), (4096, 3, 20), (8192, 6, 20), (512, 1, 0), (2048, 2, 0), (16384, 64, 1000)])
def test_adjoint_lds_presum_matches_direct_atomics(K, tuning):
    """The interp_psf adjoint sums a pixel tile's contributions in LDS before touching HBM (chunked PSF planes, bounding
    box per chunk, fallback to direct atomics when the box does not fit): same result as the direct-atomics kernel for
    any tile shape / capacity, at sizes where both the LDS path and the fallback are exercised."""
    from fetalsyngen_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(11)
    n, ss, vs = 12, 96, (72, 80, 88)
    psf = dev(S.get_psf(res_ratio=(1.4, 1.4, 5.0)).numpy())
    tr = _random_rigid(rng, n, 1.2, 12.0)
    tr[:, 2, 3] += np.linspace(-20, 20, n, dtype=np.float32)
    s = torch.rand((n, ss, ss), device=DEV)
    sm = torch.rand((n, ss, ss), device=DEV) > 0.1
    vm = torch.rand(vs, device=DEV) > 0.1
    for masks in ((None, None), (sm, vm)):
        prev = lib.fsg_set_tuning(128)  # FSG_TUNE_SA_DIRECT
        try:
            ref, refw = K.slice_acq_adjoint(dev(tr), psf, s, masks[0], masks[1], vs, 1.4, interp_psf=True, return_weight=True)
        finally:
            lib.fsg_set_tuning(prev)
        assert lib.fsg_slice_acq_set_tuning(*tuning) == 0
        try:
            got, gotw = K.slice_acq_adjoint(dev(tr), psf, s, masks[0], masks[1], vs, 1.4, interp_psf=True, return_weight=True)
        finally:
            lib.fsg_slice_acq_set_tuning(3072, 0, 20)
        assert float(refw.sum()) > 1000
        np.testing.assert_allclose(host(gotw), host(refw), rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(host(got), host(ref), rtol=1e-5, atol=2e-5)


@pytest.mark.gpu
def test_forward_plate_kernel_equals_direct_gathers_for_any_orientation(K):
    """r03: the forward model (linear PSF, no volume mask) runs per slice either by direct gathers or from a plate of the volume in
    LDS (4 x 4 pixel tiles, the taps split over four lane groups: another order of the sum).  Uniformly random orientations
    (what Scanner.scan draws, svort transform.py:178-188), slices reaching beyond every face of the volume, a slice mask:
    plate-only, direct-only and the default per-slice choice agree to fp32 rounding."""
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd.generator.artifacts.svort import get_PSF, random_init_stack_transforms

    lib = _lib.load()
    np.random.seed(3)
    vs = (72, 64, 80)
    vol = torch.rand(vs, device=DEV) * 100
    for res_ratio, ss, rs in (((1.6, 1.6, 6.0), (64, 80), 1.6), ((1.0, 1.0, 3.0), (96, 96), 1.0)):
        psf = get_PSF(res_ratio=res_ratio).to(DEV)
        for rep in range(4):
            tr = random_init_stack_transforms(12, 5.0, False, 6.0).matrix().to(DEV)
            sm = (torch.rand((12, *ss), device=DEV) > 0.2) if rep == 3 else None
            got = {}
            for name, flag in (("direct", 131072), ("plate", 262144), ("auto", 0)):
                lib.fsg_set_tuning(flag)
                try:
                    got[name] = K.slice_acq_forward(tr, vol, None, sm, psf, ss, rs, need_weight=True)
                finally:
                    lib.fsg_set_tuning(0)
            for name in ("plate", "auto"):
                np.testing.assert_allclose(host(got[name][0]), host(got["direct"][0]), rtol=2e-5, atol=2e-4)
                np.testing.assert_allclose(host(got[name][1]), host(got["direct"][1]), rtol=2e-5, atol=1e-5)
            assert float(got["direct"][1].max()) > 0  # the stack does meet the volume
