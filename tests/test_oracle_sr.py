"""CPU: the slice-acquisition oracle (oracle/fsg_oracle_sr.py) against the golden vectors captured from the
reference's torch fallback (tests/golden/slice_acq.npz), and the anchors of its CUDA-semantics restatement."""
import numpy as np
import pytest
import torch

from oracle import fsg_oracle_sr as S

VS, SS, RES = (20, 24, 28), (14, 18), 1.3


def test_psf_matches_reference(golden):
    g = golden("slice_acq")
    assert np.array_equal(S.get_psf(res_ratio=(1, 1, 3)).numpy(), g["psf_aniso"])
    assert np.array_equal(S.get_psf(res_ratio=(1.2, 1.2, 1.2)).numpy(), g["psf_iso"])
    assert np.array_equal(S.get_psf(0).numpy(), g["psf_delta"]) and g["psf_delta"].shape == (1, 1, 1)


@pytest.mark.parametrize("pk", ["aniso", "iso"])
@pytest.mark.parametrize("mk", ["nomask", "masks"])
def test_torch_semantics_pinned(golden, pk, mk):
    g = golden("slice_acq")
    vm = g["vol_mask"] if mk == "masks" else None
    sm = g["slices_mask"] if mk == "masks" else None
    s, w = S.slice_acq_forward_torch(g["transforms"], g["vol"], vm, sm, g[f"psf_{pk}"], SS, RES, True)
    # same taps, same voxels; only the fp32 summation order differs (sparse mv vs tap order)
    np.testing.assert_allclose(s, g[f"fwd_{pk}_{mk}"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(w, g[f"fwdw_{pk}_{mk}"], rtol=0, atol=1e-6)
    for eq in (0, 1):
        v = S.slice_acq_adjoint_torch(g["transforms"], g[f"psf_{pk}"], g[f"fwd_{pk}_{mk}"], sm, vm, VS, RES, bool(eq))
        np.testing.assert_allclose(v, g[f"adj_{pk}_{mk}_eq{eq}"], rtol=0, atol=1e-4)


def test_cuda_linear_delta_psf_is_grid_sample(golden):
    """Anchor (a): 1x1x1 PSF + linear mode == the reference's grid_sample path wherever the sample's 2x2x2
    neighbourhood is inside the volume (slice_acq.py:445-480 vs slice_acq_cuda_kernel.cu:110-161)."""
    g = golden("slice_acq")
    s, w = S.slice_acq_forward_cuda(g["transforms"], g["vol"], None, None, g["psf_delta"], SS, RES, True, False)
    ref = g["fwd_delta_nomask"]
    inside = w > 0
    assert inside.mean() > 0.3
    np.testing.assert_allclose(s[inside], ref[inside], rtol=0, atol=2e-4)
    assert np.all(s[~inside] == 0)


@pytest.mark.parametrize("interp_psf", [False, True])
def test_cuda_forward_adjoint_pair(golden, interp_psf):
    """Anchor (b): <A x, y> == <x, A^T y> for y supported on pixels the adjoint keeps (weight >= 0.5)."""
    g = golden("slice_acq")
    rng = np.random.default_rng(3)
    tr, psf = g["transforms"], g["psf_aniso"]
    x = rng.random(VS, dtype=np.float32)
    ax, w = S.slice_acq_forward_cuda(tr, x, None, None, psf, SS, RES, True, interp_psf)
    y = rng.random(ax.shape, dtype=np.float32) * (w >= 0.5)
    aty, _ = S.slice_acq_adjoint_cuda(tr, psf, y, None, None, VS, RES, interp_psf, False)
    lhs, rhs = float((ax.astype(np.float64) * y).sum()), float((aty.astype(np.float64) * x).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    # a constant volume is reproduced wherever the PSF sees the volume at all
    one, w1 = S.slice_acq_forward_cuda(tr, np.ones(VS, np.float32), None, None, psf, SS, RES, True, interp_psf)
    np.testing.assert_allclose(one[w1 > 0], 1.0, atol=2e-6)


def test_rigid_algebra_roundtrip():
    rng = np.random.default_rng(5)
    ax = torch.from_numpy(np.concatenate([rng.uniform(-2, 2, (32, 3)), rng.uniform(-30, 30, (32, 3))], 1).astype(np.float32))
    ax[0, :3] = 0
    ax[1, :3] = 1e-4
    m = S.axisangle2mat(ax)
    R = m[:, :, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(32, 3, 3), atol=1e-5)
    assert torch.allclose(S.mat2axisangle(m), ax, atol=2e-4)
    # compose(a, a^-1) is the identity in the translation-first convention
    inv = torch.cat((R.transpose(1, 2), -torch.matmul(R, m[:, :, 3:])), -1)
    idt = S.compose(m, inv)
    assert torch.allclose(idt[:, :, :3], torch.eye(3).expand(32, 3, 3), atol=1e-5)
    assert torch.allclose(idt[:, :, 3], torch.zeros(32, 3), atol=1e-3)


def test_interleave_index():
    assert S.interleave_index(7, 3) == [0, 3, 5, 1, 4, 6, 2]
    assert sorted(S.interleave_index(11, 2)) == list(range(11))


# ---- volumetric helpers of the artifact stages, pinned by tests/golden/sr_units.npz ---------------------------
def test_mog_pinned(golden):
    g = golden("sr_units")
    c = g["mog_centers"]
    np.testing.assert_allclose(S.mog3d((24, 20, 28), c, np.full((3, 1), 4.0)).numpy(), g["mog_a"], atol=1e-6)
    np.testing.assert_allclose(S.mog3d((24, 20, 28), c, g["mog_sig"]).numpy(), g["mog_b"], atol=1e-6)
    np.testing.assert_allclose(S.mog3d((24, 20, 28), c, np.array([[6.0], [2.0], [11.0]])).numpy(), g["mog_c"], atol=1e-6)


@pytest.mark.parametrize("tag", ["p1", "p2", "p3"])
def test_perlin_pinned(golden, tag):
    g = golden("sr_units")
    cfg = g[f"perlin_{tag}_cfg"]
    shape, res, octv, inc = tuple(int(v) for v in cfg[:3]), int(cfg[3]), int(cfg[4]), float(cfg[5])
    torch.manual_seed(17)
    n, _ = S.fractal_noise(shape, res, octv, 0.5, 2, inc)
    np.testing.assert_allclose(n.numpy(), g[f"perlin_{tag}"], atol=2e-6)
    assert float(torch.rand(1)) == g[f"perlin_{tag}_next"][1]  # same number of torch draws consumed


def test_rigid_algebra_pinned(golden):
    g = golden("sr_units")
    ax = torch.from_numpy(g["ax"])
    m = S.axisangle2mat(ax)
    assert np.array_equal(m.numpy(), g["ax_mat"])
    assert np.array_equal(S.mat2axisangle(m).numpy(), g["ax_back"])
    a, b = S.axisangle2mat(ax[:20]), S.mat_last2first(m[20:])
    assert np.array_equal(S.compose(a, b).numpy(), g["compose"])
