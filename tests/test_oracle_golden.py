"""The oracle (oracle/fsg_oracle.py) against golden vectors captured from the real
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import fsg_oracle as O


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_affine(golden):
    g = golden("affine")
    for r, s, c, A in zip(g["rot"], g["shear"], g["scale"], g["A"]):
        assert np.array_equal(O.affine_matrix(r, s, c), A)


def test_gaussian_taps(golden):
    g = golden("gauss_taps")
    for i, s in enumerate(g["sigma"]):
        assert np.array_equal(O.gaussian_taps(float(s)).numpy(), g[f"taps_{i}"])


def test_blur(golden):
    g = golden("blur")
    for si in range(3):
        x = t(g[f"x_{si}"])
        for ti, st in enumerate(g["stds"]):
            y = O.blur3d(x, st).numpy()
            # same conv3d primitive; allow for thread-count dependent summation order
            np.testing.assert_allclose(y, g[f"y_{si}_{ti}"], rtol=1e-6, atol=1e-4)


def test_zoom(golden):
    g = golden("zoom")
    for i in range(int(g["ncases"])):
        y = O.linear_zoom(t(g[f"x_{i}"]), g[f"factor_{i}"]).numpy()
        assert np.array_equal(y, g[f"y_{i}"]), i


def test_interp(golden):
    g = golden("interp")
    for ci in range(2):
        x = t(g[f"x_{ci}"])
        for tag in ("raw", "clamped"):
            co = g[f"coords_{ci}_{tag}"]
            ii, jj, kk = (t(co[a]) for a in range(3))
            assert np.array_equal(O.sample_linear(x, ii, jj, kk).numpy(), g[f"lin_{ci}_{tag}"])
            assert np.array_equal(O.sample_nearest(x, ii, jj, kk).numpy(), g[f"nn_{ci}_{tag}"])


def test_deform_image(golden):
    g = golden("deform_image")
    for i in range(int(g["ncases"])):
        shape, size = tuple(g[f"shape_{i}"]), tuple(g[f"size_{i}"])
        field = None
        if f"Fsmall_{i}" in g.files:
            field = O.nonlinear_field(t(g[f"Fsmall_{i}"]), shape)
        ii, jj, kk, m = O.deformation_coords(shape, size, t(g[f"A_{i}"]), t(g[f"c2_{i}"]), field)
        assert np.array_equal(torch.stack([ii, jj, kk]).numpy(), g[f"coords_{i}"]), i
        assert np.array_equal(m, g[f"margins_{i}"]), i
    assert g["margins_1"][0] > 0  # the engineered non-zero-margin case really has one


def test_gmm(golden):
    g = golden("gmm")
    seeds = t(g["seeds"].astype(np.int64))
    for k, gc in enumerate([O.DEFAULT_GEN_CLASSES, O.DEFAULT_SEED_LABELS]):
        torch.manual_seed(0)
        u_mu, u_sg = torch.rand(50), torch.rand(50)
        z_cls = torch.randn(41) if k == 0 else None
        z = torch.randn(seeds.shape)
        assert np.array_equal(u_mu.numpy(), g[f"tape{k}_0"])  # torch CPU stream is what the tape says
        mus, sig = O.gmm_tables(u_mu, u_sg, z_cls, O.DEFAULT_SEED_LABELS, gc)
        assert np.array_equal(mus.numpy(), g[f"mus_{k}"])
        assert np.array_equal(sig.numpy(), g[f"sigmas_{k}"])
        assert np.array_equal(O.gmm_image(seeds, mus, sig, z).numpy(), g[f"img_{k}"])


def _replay_stages(g, seed, gates):
    x = t(g["x"])
    p = 1.0 if gates == "on" else 0.0
    np.random.seed(seed)
    torch.manual_seed(seed)
    key = f"s{seed}_{gates}"
    out = {}
    a = x
    if np.random.rand() < p:
        a = O.gamma_transform(x, float(np.exp(0.1 * np.random.randn(1)[0])))
    out["gamma"] = a
    b = a
    if np.random.rand() < p:
        bscale = 0.05 + np.random.rand(1) * (0.2 - 0.05)
        bsize = np.maximum(np.round(bscale * np.array(a.shape)).astype(int), 1).tolist()
        bstd = 0.01 + (0.3 - 0.01) * np.random.rand(1)
        b = O.bias_multiply(a, torch.tensor(bstd, dtype=torch.float32) * torch.randn(bsize))
    out["bias"] = b
    c, factors = b, None
    if np.random.rand() < p:
        spacing = np.array([1.0, 1.0, 1.0]) * np.random.uniform(0.5, 1.5)
        c, factors = O.resample_down(b, [0.5] * 3, spacing, np.random.rand())
    out["resampled"] = c
    d = c
    if np.random.rand() < p:
        nstd = 5 + (15 - 5) * np.random.rand(1)
        d = O.add_noise(c, nstd, torch.randn(c.shape))
    out["noisy"] = d
    out["back"] = O.resize_back(d, factors)
    return key, out


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("gates", ["on", "off"])
def test_stages(golden, seed, gates):
    g = golden("stages")
    key, out = _replay_stages(g, seed, gates)
    for name, v in out.items():
        ref = g[f"{key}_{name}"]
        assert v.shape == ref.shape, name
        if name in ("gamma", "bias"):
            assert np.array_equal(v.numpy(), ref), name
        else:  # downstream of conv3d
            np.testing.assert_allclose(v.numpy(), ref, rtol=1e-6, atol=1e-4, err_msg=name)


def test_stages_aniso(golden):
    g = golden("stages")
    np.random.seed(5)
    np.random.rand()  # the gate draw is consumed even when `spacing` is forced (synthseg.py:63)
    c, factors = O.resample_down(t(g["x"]), [0.5] * 3, [0.5, 0.8, 1.3], np.random.rand())
    np.testing.assert_allclose(c.numpy(), g["aniso_resampled"], rtol=1e-6, atol=1e-4)
    assert np.array_equal(factors, g["aniso_factors"])
    np.testing.assert_allclose(O.resize_back(c, factors).numpy(), g["aniso_back"], rtol=1e-6, atol=1e-6)


E2E = {
    "e2e_32_s0": dict(),
    "e2e_32_s1": dict(nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2)),
    "e2e_32_s2": dict(prob=0.5),
    "e2e_48_s0": dict(nonlin_scale=(0.08, 0.2)),
    "e2e_nc_s1": dict(nonlin_scale=(0.1, 0.2)),
    "e2e_sz_s4": dict(nonlin_scale=(0.1, 0.2), size=(32, 32, 32)),
}


@pytest.mark.parametrize("name", list(E2E))
def test_e2e(golden, name):
    from fetalsyngen_amd.phantom import make_seed_volumes

    g = golden(name)
    shape = tuple(int(v) for v in g["shape"])
    kw = dict(E2E[name])
    prob = kw.pop("prob", 1.0)
    cfg = O.Config(shape, prob=prob, **kw)
    seg, seeds = make_seed_volumes(shape, int(g["variant"]))
    np.random.seed(int(g["seed"]))
    torch.manual_seed(int(g["seed"]))
    r = O.run_sample(cfg, t(seg), seeds, keep_stages=True)
    assert np.array_equal(r["params"]["mus"].numpy(), g["mus"])
    assert bool(g["flip"]) == r["params"]["flip"]
    assert np.array_equal(r["seg"].numpy().astype(np.uint8), g["seg_out"])
    for k, v in r["stages"].items():
        if f"stage_{k}" in g.files:
            ref = g[f"stage_{k}"]
            if k in ("gmm", "coords", "warped", "gamma", "bias"):
                assert np.array_equal(v.numpy(), ref), k
            else:
                np.testing.assert_allclose(v.numpy(), ref, rtol=1e-6, atol=1e-4, err_msg=k)
    np.testing.assert_allclose(r["out"].numpy(), g["out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["scaled"].numpy(), g["scaled"], rtol=0, atol=1e-6)


def test_config1_sta21_128(golden):
    """BASELINE config 1: sub-sta21 decimated to 128^3, seed 0, YAML probabilities."""
    g = golden("config1_sta21_128")
    m2s = g["mlabel2subclusters"]
    # the combined seed map is what load_seeds would have summed; feed it as meta-label 1
    # and zeros for the others so the replay draws the same randints and sums to the same map
    comb = g["seeds_in"].astype(np.int64)
    zero = np.zeros_like(comb)
    seeds = {n: {m: (comb if m == 1 else zero) for m in range(1, 5)} for n in range(1, 7)}
    cfg = O.Config((128, 128, 128), resolution=(1.0, 1.0, 1.0), prob=0.9, res_range=(1.0, 3.0))
    np.random.seed(0)
    torch.manual_seed(0)
    r = O.run_sample(cfg, t(g["seg_in"].astype(np.float32)), seeds)
    assert [r["params"]["mlabel2subclusters"][m] for m in range(1, 5)] == list(m2s)
    so = r["seg"].numpy().astype(np.uint8)
    assert np.array_equal(np.bincount(so.reshape(-1), minlength=8), g["label_counts"])
    assert np.array_equal(so[::4, ::4, ::4], g["seg_sub4"])
    sc = r["scaled"].numpy()
    np.testing.assert_allclose(sc[::8, ::8, ::8], g["sub8"], rtol=0, atol=2e-6)
    st = np.array([sc.min(), sc.max(), sc.mean(dtype=np.float64), sc.std(dtype=np.float64)])
    np.testing.assert_allclose(st, g["stats"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sc[64], g["slice_x"].astype(np.float32), atol=1e-3)


def test_sublattice_option_equals_the_full_evaluation():
    """`index=` of linear_zoom / deformation_coords (used by the 384^3 GPU test) selects exactly the values of
    the full evaluation."""
    import numpy as np
    import torch

    from oracle import fsg_oracle as O

    torch.manual_seed(3)
    shape = (40, 36, 28)
    fs = 2.5 * torch.randn(5, 4, 3, 3)
    index = [np.unique(np.r_[np.arange(0, n, 7), n - 1]) for n in shape]
    factor = np.array(shape) / np.array(fs.shape[:3])
    full = O.linear_zoom(fs, factor)
    sub = O.linear_zoom(fs, factor, index=index)
    assert torch.equal(sub, full[np.ix_(*index)])
    A = torch.tensor(O.affine_matrix([0.2, -0.1, 0.15], [0.01, -0.02, 0.0], [1.05, 0.95, 1.0]), dtype=torch.float32)
    c2 = O.centre_with_shift(shape, shape)
    fi, fj, fk, fm = O.deformation_coords(shape, shape, A, c2, full)
    si, sj, sk, sm = O.deformation_coords(shape, shape, A, c2, sub, index=index)
    assert tuple(fm[:3]) == (0, 0, 0) and tuple(sm[:3]) == (0, 0, 0)
    ix = np.ix_(*index)
    assert torch.equal(si, fi[ix]) and torch.equal(sj, fj[ix]) and torch.equal(sk, fk[ix])


def test_reference_loop_form_of_the_zoom_equals_the_vectorised_form():
    """`REFERENCE_LOOPS` (what bench.py's cpu_baseline times: the reference's per-slice loops, utils/generation.py:374-386)
    changes the cost structure of the oracle's zoom, not one value."""
    import numpy as np
    import torch

    from oracle import fsg_oracle as O

    rs = np.random.RandomState(3)
    for shape, f in (((7, 5, 6), (2.3, 1.9, 3.4)), ((9, 8, 7, 3), (1.5, 2.0, 2.5)), ((12, 10, 8), (0.6, 0.75, 0.5))):
        x = torch.from_numpy(rs.rand(*shape).astype(np.float32))
        a = O.linear_zoom(x, np.array(f))
        O.REFERENCE_LOOPS = True
        try:
            b = O.linear_zoom(x, np.array(f))
        finally:
            O.REFERENCE_LOOPS = False
        assert torch.equal(a, b)
