"""CPU: the host side of the product (parameter tables, random plans) against the oracle and the
golden RNG tapes.  No kernel is launched here."""
import numpy as np
import pytest
import torch

from oracle import fsg_oracle as O
from fetalsyngen_amd import rng
from fetalsyngen_amd import tables as T
from tests.util_cases import E2E, t


@pytest.mark.parametrize("n_src,n_dst", [(3, 24), (5, 48), (40, 48), (1, 32), (15, 256), (85, 256), (171, 256), (256, 256)])
def test_zoom_table_matches_oracle(n_src, n_dst):
    f = n_dst / n_src
    tab = T.zoom_table(n_src, f, n_dst)
    lo, hi, w_lo, w_hi = O.zoom_axis_table(n_src, f, n_dst)
    assert np.array_equal(tab["lo"], lo.numpy()) and np.array_equal(tab["hi"], hi.numpy())
    assert np.array_equal(tab["w_lo"], w_lo.numpy()) and np.array_equal(tab["w_hi"], w_hi.numpy())
    assert tab.dtype.itemsize == 16


def test_memoised_zoom_table_lists_are_the_plain_ones():
    """tables.zoom_tables_between (memoised on the two shapes; three calls per sample on the host's critical path) returns
    the very table objects of zoom_tables with the factor written out as the call sites used to: shape / coarse shape for
    the deformation and bias grids, 1 / (new_size / size) for RandResample's zoom-back."""
    from fetalsyngen_amd import tables as T

    rs = np.random.RandomState(0)
    for _ in range(100):
        size = tuple(int(v) for v in rs.randint(20, 300, 3))
        small = tuple(int(v) for v in np.maximum(np.round(rs.uniform(0.03, 0.2) * np.array(size)), 1).astype(int))
        a, na = T.zoom_tables(small, np.array(size) / np.array(small))
        b, nb = T.zoom_tables_between(small, size)
        assert na == nb and all(x is y for x, y in zip(a, b))
        _stds, ns, fac, _tabs = T.resample_plan(size, [0.5] * 3, rs.uniform(0.5, 1.5, 3), 0.3)
        a, na = T.zoom_tables(ns, 1 / np.asarray(fac))
        b, nb = T.zoom_tables_between(tuple(ns), size, True)
        assert na == nb and all(x is y for x, y in zip(a, b)), (size, ns)
    assert T.zoom_tables_between((5, 6, 7), (50, 60, 70))[0] is T.zoom_tables_between((5, 6, 7), (50, 60, 70))[0]


def test_gaussian_taps_golden(golden):
    g = golden("gauss_taps")
    for i, s in enumerate(g["sigma"]):
        assert np.array_equal(T.gaussian_taps(float(s)), g[f"taps_{i}"])


def test_resample_plan_matches_oracle():
    for spacing, u in [([1.1] * 3, 0.3), ([0.5, 0.8, 1.3], 0.9), ([1.5] * 3, 0.0), ([0.5] * 3, 0.5)]:
        stds, new, fac, tabs = T.resample_plan((32, 40, 48), [0.5] * 3, spacing, u)
        s2, n2, f2, pos = O.resample_plan((32, 40, 48), [0.5] * 3, spacing, u)
        assert np.array_equal(stds, s2) and tuple(n2) == new and np.array_equal(fac, f2)
        for a in range(3):
            p32 = torch.tensor(pos[a], dtype=torch.float32)
            n = (32, 40, 48)[a]
            ok = ((p32 > 0) & (p32 <= n - 1)).numpy()
            assert np.array_equal(tabs[a]["lo"] >= 0, ok)
            lo = torch.floor(p32)
            assert np.array_equal(tabs[a]["lo"][ok], lo.numpy()[ok].astype(np.int32))
            assert np.array_equal(tabs[a]["w_hi"][ok], (p32 - lo).numpy()[ok])


@pytest.mark.parametrize("name", list(E2E))
def test_plans_consume_rng_like_the_reference(golden, name):
    """Run only the host plans of one sample under the golden seed (rng mode 'reference') and compare
    every drawn quantity with what the reference produced / recorded on its RNG tape."""
    from fetalsyngen_amd.generator.augmentation.synthseg import RandBiasField, RandGamma, RandNoise, RandResample
    from fetalsyngen_amd.generator.deformation.affine_nonrigid import SpatialDeformation
    from fetalsyngen_amd.generator.intensity.rand_gmm import ImageFromSeeds

    g = golden(name)
    kw = dict(E2E[name])
    prob = kw.get("prob", 1.0)
    shape = tuple(int(v) for v in g["shape"])
    size = kw.get("size", shape)
    ns = kw.get("nonlin_scale", (0.03, 0.06))
    bs = kw.get("bf_scale", (0.004, 0.02))
    np.random.seed(int(g["seed"]))
    torch.manual_seed(int(g["seed"]))
    with rng.use("reference"):
        ig = ImageFromSeeds(1, 6, O.DEFAULT_SEED_LABELS, O.DEFAULT_GEN_CLASSES)
        m2s = ig.draw_subclusters({})
        gp = ig.plan_intensities(shape, {})
        dp = SpatialDeformation(20, 0.02, 0.1, list(size), prob, True, ns[0], ns[1], 4, 0.5, "cuda:0").plan(shape)
        gam = RandGamma(prob, 0.1).plan({})
        bp = RandBiasField(prob, bs[0], bs[1], 0.01, 0.3).plan(shape, {})
        rp = RandResample(prob, 0.5, 1.5).plan(shape, np.array([0.5] * 3), {})
        npn = RandNoise(prob, 5, 15).plan(rp.new_size if rp.active else shape, {})
    assert [m2s[m] for m in range(1, 5)] == list(g["mlabel2subclusters"])
    assert np.array_equal(gp.mus.numpy(), g["mus"]) and np.array_equal(gp.sigmas.numpy(), g["sigmas"])
    assert dp.flip == bool(g["flip"])
    if "rotations" in g.files:
        assert dp.active
        assert np.array_equal(dp.params["affine"]["rotations"], g["rotations"])
        assert np.array_equal(dp.params["affine"]["shears"], g["shears"])
        assert np.array_equal(dp.params["affine"]["scalings"], g["scalings"])
        assert list(dp.params["non_rigid"]["size_F_small"]) == list(g["size_F_small"])
        assert dp.params["non_rigid"]["nonlin_std"] == float(g["nonlin_std"])
    else:
        assert not dp.active
    assert (gam is None and np.isnan(g["gamma"])) or gam == float(g["gamma"])
    if "bf_size" in g.files:
        assert list(bp.params["bf_size"]) == list(g["bf_size"])
    else:
        assert not bp.active
    if np.isnan(g["spacing"]).any():
        assert not rp.active
    else:
        assert np.array_equal(rp.spacing, g["spacing"])
    assert (not npn.active and np.isnan(g["noise_std"])) or npn.std32 == float(g["noise_std"])
    # the torch AND numpy streams are now exactly where the reference's were: replay the recorded
    # tape (same calls, same shapes) from the same seed and compare the next draws
    nxt_t, nxt_n = torch.rand(4), np.random.rand(4)
    np.random.seed(int(g["seed"]))
    torch.manual_seed(int(g["seed"]))
    for i, nme in enumerate(str(s) for s in g["tape_names"]):
        if f"tape_{i}" in g.files:
            shp, dt = g[f"tape_{i}"].shape, g[f"tape_{i}"].dtype
        else:
            shp, dt = tuple(int(v) for v in g[f"tape_{i}_shape"]), np.float32
        if nme.startswith("torch."):
            getattr(torch, nme[6:])(tuple(shp), dtype=torch.float64 if dt == np.float64 else torch.float32)
        elif nme == "np.randint":
            np.random.randint(1, 7)
        elif nme == "np.uniform":
            np.random.uniform(0.5, 1.5)
        else:
            getattr(np.random, nme[3:])(*shp)
    assert torch.equal(nxt_t, torch.rand(4))
    assert np.array_equal(nxt_n, np.random.rand(4))


def test_device_rng_mode_draws_keys_not_fields():
    torch.manual_seed(0)
    with rng.use("device"):
        f = rng.normal_field((256, 256, 256), stream_id=1)
    assert f.host is None and 0 <= f.seed < 2**62 and f.stream_id == 1
    torch.manual_seed(0)
    with rng.use("device"):
        f2 = rng.normal_field((256, 256, 256), stream_id=1)
    assert f.seed == f2.seed
    with pytest.raises(ValueError):
        rng.set_mode("nope")


def test_genparams_force_gates_and_none_stripping():
    from fetalsyngen_amd.generator.augmentation.synthseg import RandGamma, RandNoise, RandResample
    from tests.util_cases import make_generator

    np.random.seed(0)
    assert RandGamma(0.0, 0.1).plan({}) is None
    assert RandGamma(0.0, 0.1).plan({"gamma": 1.25}) == 1.25
    p = RandNoise(0.0, 5, 15).plan((4, 4, 4), {"noise_std": 7.0})
    assert p.active and p.std32 == 7.0
    r = RandResample(0.0, 0.5, 1.5).plan((32, 32, 32), np.array([0.5] * 3), {"spacing": [1.0, 1.0, 1.0]})
    assert r.active and r.new_size == (16, 16, 16)
    gen = make_generator((8, 8, 8), "cuda:0")
    assert gen._validated_genparams({"a": None, "b": {"c": None, "d": 1}}) == {"b": {"d": 1}}


def test_image_from_seeds_validation():
    from fetalsyngen_amd.generator.intensity.rand_gmm import ImageFromSeeds

    with pytest.raises(ValueError, match="unique"):
        ImageFromSeeds(1, 6, [0, 1, 1], [0, 1, 1])
    with pytest.raises(ValueError, match="same lengths"):
        ImageFromSeeds(1, 6, [0, 1, 2], [0, 1])


def test_load_seeds_sums_meta_labels():
    from fetalsyngen_amd.generator.intensity.rand_gmm import ImageFromSeeds
    from fetalsyngen_amd.phantom import combined_seed_labels, make_seed_volumes

    seg, seeds = make_seed_volumes((16, 16, 16))
    ig = ImageFromSeeds(1, 6, O.DEFAULT_SEED_LABELS, O.DEFAULT_GEN_CLASSES)
    np.random.seed(3)
    lab, sel = ig.load_seeds(seeds)
    assert lab.dtype == torch.int64
    assert np.array_equal(lab.numpy(), combined_seed_labels(seeds, sel["mlabel2subclusters"]).astype(np.int64))
    lab2, sel2 = ig.load_seeds(seeds, genparams={"mlabel2subclusters": {1: 2, 2: 2, 3: 2, 4: 2}})
    assert sel2["mlabel2subclusters"] == {1: 2, 2: 2, 3: 2, 4: 2}
    assert set(np.unique(lab2.numpy())) <= {0, 10, 11, 20, 21, 30, 31, 40, 41}


@pytest.mark.parametrize("mode", ["device", "reference"])
@pytest.mark.parametrize("prob", [1.0, 0.9, 0.5, 0.0])
def test_bulk_draws_equal_per_stage_plans(prob, mode):
    """`FetalSynthGen._draw_all_fast` (draws between two gates fetched with one numpy call, torch.rand(2n) for the two GMM
    tables) against the per-stage plan() functions: every drawn quantity identical, both global generators left at the same
    position -- for many seeds, gates passing and failing, nonlinear field on and off, shape != size (random shift)."""
    from tests.util_cases import make_generator

    def same(a, b):
        if a is None or b is None:
            return a is None and b is None
        if torch.is_tensor(a):
            return torch.is_tensor(b) and a.dtype == b.dtype and torch.equal(a, b)
        if isinstance(a, np.ndarray):
            return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
        if isinstance(a, dict):
            return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, (list, tuple)):
            return type(a) is type(b) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return type(a) is type(b) and a == b

    for shape, size, nonlin in (((32, 32, 32), None, True), ((40, 36, 28), (32, 32, 24), True), ((24, 24, 24), None, False)):
        gen = make_generator(shape, "cuda:0", prob=prob, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2), size=size, rng=mode)
        gen.spatial_deform.nonlinear_transform = nonlin
        for seed in range(25):
            outs = []
            for fast in (False, True):
                np.random.seed(seed)
                torch.manual_seed(seed)
                plans = gen.plan_only(shape, fast=fast)
                outs.append((plans, np.random.rand(3), torch.rand(3)))
            (pa, na, ta), (pb, nb, tb) = outs
            assert np.array_equal(na, nb) and torch.equal(ta, tb), (shape, seed)
            m2s_a, gmm_a, d_a, g_a, b_a, r_a, n_a = pa
            m2s_b, gmm_b, d_b, g_b, b_b, r_b, n_b = pb
            assert m2s_a == m2s_b and all(type(v) is int for v in m2s_b.values())
            assert same(gmm_a.mus, gmm_b.mus) and same(gmm_a.sigmas, gmm_b.sigmas)
            assert (gmm_a.field.seed, gmm_a.field.stream_id) == (gmm_b.field.seed, gmm_b.field.stream_id)
            assert same(gmm_a.field.host, gmm_b.field.host)
            assert d_a.active == d_b.active and d_a.flip == d_b.flip and same(d_a.A, d_b.A) and same(d_a.c2, d_b.c2)
            assert same(d_a.field_small, d_b.field_small) and same(d_a.params, d_b.params)
            assert same(g_a, g_b)
            assert b_a.active == b_b.active and same(b_a.grid, b_b.grid) and same(b_a.params, b_b.params)
            assert r_a.active == r_b.active and same(r_a.spacing, r_b.spacing) and same(r_a.stds, r_b.stds)
            assert r_a.new_size == r_b.new_size and same(r_a.factors, r_b.factors)
            assert (r_a.tabs is None and r_b.tabs is None) or all(x is y for x, y in zip(r_a.tabs, r_b.tabs))
            assert n_a.active == n_b.active and n_a.std32 == n_b.std32
            if n_a.active:
                assert (n_a.field.seed, n_a.field.shape) == (n_b.field.seed, n_b.field.shape) and same(n_a.field.host, n_b.field.host)
