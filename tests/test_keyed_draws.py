"""Keyed mode (`rng="keyed"`, csrc/fsg_keyed.hip): a sample is a function of its 64-bit key.

CPU: the Philox core against the published Random123 known-answer vectors; the C host draws against the oracle-side numpy
restatement (oracle/fsg_keyed_draws.py -- the reference's arithmetic from draw to parameter, paths cited there); gate
frequencies and ranges (the reference's distributions: SURVEY 8(a) rows K1, K2a, K2b, K5a, K5b, K7, K8); host cost.
GPU (`-m gpu`): the draw kernel against the restatement; whole keyed samples against the pinned oracle fed the exported
draws (labels bit-exact, image 2e-5 -- the same bar as the other modes); determinism and independence properties.
"""
import ctypes as C
import time

import numpy as np
import pytest
import torch

from oracle import fsg_keyed_draws as R
from oracle import fsg_oracle as O
from tests.util_cases import make_generator


def _ctx(shape, **kw):
    from fetalsyngen_amd import keyed

    gen = make_generator(shape, "cuda:0", rng="keyed", **kw)
    kc = keyed.KeyedContext(gen, shape)
    return gen, kc, keyed.config_dict(kc.cfg)


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        got = tuple(int(v) for v in R.philox4x32_10(*ctr, *key))
        assert got == want, (hex(got[0]), hex(want[0]))


@pytest.mark.parametrize("case", [dict(shape=(256, 256, 256), prob=1.0), dict(shape=(64, 56, 48), prob=0.5, size=(48, 48, 40)),
                                  dict(shape=(96, 96, 96), prob=0.9, nonlin_scale=(0.08, 0.2), bf_scale=(0.02, 0.1),
                                       res_range=(0.5, 2.5))])
def test_c_draws_equal_the_numpy_restatement(case):
    from fetalsyngen_amd import sharding

    shape = case.pop("shape")
    _gen, kc, cfg = _ctx(shape, **case)
    seen = {k: 0 for k in ("deform", "gamma", "bias", "resample", "noise")}
    n = 300
    for i in range(n):
        key = sharding.sample_key(77, i)
        d, r = kc.draws(key), R.host_draws(cfg, key)
        assert d.key == key and list(d.subclusters)[: cfg["meta_labels"]] == r["subclusters"]
        assert bool(d.deform_active) == r["deform_active"] and bool(d.gamma_active) == r["gamma_active"]
        assert bool(d.bias_active) == r["bias_active"] and bool(d.resample_active) == r["resample_active"]
        assert bool(d.noise_active) == r["noise_active"] and list(d.low_shape) == r["low_shape"]
        if r["deform_active"]:
            seen["deform"] += 1
            assert bool(d.flip) == r["flip"]
            # exact draws; cos / sin / exp / log come from glibc here and from numpy there: a few ulp at most
            np.testing.assert_allclose(np.array(d.rotations), r["rotations"], rtol=1e-15, atol=0)
            np.testing.assert_allclose(np.array(d.shears), r["shears"], rtol=1e-15, atol=0)
            np.testing.assert_allclose(np.array(d.scalings), r["scalings"], rtol=1e-15, atol=0)
            np.testing.assert_allclose(np.array(d.A).reshape(3, 3), r["A"], rtol=0, atol=2e-7)
            np.testing.assert_allclose(np.array(d.c2), r["c2"], rtol=1e-15, atol=0)
            if r["nonlinear"]:
                assert list(d.field_dims) == r["field_dims"]
                assert d.nonlin_scale == r["nonlin_scale"] and d.nonlin_std == r["nonlin_std"]
        if r["gamma_active"]:
            seen["gamma"] += 1
            assert abs(d.gamma - r["gamma"]) <= 1e-14 * r["gamma"]
        if r["bias_active"]:
            seen["bias"] += 1
            assert list(d.bias_dims) == r["bias_dims"] and d.bf_scale == r["bf_scale"] and d.bf_std == r["bf_std"]
        if r["resample_active"]:
            seen["resample"] += 1
            assert d.spacing == r["spacing"] and d.u_std == r["u_std"]
            np.testing.assert_allclose(np.array(d.stds), r["stds"], rtol=1e-15, atol=0)
            for a in range(3):
                assert d.blur_ntaps[a] == (2 * int(np.ceil(3 * r["stds"][a])) + 1 if r["stds"][a] > 0 else 0)
        if r["noise_active"]:
            seen["noise"] += 1
            assert d.noise_std == r["noise_std"] and d.noise_std32 == np.float32(r["noise_std"])
        assert d.block_bytes <= kc.block_bytes and d.off_field % 256 == 0 and d.off_bias % 256 == 0
    p = cfg["deform_prob"]
    for k, v in seen.items():  # gates fire at the configured rate (binomial, 5 sigma)
        assert abs(v - p * n) <= 5 * np.sqrt(max(p * (1 - p), 1e-9) * n) + 1e-9, (k, v)


def test_draw_ranges_are_the_reference_distributions():
    from fetalsyngen_amd import sharding

    _gen, kc, cfg = _ctx((256, 256, 256), prob=1.0)
    D = [kc.draws(sharding.sample_key(5, i)) for i in range(2000)]
    rot = np.array([list(d.rotations) for d in D]) * 180 / np.pi
    assert rot.min() >= -20 and rot.max() <= 20 and abs(rot.mean()) < 0.7 and abs(rot.std() - 40 / np.sqrt(12)) < 0.4
    sc = np.array([list(d.scalings) for d in D])
    assert sc.min() >= 0.9 and sc.max() <= 1.1
    sub = np.array([list(d.subclusters) for d in D])
    assert sub.min() == 1 and sub.max() == 6 and all(abs((sub == v).mean() - 1 / 6) < 0.02 for v in range(1, 7))
    sp = np.array([d.spacing for d in D])
    assert sp.min() >= 0.5 and sp.max() <= 1.5 and abs(sp.mean() - 1.0) < 0.03
    m = np.array([d.low_shape[0] for d in D])
    assert m.min() >= 85 and m.max() <= 256
    g = np.log(np.array([d.gamma for d in D]))
    assert abs(g.mean()) < 0.01 and abs(g.std() - 0.1) < 0.006       # exp(0.1 N(0,1))
    ns = np.array([d.noise_std for d in D])
    assert ns.min() >= 5 and ns.max() <= 15
    fl = np.array([d.flip for d in D])
    assert abs(fl.mean() - 0.5) < 0.05
    fd = np.array([d.field_dims[0] for d in D])
    assert fd.min() >= 8 and fd.max() <= 15


def test_keyed_host_cost_and_process_independence():
    """The C draws of a key do not depend on what was drawn before; a key costs microseconds on the host."""
    from fetalsyngen_amd import sharding

    _gen, kc, _cfg = _ctx((256, 256, 256), prob=0.9)
    keys = [sharding.sample_key(1, i) for i in range(50)]
    a = [bytes(kc.draws(k)) for k in keys]
    b = [bytes(kc.draws(k)) for k in reversed(keys)][::-1]
    assert a == b and len(set(a)) == len(a)
    t0 = time.perf_counter()
    for k in range(5000):
        kc.draws(k)
    assert (time.perf_counter() - t0) / 5000 < 30e-6


# ---- GPU ---------------------------------------------------------------------------------------------------------------
DEV = "cuda:0"


def _export(kc, d, block):
    """Exported draws + the device block read back -> the oracle's `draws=` dict."""
    f = block.view(torch.float32).cpu()
    ex = {"m2s": {m + 1: int(d.subclusters[m]) for m in range(4)},
          "mus": f[d.off_mus // 4: d.off_mus // 4 + d.ntab].clone(), "sigmas": f[d.off_sigmas // 4: d.off_sigmas // 4 + d.ntab].clone(),
          "deform": None, "gamma": d.gamma if d.gamma_active else None, "bias": None, "resample": None,
          "noise_std": d.noise_std if d.noise_active else None}
    if d.deform_active:
        fs = None
        if d.nonlinear:
            n = int(np.prod(list(d.field_dims))) * 3
            fs = f[d.off_field // 4: d.off_field // 4 + n].reshape(*d.field_dims, 3).clone()
        ex["deform"] = {"flip": bool(d.flip), "A": torch.tensor(np.array(d.A, dtype=np.float32).reshape(3, 3)),
                        "c2": torch.tensor(np.array(d.c2, dtype=np.float64)), "f_small": fs}
    if d.bias_active:
        n = int(np.prod(list(d.bias_dims)))
        ex["bias"] = f[d.off_bias // 4: d.off_bias // 4 + n].reshape(*d.bias_dims).clone()
    if d.resample_active:
        ex["resample"] = {"spacing": d.spacing, "u_std": d.u_std}
    return ex


def _oracle(kc, K, shape, key, seg, seeds, cfg_kw):
    from fetalsyngen_amd import _lib

    d = kc.draws(key)
    block = torch.empty(kc.block_bytes, dtype=torch.uint8, device=DEV)
    _lib.check(kc.lib.fsg_keyed_fill_block(kc.handle, C.byref(d), C.c_void_p(block.data_ptr()), K._stream(block)), "fill")
    torch.cuda.synchronize()
    ex = _export(kc, d, block)
    r = O.run_sample(O.Config(shape, **cfg_kw), torch.from_numpy(seg), seeds, draws=ex,
                     noise_gmm=lambda shp: K.randn(shp, key, 1, DEV).cpu(), noise_lowres=lambda shp: K.randn(shp, key, 2, DEV).cpu())
    return d, ex, r


@pytest.fixture(scope="module")
def K():
    from fetalsyngen_amd import kernels

    return kernels


@pytest.mark.gpu
def test_draw_kernel_equals_the_restatement(K):
    from fetalsyngen_amd import _lib, sharding

    shape = (96, 96, 96)
    _gen, kc, cfg = _ctx(shape, prob=1.0, nonlin_scale=(0.08, 0.2), bf_scale=(0.02, 0.1))
    for i in range(6):
        key = sharding.sample_key(3, i)
        d = kc.draws(key)
        block = torch.full((kc.block_bytes,), 0xAB, dtype=torch.uint8, device=DEV)
        _lib.check(kc.lib.fsg_keyed_fill_block(kc.handle, C.byref(d), C.c_void_p(block.data_ptr()), K._stream(block)), "fill")
        torch.cuda.synchronize()
        ex = _export(kc, d, block)
        mus, sigmas = R.gmm_tables(cfg, key)
        assert np.array_equal(ex["sigmas"].numpy(), sigmas)                       # uniforms: exact
        untied = [l for l in range(cfg["nlabels"]) if l not in cfg["seed_labels"]]
        assert np.array_equal(ex["mus"].numpy()[untied], mus[untied])
        np.testing.assert_allclose(ex["mus"].numpy(), mus, rtol=0, atol=2e-4)      # tied means: 25 * (GPU normal)
        z = R.device_normals(key, 3, int(np.prod(list(d.field_dims))) * 3) * np.float32(d.nonlin_std)
        np.testing.assert_allclose(ex["deform"]["f_small"].numpy().reshape(-1), z, rtol=0, atol=4e-5 * max(d.nonlin_std, 1e-3))
        zb = R.device_normals(key, 4, int(np.prod(list(d.bias_dims)))) * np.float32(d.bf_std)
        np.testing.assert_allclose(ex["bias"].numpy().reshape(-1), zb, rtol=0, atol=4e-5 * d.bf_std)
        i32 = block.view(torch.int32).cpu().numpy()
        assert list(i32[d.off_mm8 // 4: d.off_mm8 // 4 + 8]) == [0x7F800000] * 4 + [-2139095041] * 4
        slots = i32[d.off_slots // 4: d.off_slots // 4 + 64 * 16].reshape(64, 16)
        assert (slots[:, 0] == 0x7F800000).all() and (slots[:, 1] == -2139095041).all() and (slots[:, 2:] == 0).all()
        # nothing outside the regions the draws name is written
        used = np.zeros(kc.block_bytes, dtype=bool)
        for off, nbytes in ((d.off_mm8, 32), (d.off_slots, 4096), (d.off_mus, 4 * d.ntab), (d.off_sigmas, 4 * d.ntab),
                            (d.off_bias, 4 * int(np.prod(list(d.bias_dims)))), (d.off_field, 12 * int(np.prod(list(d.field_dims))))):
            used[off: off + nbytes] = True
        assert (block.cpu().numpy()[~used] == 0xAB).all()


@pytest.mark.gpu
@pytest.mark.parametrize("prob", [1.0, 0.5])
def test_keyed_samples_equal_the_oracle_on_the_exported_draws(K, prob):
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (48, 48, 48)
    kw = dict(prob=prob, nonlin_scale=(0.08, 0.2), bf_scale=(0.05, 0.2))
    seg, seeds = make_seed_volumes(shape)
    gen, kc, _cfg = _ctx(shape, **kw)
    kc = gen.keyed_context(shape)
    bank, seg_d = SeedBank(seeds, DEV), torch.from_numpy(seg).to(DEV)
    combos = set()
    for i in range(14 if prob < 1 else 4):
        key = sharding.sample_key(11, i)
        out, seg_o, _img, params = gen._pipeline(None, seg_d, bank, {}, scale01=True, key=key)
        d, ex, r = _oracle(kc, K, shape, key, seg, seeds, kw)
        combos.add((d.deform_active, d.gamma_active, d.bias_active, d.resample_active, d.noise_active))
        assert np.array_equal(seg_o.cpu().numpy().astype(np.uint8), r["seg"].numpy().astype(np.uint8)), (i, "labels")
        np.testing.assert_allclose(out.cpu().numpy(), r["scaled"].numpy(), rtol=0, atol=2e-5, err_msg=f"sample {i}")
        assert params["key"] == key and params["selected_seeds"]["mlabel2subclusters"] == ex["m2s"]
        assert torch.equal(params["seed_intensities"]["mus"].cpu(), ex["mus"])
        assert (params["deform_params"]["affine"] is None) == (not d.deform_active)
        assert (params["resample_params"]["spacing"] is None) == (not d.resample_active)
    if prob < 1:
        assert len(combos) >= 6  # the gates really were exercised in combination


@pytest.mark.gpu
def test_keyed_sample_at_256_equals_the_oracle(K):
    """BASELINE configs[1] in keyed mode: one 256^3 volume against the pinned oracle on the exported draws."""
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (256, 256, 256)
    seg, seeds = make_seed_volumes(shape, 2)
    gen = make_generator(shape, DEV, rng="keyed")
    kc = gen.keyed_context(shape)
    bank, seg_d = SeedBank(seeds, DEV), torch.from_numpy(seg).to(DEV)
    key = sharding.sample_key(1234, 17)
    out, seg_o, _img, params = gen._pipeline(None, seg_d, bank, {}, scale01=True, key=key)
    out2, seg2, _i, _p = gen._pipeline(None, seg_d, bank, {}, scale01=True, key=key)
    assert torch.equal(out, out2) and torch.equal(seg_o, seg2)
    _d, _ex, r = _oracle(kc, K, shape, key, seg, seeds, dict(prob=1.0))
    assert np.array_equal(seg_o.cpu().numpy().astype(np.uint8), r["seg"].numpy().astype(np.uint8)), "labels bit-exact at 256^3"
    np.testing.assert_allclose(out.cpu().numpy(), r["scaled"].numpy(), rtol=0, atol=2e-5)
    o = out.cpu().numpy()
    assert o.min() == 0.0 and o.max() == 1.0


@pytest.mark.gpu
def test_keyed_samples_depend_on_the_key_only(K):
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import MemorySynthDataset
    from fetalsyngen_amd.data.staging import PrefetchingStream
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (64, 56, 72)
    segs, banks = zip(*[make_seed_volumes(shape, v) for v in range(3)])
    kw = dict(prob=0.8, nonlin_scale=(0.08, 0.2), bf_scale=(0.05, 0.2))
    gen = make_generator(shape, DEV, rng="keyed", **kw)
    ds = MemorySynthDataset(gen, list(segs), list(banks), base_seed=21)
    np.random.seed(0)
    torch.manual_seed(0)
    state = (np.random.get_state()[1].copy(), torch.get_rng_state().clone())
    ref = {}
    for i in range(9):
        bank, seg, _t = ds._subject(i % 3)
        ref[i] = gen._pipeline(None, seg, bank, {}, scale01=True, key=sharding.sample_key(21, i))
    # the global generators were not touched
    assert np.array_equal(np.random.get_state()[1], state[0]) and torch.equal(torch.get_rng_state(), state[1])
    # another generator object, another order, uint8 labels: the same volumes
    gen2 = make_generator(shape, DEV, rng="keyed", **kw)
    ds2 = MemorySynthDataset(gen2, list(segs), list(banks), base_seed=21)
    for i in (7, 2, 8, 0):
        bank, seg, _t = ds2._subject(i % 3)
        o, s_, _i, _p = gen2._pipeline(None, seg, bank, {}, scale01=True, key=sharding.sample_key(21, i), labels_u8=True)
        assert torch.equal(o, ref[i][0]) and s_.dtype == torch.uint8 and torch.equal(s_, ref[i][1].to(torch.uint8))
    assert not torch.equal(ref[0][0], ref[3][0])  # same subject, other key
    # through the dataset (keys (base_seed, idx)), the prefetching stream (single and batched) and sample_batch
    for i in range(3):
        item = ds[i]
        assert torch.equal(item["image"][0], ref[i][0].cpu()) and torch.equal(item["label"][0], ref[i][1].cpu().long())
    for kwargs in (dict(to_host=False), dict(to_host=True, depth=2), dict(to_host=False, batch_size=3, batch_streams=2)):
        imgs = []
        for g in PrefetchingStream(ds, range(9), base_seed=21, **kwargs):  # host items are views of a pinned ring: copy now
            if kwargs.get("batch_size"):
                imgs += [g["image"][b, 0].clone() for b in range(g["image"].shape[0])]
            else:
                imgs.append(g["image"][0].clone())
        for i in range(9):
            assert torch.equal(imgs[i].cpu(), ref[i][0].cpu()), (kwargs, i)
    # schema of the params dictionary = the other modes'
    gen3 = make_generator(shape, DEV, rng="device", **kw)
    np.random.seed(1)
    torch.manual_seed(1)
    bank, seg, _t = ds._subject(0)
    _o, _s, _i, p_dev = gen3._pipeline(None, seg, bank, {}, scale01=True)
    assert set(p_dev) <= set(ref[0][3]) and set(ref[0][3]) - set(p_dev) == {"key"}


# ---- the subject's seed volumes as one code volume (fetalsyngen_amd/seedcodes.py, fsg_sample_head_codes_f32) --------------------
def _phantom_parts(shape, variant=0):
    from fetalsyngen_amd.phantom import make_seed_volumes

    _seg, seeds = make_seed_volumes(shape, variant)
    return [torch.from_numpy(seeds[n][m].astype(np.uint8)) for n in range(1, 7) for m in range(1, 5)]


def test_seed_codes_stand_for_the_byte_wise_sum_of_the_selected_volumes():
    """reference rand_gmm.py:91-99: the seed label map is the sum of one volume per meta label; the code volume + tuple rows
    give the same labels for every selection."""
    from fetalsyngen_amd import seedcodes as SC

    parts = _phantom_parts((40, 36, 28), 1)
    codes, tuples = SC.build(parts, 25)
    assert codes.dtype == torch.int16 and tuples.dtype == torch.uint8 and tuples.shape[1] == 25
    assert int(codes.max()) + 1 == tuples.shape[0] <= SC.CODES_MAX and (tuples[:, 24] == 0).all()
    rs = np.random.RandomState(3)
    for _ in range(8):
        ns = rs.randint(1, 7, 4)
        sel = [4 * (ns[m] - 1) + m for m in range(4)]
        want = (sum(parts[s].to(torch.int64) for s in sel) & 255).to(torch.uint8)
        assert torch.equal(SC.labels_of(codes, tuples, sel), want)
    # three meta labels: the fourth selects the zero byte of a row
    sel3 = [0, 4 * 2 + 1, 4 * 5 + 2, 24]
    assert torch.equal(SC.labels_of(codes, tuples, sel3), (sum(parts[s].to(torch.int64) for s in sel3[:3]) & 255).to(torch.uint8))
    # more distinct columns than a workgroup's table holds: no codes (the caller keeps the four volumes)
    wide = [torch.arange(4096, dtype=torch.int64).remainder(256).to(torch.uint8), (torch.arange(4096) // 256).to(torch.uint8)]
    assert SC.build(wide, 3) is None
    with pytest.raises(ValueError):
        SC.build(parts, 24)


@pytest.mark.gpu
def test_code_volume_head_is_bit_identical_to_the_four_volume_head(K):
    """From a subject's second sample on the head kernel reads its code volume (6 instead of 8 B/voxel): same volumes as with
    the four label volumes, also after a seed volume was rewritten in place, and `invalidate_label_twins` drops the codes."""
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    for shape in ((64, 56, 72), (128, 128, 128)):
        seg, seeds = make_seed_volumes(shape, 1)
        kw = dict(nonlin_scale=(0.08, 0.2), bf_scale=(0.05, 0.2))
        seg_d = torch.from_numpy(seg).to(DEV)
        outs = {}
        for use in (False, True):
            gen = make_generator(shape, DEV, rng="keyed", **kw)
            kc = gen.keyed_context(shape)
            kc.use_codes = use
            bank = SeedBank(seeds, DEV)
            res = []
            for i in range(5):
                res.append(gen._pipeline(None, seg_d, bank, {}, scale01=True, key=sharding.sample_key(5, i))[:2])
            assert (getattr(bank, "_seed_codes", None) is not None) == use
            if use:
                assert bank._seed_codes[2] is not None and bank._seed_codes[2][1].shape[0] <= 256
            # a seed volume rewritten through torch: the codes follow (their build is keyed to the volumes' versions)
            bank.vol[3][2].copy_(torch.roll(bank.vol[3][2], 5, 0))
            for i in range(5, 9):
                res.append(gen._pipeline(None, seg_d, bank, {}, scale01=True, key=sharding.sample_key(5, i))[:2])
            gen.invalidate_label_twins()
            assert getattr(bank, "_seed_codes", None) is None
            res.append(gen._pipeline(None, seg_d, bank, {}, scale01=True, key=sharding.sample_key(5, 9))[:2])
            outs[use] = res
        for a, b in zip(outs[False], outs[True]):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # the library-side A/B switch (FSG_TUNE_NO_SEED_CODES): a plan that carries codes is served from the four volumes
        from fetalsyngen_amd import _lib as _L

        gen = make_generator(shape, DEV, rng="keyed", **kw)
        bank = SeedBank(seeds, DEV)
        _L.load().fsg_set_tuning(65536)
        try:
            got = [gen._pipeline(None, seg_d, bank, {}, scale01=True, key=sharding.sample_key(5, i))[:2] for i in range(3)]
        finally:
            _L.load().fsg_set_tuning(0)
        assert getattr(bank, "_seed_codes", None) is not None
        for a, b in zip(outs[False][:3], got):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # the selection really changed between samples (otherwise the test would not see a wrong tuple column)
        assert len({tuple(gen.keyed_context(shape).draws(sharding.sample_key(5, i)).subclusters[:4]) for i in range(9)}) > 3


@pytest.mark.gpu
def test_device_code_builder_is_exact(K):
    """fsg_seed_codes_build (one pass, hash set of columns) against the torch formulation: the same partition of the voxels and
    the same labels for every selection (the numbering of the codes is free); more columns than FSG_CODES_MAX -> no codes."""
    from fetalsyngen_amd import seedcodes as SC

    for shape, variant in (((40, 36, 28), 1), ((128, 128, 128), 2)):
        parts = [p.to(DEV) for p in _phantom_parts(shape, variant)]
        got = SC.build_device(parts, 25)
        want = SC.build(parts, 25)
        assert got is not None and got[1].shape == want[1].shape
        assert (got[1][:, 24] == 0).all()
        # same partition: the pair (device code, torch code) takes exactly ntuples distinct values
        pair = got[0].reshape(-1).to(torch.int64) * 4096 + want[0].reshape(-1).to(torch.int64)
        assert torch.unique(pair).numel() == want[1].shape[0]
        rs = np.random.RandomState(7)
        for _ in range(6):
            ns = rs.randint(1, 7, 4)
            sel = [4 * (ns[m] - 1) + m for m in range(4)]
            assert torch.equal(SC.labels_of(*got, sel), SC.labels_of(*want, sel))
    # 4 096 distinct columns
    n = 1 << 16
    a = (torch.arange(n, device=DEV) % 256).to(torch.uint8)
    b = ((torch.arange(n, device=DEV) // 256) % 16).to(torch.uint8)
    assert SC.build_device([a, b], 3) is None
    # 2 048 exactly: usable
    b2 = ((torch.arange(n, device=DEV) // 256) % 8).to(torch.uint8)
    got = SC.build_device([a, b2], 3)
    assert got is not None and got[1].shape[0] == 2048
    assert torch.equal(SC.labels_of(*got, [0, 1, 2, 2]), ((a.to(torch.int64) + b2.to(torch.int64)) & 255).to(torch.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("prob", [1.0, 0.6])
def test_look_ahead_carries_the_next_samples_draw_without_changing_a_voxel(K, prob):
    """fsg_keyed_sample_run's look-ahead: with the next key named, the next sample's draw job rides in this sample's floor(min)
    launch.  Same volumes as without; a wrong announcement, no announcement, or a deformation gate that switches the carrying
    launch off must all fall back to a draw launch of the sample's own."""
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (64, 56, 72)
    kw = dict(prob=prob, nonlin_scale=(0.08, 0.2), bf_scale=(0.05, 0.2))
    subj = [make_seed_volumes(shape, v) for v in range(3)]
    segs = [torch.from_numpy(s).to(DEV) for s, _ in subj]
    keys = [sharding.sample_key(77, i) for i in range(14)]
    which = [0, 1, 1, 2, 0, 0, 1, 2, 2, 2, 0, 1, 0, 2]

    ref_gen = make_generator(shape, DEV, rng="keyed", **kw)
    ref_banks = [SeedBank(b, DEV) for _s, b in subj]
    ref = [ref_gen._pipeline(None, segs[w], ref_banks[w], {}, scale01=True, key=k)[:2] for k, w in zip(keys, which)]

    gen = make_generator(shape, DEV, rng="keyed", **kw)
    banks = [SeedBank(b, DEV) for _s, b in subj]
    carried = []
    for i, (k, w) in enumerate(zip(keys, which)):
        nxt = keys[i + 1] if i + 1 < len(keys) else None
        if i == 4:    # announce the wrong key
            nxt ^= 1
        elif i == 9:  # announce nothing
            nxt = None
        if i == 11:   # another sample between two announced ones
            gen._pipeline(None, segs[0], banks[0], {}, scale01=True, key=12345)
        out, seg_o, _img, params = gen._pipeline(None, segs[w], banks[w], {}, scale01=True, key=k, next_key=nxt)
        assert torch.equal(out, ref[i][0]) and torch.equal(seg_o, ref[i][1]), f"sample {i}"
        assert params["key"] == k
        carried.append(bool(gen.__dict__.get("_pre")))
    if prob == 1.0:  # the deformation gate is on: the job rides whenever a next key was named
        assert all(c for j, c in enumerate(carried[:-1]) if j != 9) and not carried[9], carried
    else:
        assert any(carried) and not all(c for j, c in enumerate(carried[:-1]) if j != 9), carried
