"""GPU parity: the HIP path (through the C ABI) against the oracle and the golden vectors.

Tolerances (0..255 intensity scale unless noted):
  * integer / index work (labels, nearest gather, sampling coordinates, zoom, GMM with injected
    noise): BIT-EXACT;
  * trilinear gather: bit-exact (same operation order, FMA contraction off);
  * blur / gamma (powf) / bias (expf): atol 1e-3, rtol 1e-5;
  * final [0,1] image: atol 2e-5.
"""
import numpy as np
import pytest
import torch

from oracle import fsg_oracle as O
from tests.util_cases import E2E, make_generator, t

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ATOL, RTOL = 1e-3, 1e-5


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (and libfsg_hip.so); there is no fallback to skip to")
    from fetalsyngen_amd import kernels

    return kernels


def dev(a):
    return t(a).to(DEV)


def host(x):
    return x.detach().cpu().numpy()


# ---- RNG ------------------------------------------------------------------------------------------
def test_randn_statistics_and_determinism(K):
    a = K.randn((1 << 22,), 1234, 7, DEV)
    b = K.randn((1 << 22,), 1234, 7, DEV)
    c = K.randn((1 << 22,), 1235, 7, DEV)
    d = K.randn((1 << 22,), 1234, 8, DEV)
    assert torch.equal(a, b)
    assert not torch.equal(a, c) and not torch.equal(a, d)
    x = host(a).astype(np.float64)
    assert abs(x.mean()) < 3e-3 and abs(x.var() - 1) < 5e-3
    assert abs((x**3).mean()) < 1e-2 and abs((x**4).mean() - 3) < 3e-2
    assert np.isfinite(x).all() and np.abs(x).max() < 6.0
    # prefix property: a shorter request is the head of a longer one (ragged sizes included)
    e = K.randn((1001,), 1234, 7, DEV)
    assert torch.equal(e, a[:1001])
    # lag-1 autocorrelation (consecutive elements share a Philox block)
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 3e-3


# ---- K1 -------------------------------------------------------------------------------------------
def test_gmm_injected_noise_exact(K, golden):
    g = golden("gmm")
    seeds = g["seeds"]
    for k in range(2):
        torch.manual_seed(0)
        torch.rand(50), torch.rand(50)
        if k == 0:
            torch.randn(41)
        z = torch.randn(seeds.shape)
        for lab in (dev(seeds), dev(seeds.astype(np.int64))):
            img = K.gmm_sample(lab, dev(g[f"mus_{k}"]), dev(g[f"sigmas_{k}"]), noise=z.to(DEV))
            assert np.array_equal(host(img), g[f"img_{k}"])


def test_gmm_philox_matches_exposed_noise_and_stats(K):
    n = (64, 96, 80)
    rs = np.random.RandomState(5)
    labs = np.array([0, 10, 11, 20, 30, 31, 32, 40, 49], dtype=np.uint8)
    seeds = labs[rs.randint(0, len(labs), n)]
    seeds[:, :48] = 0  # piecewise-constant region as in real label maps
    mus = (25 + 200 * rs.rand(50)).astype(np.float32)
    sig = (5 + 20 * rs.rand(50)).astype(np.float32)
    img = K.gmm_sample(dev(seeds), dev(mus), dev(sig), seed=99, stream_id=1)
    z = K.randn(n, 99, 1, DEV)
    ref = O.gmm_image(t(seeds.astype(np.int64)), t(mus), t(sig), z.cpu())
    assert np.array_equal(host(img), ref.numpy())
    # per-label statistics on the device (wave-level reductions) vs numpy
    cnt, mean, var = K.label_stats(dev(seeds), img, 50)
    cnt, mean, var = host(cnt), host(mean), host(var)
    x = host(img).astype(np.float64)
    for l in labs:
        m = seeds == l
        assert cnt[l] == m.sum()
        assert abs(mean[l] - x[m].mean()) < 1e-3 and abs(var[l] - x[m].var()) < 0.1
        # and they are what the GMM asked for (clamp at 0 is negligible for these parameters)
        if mus[l] > 4 * sig[l]:
            assert abs(mean[l] - mus[l]) < 5 * sig[l] / np.sqrt(m.sum()) + 1e-3
    assert cnt.sum() == seeds.size


def test_gmm_ragged_and_empty(K):
    mus, sig = dev(np.linspace(10, 200, 50, dtype=np.float32)), dev(np.full(50, 3, np.float32))
    for n in (1, 2, 3, 5, 63, 257):
        seeds = (np.arange(n) % 50).astype(np.uint8)
        z = torch.randn(n)
        img = K.gmm_sample(dev(seeds), mus, sig, noise=z.to(DEV))
        assert np.array_equal(host(img), O.gmm_image(t(seeds.astype(np.int64)), mus.cpu(), sig.cpu(), z).numpy())
    empty = K.gmm_sample(torch.empty(0, dtype=torch.uint8, device=DEV), mus, sig)
    assert empty.numel() == 0


# ---- zoom / interp ----------------------------------------------------------------------------------
def test_zoom_exact(K, golden):
    from fetalsyngen_amd.utils.generation import myzoom_torch

    g = golden("zoom")
    for i in range(int(g["ncases"])):
        y = myzoom_torch(dev(g[f"x_{i}"]), g[f"factor_{i}"])
        assert np.array_equal(host(y), g[f"y_{i}"]), i


def test_interp_exact(K, golden):
    from fetalsyngen_amd.utils.generation import fast_3D_interp_torch

    g = golden("interp")
    for ci in range(2):
        x = dev(g[f"x_{ci}"])
        for tag in ("raw", "clamped"):
            co = g[f"coords_{ci}_{tag}"]
            ii, jj, kk = (dev(co[a]) for a in range(3))
            assert np.array_equal(host(fast_3D_interp_torch(x, ii, jj, kk, "linear")), g[f"lin_{ci}_{tag}"])
            assert np.array_equal(host(fast_3D_interp_torch(x, ii, jj, kk, "nearest")), g[f"nn_{ci}_{tag}"])
    assert fast_3D_interp_torch(x, None, None, None, "linear") is x
    with pytest.raises(Exception, match="mode must be linear or nearest"):
        fast_3D_interp_torch(x, ii, jj, kk, "cubic")


# ---- deformation ------------------------------------------------------------------------------------
def _spec_from_golden(K, g, i, flip=False):
    from fetalsyngen_amd import tables as T

    shape, size = tuple(int(v) for v in g[f"shape_{i}"]), tuple(int(v) for v in g[f"size_{i}"])
    field, tabs = None, None
    if f"Fsmall_{i}" in g.files:
        fs = g[f"Fsmall_{i}"]
        ht, new = T.zoom_tables(fs.shape[:3], np.array(shape) / np.array(fs.shape[:3]))
        assert new == shape
        tabs, field = K.DeviceTables(ht, DEV), dev(fs)
    c2 = t(g[f"c2_{i}"]).to(torch.float32).numpy()
    return K.DeformSpec(shape, g[f"A_{i}"], (np.array(size) - 1) / 2, c2, flip, field, tabs, device=DEV), shape


def test_deform_coords_exact(K, golden):
    g = golden("deform_image")
    for i in range(int(g["ncases"])):
        spec, shape = _spec_from_golden(K, g, i)
        mm6 = K.coords_minmax(spec)
        xx, yy, zz = K.coords(spec, mm6)
        got = np.stack([host(xx), host(yy), host(zz)])
        assert np.array_equal(got, g[f"coords_{i}"]), i
        keys = host(mm6)
        lo = [int(np.floor(K.key_to_float(k))) for k in keys[:3]]
        hi = [int(1 + np.ceil(K.key_to_float(k))) for k in keys[3:]]
        assert lo + hi == list(g[f"margins_{i}"])


def test_fused_warp_equals_materialised_path(K, golden):
    """warp kernel (coordinates recomputed in-kernel, flip folded into the index) == oracle samplers on
    the golden coordinates, for fp32 and uint8 label volumes."""
    g = golden("deform_image")
    rs = np.random.RandomState(3)
    for i in range(int(g["ncases"])):
        for flip in (False, True):
            spec, shape = _spec_from_golden(K, g, i, flip)
            img = (rs.rand(*shape) * 255).astype(np.float32)
            lab = rs.randint(0, 8, shape).astype(np.uint8)
            mm6 = K.coords_minmax(spec)
            out, seg = K.warp(spec, mm6, src_lin=dev(img), src_nn=dev(lab.astype(np.float32)))
            _, seg8 = K.warp(spec, mm6, src_nn=dev(lab))
            co = [t(g[f"coords_{i}"][a]) for a in range(3)]
            r_img, r_seg = O.apply_deformation(t(img), t(lab.astype(np.float32)), co, flip)
            assert np.array_equal(host(out), r_img.numpy())
            assert np.array_equal(host(seg), r_seg.numpy())
            assert np.array_equal(host(seg8), r_seg.numpy().astype(np.uint8))


def test_warp_identity_and_flip_properties(K):
    """Identity affine, no field: interior voxels are copied (coordinate 0 planes are 'outside' by the
    strict >0 rule); flip twice restores the volume."""
    shape = (40, 36, 28)
    rs = np.random.RandomState(0)
    img = (rs.rand(*shape) * 255).astype(np.float32)
    eye = np.eye(3, dtype=np.float32)
    c = (np.array(shape) - 1) / 2
    spec = K.DeformSpec(shape, eye, c, c.astype(np.float32), False, device=DEV)
    mm6 = K.coords_minmax(spec)
    out, _ = K.warp(spec, mm6, src_lin=dev(img))
    o = host(out)
    assert np.array_equal(o[1:, 1:, 1:], img[1:, 1:, 1:])
    assert not o[0].any() and not o[:, 0].any() and not o[:, :, 0].any()
    # half a voxel along z: interior outputs are the mean of two z-neighbours; the last column's position (n2 - 0.5) is clamped
    # onto the last column exactly -- the lean kernel then reads the pair one to the LEFT with weights (0, 1) -- and must
    # return the last column itself, like the reference's clamped upper neighbour with weights (1, 0)
    c_half = c.astype(np.float32).copy()
    c_half[2] += 0.5
    spech = K.DeformSpec(shape, eye, c, c_half, False, device=DEV)
    outh, _ = K.warp(spech, K.coords_minmax(spech), src_lin=dev(img))
    oh = host(outh)
    want = img[1:, 1:, :-1] * np.float32(0.5) + img[1:, 1:, 1:] * np.float32(0.5)
    assert np.array_equal(oh[1:, 1:, :-1], want)
    assert np.array_equal(oh[1:, 1:, -1], img[1:, 1:, -1])
    specf = K.DeformSpec(shape, eye, c, c.astype(np.float32), True, device=DEV)
    _, f1 = K.warp(specf, mm6, src_nn=dev(img))
    _, f2 = K.warp(specf, mm6, src_nn=f1)
    assert np.array_equal(host(f1), img[::-1]) and np.array_equal(host(f2), img)


# ---- blur ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("generic", [False, True])
def test_blur_golden(K, golden, generic):
    from fetalsyngen_amd import tables as T

    g = golden("blur")
    for si in range(3):
        x = dev(g[f"x_{si}"])
        for ti, st in enumerate(g["stds"]):
            y = x
            for axis in range(3):
                if st[axis] > 0:
                    y = K.blur_axis(y, axis, T.gaussian_taps(float(st[axis])), force_generic=generic)
            np.testing.assert_allclose(host(y), g[f"y_{si}_{ti}"], rtol=RTOL, atol=ATOL)


def test_blur_fast_paths_match_oracle_and_properties(K):
    from fetalsyngen_amd import tables as T
    from fetalsyngen_amd.utils.generation import gaussian_blur_3d

    rs = np.random.RandomState(1)
    for shape in [(64, 64, 64), (40, 72, 128), (33, 20, 256)]:
        x = (rs.rand(*shape) * 255).astype(np.float32)
        for sigma in (0.44, 0.9, 1.3, 1.77, 2.6):
            st = [sigma] * 3
            y = gaussian_blur_3d(dev(x), st, DEV)
            np.testing.assert_allclose(host(y), O.blur3d(t(x), st).numpy(), rtol=RTOL, atol=ATOL)
            for axis in range(3):
                taps = T.gaussian_taps(sigma)
                a = K.blur_axis(dev(x), axis, taps)
                b = K.blur_axis(dev(x), axis, taps, force_generic=True)
                np.testing.assert_allclose(host(a), host(b), rtol=1e-6, atol=1e-4)
    # linearity and interior constant-preservation / border attenuation (zero padding, no renormalisation)
    x1, x2 = dev((rs.rand(64, 64, 64) * 255).astype(np.float32)), dev((rs.rand(64, 64, 64) * 255).astype(np.float32))
    taps = T.gaussian_taps(1.77)
    for axis in range(3):
        lhs = K.blur_axis(2 * x1 + x2, axis, taps)
        rhs = 2 * K.blur_axis(x1, axis, taps) + K.blur_axis(x2, axis, taps)
        np.testing.assert_allclose(host(lhs), host(rhs), rtol=1e-5, atol=1e-3)
        ones = K.blur_axis(torch.ones(64, 64, 64, device=DEV), axis, taps)
        o = np.moveaxis(host(ones), axis, 0)
        R = len(taps) // 2
        np.testing.assert_allclose(o[R:-R], 1.0, atol=1e-6)
        assert (o[0] < 0.75).all() and (o[-1] < 0.75).all()


def test_blur_resample_fused_pair_equals_unfused_and_oracle(K):
    """K6 + K7 + K8 as the fused pair of launches (csrc/fsg_blur_rs.hip: blur and down-sampling of an axis as one FIR with
    position-dependent coefficients) against (a) the unfused HIP sequence, (b) the oracle's gaussian_blur_3d +
    fast_3D_interp_torch restatement (reference utils/generation.py:84-110, :227-278; synthseg.py:78-105, :230-233).
    A re-ordered linear map in float32: the blur's tolerance (atol 1e-3 on 0..255)."""
    from fetalsyngen_amd import tables as T

    rs = np.random.RandomState(5)
    cases = [((64, 56, 72), (0.5, 0.5, 0.5), 0.74, 0.3), ((64, 56, 72), (0.5, 0.5, 0.5), 1.5, 0.9),
             ((48, 40, 64), (0.5, 0.5, 0.5), 0.55, 0.0), ((96, 96, 96), (0.5, 0.5, 0.5), 1.1, 0.5),
             ((40, 48, 56), (0.5, 0.6, 0.7), 1.3, 0.7),   # anisotropic resolution: another radius and size per axis
             ((33, 20, 128), (0.5, 0.5, 0.5), 0.9, 0.2), ((256, 256, 256), (0.5, 0.5, 0.5), 0.75, 0.4),
             ((40, 36, 28), (0.5, 0.5, 0.5), 0.8, 0.6), ((20, 24, 12), (0.5, 0.5, 0.5), 1.4, 0.1),   # rows shorter than a wave
             ((24, 300, 320), (0.5, 0.5, 0.5), 1.0, 0.5)]                                            # more than 256 voxels per row
    for shape, res, spacing, u_std in cases:
        x = (rs.rand(*shape) * 255).astype(np.float32)
        x[: shape[0] // 3] *= 0.1  # structure across the volume, not only white noise
        stds, new, _fac, tabs = T.resample_plan(shape, np.array(res), np.array([spacing] * 3), u_std)
        taps = [T.gaussian_taps(float(s_)) for s_ in stds]
        rt = K.DeviceTables(tabs, DEV)
        xd = dev(x)
        fused = K.blur_resample(xd, rt, taps)
        assert fused is not None, (shape, spacing)
        y = xd
        for axis in range(3):
            y = K.blur_axis(y, axis, taps[axis])
        unfused = K.resample_noise(y, rt)
        assert tuple(fused.shape) == tuple(new)
        np.testing.assert_allclose(host(fused), host(unfused), rtol=RTOL, atol=ATOL)
        if np.prod(shape) <= 96 ** 3:
            ref, _f = O.resample_down(t(x), res, np.array([spacing] * 3), u_std)
            np.testing.assert_allclose(host(fused), ref.numpy(), rtol=RTOL, atol=ATOL)
        # with the noise epilogue: the same Philox field (indexed by the flat output quad), clamped at 0
        a = K.blur_resample(xd, rt, taps, noise_std=40.0, seed=99, stream_id=2)
        b = K.resample_noise(y, rt, noise_std=40.0, seed=99, stream_id=2)
        np.testing.assert_allclose(host(a), host(b), rtol=RTOL, atol=ATOL)
        assert float(a.min()) == 0.0
        z = K.randn(tuple(new), 7, 2, DEV)
        a = K.blur_resample(xd, rt, taps, noise_std=11.0, noise=z)
        b = K.resample_noise(y, rt, noise_std=11.0, noise=z)
        np.testing.assert_allclose(host(a), host(b), rtol=RTOL, atol=ATOL)
    # properties at the full size: linearity, zero stays zero, a constant stays constant away from the borders
    shape = (256, 256, 256)
    stds, new, _fac, tabs = T.resample_plan(shape, np.array([0.5] * 3), np.array([1.2] * 3), 0.6)
    taps = [T.gaussian_taps(float(s_)) for s_ in stds]
    rt = K.DeviceTables(tabs, DEV)
    x1, x2 = torch.rand(shape, device=DEV) * 255, torch.rand(shape, device=DEV) * 255
    lhs = K.blur_resample(2 * x1 + x2, rt, taps)
    rhs = 2 * K.blur_resample(x1, rt, taps) + K.blur_resample(x2, rt, taps)
    np.testing.assert_allclose(host(lhs), host(rhs), rtol=1e-5, atol=2e-3)
    assert float(K.blur_resample(torch.zeros(shape, device=DEV), rt, taps).abs().max()) == 0.0
    ones = host(K.blur_resample(torch.ones(shape, device=DEV), rt, taps))
    np.testing.assert_allclose(ones[4:-4, 4:-4, 4:-4], 1.0, atol=2e-6)
    # outside the fused domain (an axis without blur: spacing <= resolution): the caller is told to use the unfused path
    stds, new, _fac, tabs = T.resample_plan((64, 64, 64), np.array([0.5] * 3), np.array([0.5] * 3), 0.5)
    assert K.blur_resample(torch.zeros(64, 64, 64, device=DEV), K.DeviceTables(tabs, DEV), [np.ones(1, np.float32)] * 3) is None


def test_zoom_tile_kernel_equals_row_kernels(K):
    """The tile kernel (x-blended source rows staged once per workgroup, four outputs and one Philox block per thread) and the
    slab kernel (x, y, z stages through LDS, four consecutive outputs per lane) against the row-per-wave kernels, for every
    epilogue, up- and down-sampling, ragged sizes: bit-identical (same x -> y -> z operation order, same Philox counter ->
    element mapping)."""
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import tables as T

    lib = _lib.load()
    rs = np.random.RandomState(4)
    for shape, m in (((64, 64, 64), 43), ((48, 40, 36), 29), ((40, 72, 128), 97), ((64, 64, 64), 64)):
        x = dev((rs.rand(*shape) * 255).astype(np.float32))
        spacing = [0.5 * n / max(int(n * m / shape[0]), 1) for n in shape]
        stds, new, fac, rtabs = T.resample_plan(shape, [0.5] * 3, spacing, 0.5)
        rt = K.DeviceTables(rtabs, DEV)
        bt, _ = T.zoom_tables(new, 1 / np.asarray(fac))
        zt = K.DeviceTables(bt, DEV)
        res = {}
        for flag in (256, 512, 8192, 32768, 0):  # FSG_TUNE_ROW_ZOOM, _TILE_ZOOM, _SLAB_ZOOM, _WAVE_ZOOM (r03, opt-in), defaults
            prev = lib.fsg_set_tuning(flag)
            try:
                for ty in ((16,) if flag in (256, 32768, 0) else (1, 5, 16, 32)):
                    lib.fsg_zoom_set_tuning(ty, 12288)
                    low = K.resample_noise(x, rt, noise_std=9.0, seed=11, stream_id=2)
                    z = K.randn(tuple(new), 5, 6, DEV)
                    low2 = K.resample_noise(x, rt, noise_std=3.0, noise=z)
                    plain = K.zoom3d(x, rt)
                    mm = K.zoom_minmax(low, zt)
                    up = [K.zoom_normalise(low, zt, mm, mode) for mode in (0, 1)]
                    # the same pair with the keys sharded over slots (what fsg_sample_run uses)
                    for nslots in (2, 32, 64):
                        slots = K.zoom_minmax_sharded(low, zt, nslots)
                        sk = host(slots)
                        assert sk[:, 0].min() == int(mm[0]) and sk[:, 1].max() == int(mm[1]) and not sk[:, 2:].any()
                        for mode in (0, 1):
                            assert torch.equal(K.zoom_normalise(low, zt, slots, mode), up[mode]), (shape, m, flag, nslots)
                    res[(flag, ty)] = [low, low2, plain, mm] + up
            finally:
                lib.fsg_set_tuning(prev)
                lib.fsg_zoom_set_tuning(16, 12288)
        ref = res[(256, 16)]
        for key, val in res.items():
            for a, b in zip(ref, val):
                assert torch.equal(a, b), (shape, m, key)


def test_uniform_division_in_k9b_is_the_ieee_quotient(K):
    """K9b divides every voxel by the same maximum through a reciprocal + two FMAs (`unidiv`, fsg_zoom.hip) instead of the
    IEEE division expansion.  Exhaustive over the significand: a 2^23-voxel volume holding EVERY float32 significand of one
    binade goes through an identity zoom + normalise (mode 0: v / max) for divisors that include powers of two, all-ones
    significands (where the shortcut is not valid and the kernel must fall back), tiny and huge values; the result must be the
    correctly rounded quotient, bit for bit (numpy float32 division on the host; torch's device division is not used as
    the reference: it is not the correctly rounded one on this build)."""
    from fetalsyngen_amd import tables as T

    shape = (128, 256, 256)
    n = int(np.prod(shape))
    assert n == 1 << 23
    tabs, new = T.zoom_tables(shape, np.array([1.0, 1.0, 1.0]))
    assert new == shape
    zt = K.DeviceTables(tabs, DEV)
    rs = np.random.RandomState(8)
    divisors = [1.0, 2.0, 0.5, 255.0, 300.0, 3.0, 1e-3, 7e7,
                float(np.float32(np.uint32(0x437FFFFF).view(np.float32))),   # 255.99998: significand all ones
                float(np.float32(np.uint32(0x3FFFFFFF).view(np.float32))),   # 1.9999999
                float(np.float32(np.uint32(0x3F800001).view(np.float32)))]   # 1 + ulp
    divisors += [float(v) for v in (30 + 370 * rs.rand(12)).astype(np.float32)]
    divisors += [float(np.uint32(b).view(np.float32)) for b in rs.randint(0x3D000000, 0x46000000, 12, dtype=np.int64).astype(np.uint32)]
    mant = np.arange(n, dtype=np.uint32)
    for exp_bits in (127, 120, 134):  # binades [1,2), [2^-7, 2^-6), [128, 256)
        vals = (mant | np.uint32(exp_bits << 23)).view(np.float32).reshape(shape)
        x = dev(vals)
        # the host's float32 division is the reference (IEEE, correctly rounded); every divisor on the first binade, a
        # subset on the others
        for d in (divisors if exp_bits == 127 else divisors[:4] + divisors[8:11] + divisors[-3:]):
            d32 = np.float32(d)
            mm = torch.tensor([0, int(d32.view(np.int32))], dtype=torch.int32, device=DEV)  # keys of (min = +0.0, max = d)
            got = host(K.zoom_normalise(x, zt, mm, 0))
            want = vals / d32
            assert np.array_equal(got, want), (exp_bits, d, int((got != want).sum()))
    # zeros, denormals and mode 1 (min-max scaling on top)
    tiny = np.concatenate([np.zeros(1000, np.float32), (np.arange(1, 5000, dtype=np.uint32)).view(np.float32),
                           rs.rand(n - 5999).astype(np.float32) * 200]).astype(np.float32).reshape(shape)
    x = dev(tiny)
    d32 = np.float32(199.77)
    mm = torch.tensor([int(np.float32(0.0).view(np.int32)), int(d32.view(np.int32))], dtype=torch.int32, device=DEV)
    assert np.array_equal(host(K.zoom_normalise(x, zt, mm, 0)), tiny / d32)
    assert np.array_equal(host(K.zoom_normalise(x, zt, mm, 1)), tiny / d32)  # min = 0: (t - 0) / 1
    # mode 1 with a non-zero minimum: the second division, by the uniform (1 - min/max), goes through the same shortcut.
    # Every significand of [1, 2) again, for minima that make the divisor ordinary, close to 1 and all-ones
    vals = (mant | np.uint32(127 << 23)).view(np.float32).reshape(shape)
    x = dev(vals)
    for mx, mn in [(1.9999999, 1.0), (2.0, 1.0), (255.0, 3.0), (1.9999999, 1e-7), (300.0, 299.0),
                   (float(np.float32(np.uint32(0x40000000).view(np.float32))), float(np.float32(np.uint32(0x33800000).view(np.float32))))]:
        mx32, mn32 = np.float32(mx), np.float32(mn)
        mm = torch.tensor([int(mn32.view(np.int32)), int(mx32.view(np.int32))], dtype=torch.int32, device=DEV)
        mnq = mn32 / mx32
        den = np.float32(1.0) - mnq
        t = vals / mx32
        want = (t - mnq) if den == np.float32(1.0) else ((t - mnq) / den)
        got = host(K.zoom_normalise(x, zt, mm, 1))
        assert np.array_equal(got, want.astype(np.float32)), (mx, mn, int((got != want).sum()))


def test_blur_yz_fused_equals_two_passes(K):
    """y + z pass in one launch (intermediate in LDS) against the two single-axis launches: bit-identical, for tile-ragged
    shapes and every radius 1..8; outside its domain (different taps per axis, nz % 4, radius > 8) it declines (None)."""
    from fetalsyngen_amd import tables as T

    rs = np.random.RandomState(6)
    for shape in [(8, 64, 64), (5, 40, 72), (3, 33, 20), (2, 100, 132), (4, 32, 256), (2, 70, 384)]:
        x = dev((rs.rand(*shape) * 255).astype(np.float32))
        for sg in (0.3, 0.44, 0.9, 1.3, 1.6, 1.77, 2.2, 2.5, 2.66):  # radii 1..8
            tp = T.gaussian_taps(sg)
            ref = K.blur_axis(K.blur_axis(x, 1, tp), 2, tp)
            got = K.blur_yz(x, tp, tp)
            assert got is not None and torch.equal(got, ref), (shape, sg)
        assert K.blur_yz(x, T.gaussian_taps(0.9), T.gaussian_taps(1.3)) is None  # different taps per axis: two passes
    x = dev((rs.rand(4, 16, 30) * 255).astype(np.float32))  # nz % 4 != 0
    assert K.blur_yz(x, T.gaussian_taps(1.0), T.gaussian_taps(1.0)) is None
    x = dev((rs.rand(4, 64, 64) * 255).astype(np.float32))  # radius 9
    assert K.blur_yz(x, T.gaussian_taps(3.0), T.gaussian_taps(1.0)) is None


def test_blur_long_kernels(K):
    """Radius 9..64 (BlurCortex draws its sigmas from a gamma distribution, augmentation/artifacts.py:104): the run-time
    radius kernels against the generic kernel and the oracle, including rows shorter than the kernel."""
    from fetalsyngen_amd import tables as T

    rs = np.random.RandomState(2)
    for shape in [(64, 48, 64), (24, 40, 128)]:
        x = (rs.rand(*shape) * 255).astype(np.float32)
        for sigma in (3.0, 4.9, 7.3, 21.0):
            taps = T.gaussian_taps(sigma)
            assert len(taps) // 2 > 8
            for axis in range(3):
                a = K.blur_axis(dev(x), axis, taps)
                b = K.blur_axis(dev(x), axis, taps, force_generic=True)
                np.testing.assert_allclose(host(a), host(b), rtol=1e-6, atol=2e-4)
        st = [4.9, 3.0, 7.3]
        y = dev(x)
        for axis in range(3):
            y = K.blur_axis(y, axis, T.gaussian_taps(st[axis]))
        np.testing.assert_allclose(host(y), O.blur3d(t(x), st).numpy(), rtol=RTOL, atol=ATOL)


# ---- augmentation stages (un-fused API) -------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("gates", ["on", "off"])
def test_stage_classes_golden(K, golden, seed, gates):
    from fetalsyngen_amd import rng
    from fetalsyngen_amd.generator.augmentation.synthseg import RandBiasField, RandGamma, RandNoise, RandResample

    g = golden("stages")
    p = 1.0 if gates == "on" else 0.0
    key = f"s{seed}_{gates}"
    np.random.seed(seed)
    torch.manual_seed(seed)
    with rng.use("reference"):
        a, pa = RandGamma(p, 0.1)(dev(g["x"]), DEV)
        b, pb = RandBiasField(p, 0.05, 0.2, 0.01, 0.3)(a, DEV)
        rs = RandResample(p, 0.5, 1.5)
        c, factors, pc = rs(b, np.array([0.5, 0.5, 0.5]), DEV)
        d, pd = RandNoise(p, 5, 15)(c, DEV)
        e = rs.resize_back(d, factors)
    for name, v in (("gamma", a), ("bias", b), ("resampled", c), ("noisy", d)):
        np.testing.assert_allclose(host(v), g[f"{key}_{name}"], rtol=RTOL, atol=ATOL, err_msg=name)
    np.testing.assert_allclose(host(e), g[f"{key}_back"], rtol=RTOL, atol=2e-5 if gates == "on" else ATOL)
    if gates == "on":
        assert pa["gamma"] == float(g[f"{key}_p_gamma"])
        assert np.array_equal(np.asarray(pb["bf_size"]), g[f"{key}_p_bf_size"])
        assert np.array_equal(np.asarray(pc["spacing"]), g[f"{key}_p_spacing"])
        assert np.array_equal(np.asarray(factors), g[f"{key}_p_factors"])
        assert pd["noise_std"] == float(g[f"{key}_p_noise_std"])
    else:
        assert pa["gamma"] is None and pb["bf_size"] is None and pc["spacing"] is None and pd["noise_std"] is None
        assert factors is None and e is d


def test_resample_aniso_golden(K, golden):
    from fetalsyngen_amd import rng
    from fetalsyngen_amd.generator.augmentation.synthseg import RandResample

    g = golden("stages")
    rs = RandResample(0.0, 0.5, 1.5)
    np.random.seed(5)
    torch.manual_seed(5)
    with rng.use("reference"):
        c, factors, pc = rs(dev(g["x"]), np.array([0.5, 0.5, 0.5]), DEV, genparams={"spacing": [0.5, 0.8, 1.3]})
    np.testing.assert_allclose(host(c), g["aniso_resampled"], rtol=RTOL, atol=ATOL)
    assert np.array_equal(factors, g["aniso_factors"])
    np.testing.assert_allclose(host(rs.resize_back(c, factors)), g["aniso_back"], rtol=RTOL, atol=2e-5)


# ---- end to end --------------------------------------------------------------------------------------
def _run_hip(name, g, api="sample"):
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = tuple(int(v) for v in g["shape"])
    gen = make_generator(shape, DEV, rng="reference", **E2E[name])
    seg, seeds = make_seed_volumes(shape, int(g["variant"]))
    np.random.seed(int(g["seed"]))
    torch.manual_seed(int(g["seed"]))
    if api == "sample":
        return gen.sample(image=None, segmentation=t(seg), seeds=seeds)
    if api == "scaled":
        return gen._pipeline(None, t(seg), seeds, {}, scale01=True)
    out, seg_d, img, p1 = gen.generate(image=None, segmentation=t(seg), seeds=seeds)
    out, p2 = gen.augment(image=out, segmentation=seg_d)
    return out, seg_d, img, {**p1, **p2}


@pytest.mark.parametrize("api", ["sample", "scaled", "stagewise"])
@pytest.mark.parametrize("name", list(E2E))
def test_e2e_same_seed_as_reference(K, golden, name, api):
    """Same numpy+torch seed as the reference run that produced the golden file (host-tape RNG mode)."""
    g = golden(name)
    out, seg, _img, params = _run_hip(name, g, api)
    assert np.array_equal(host(params["seed_intensities"]["mus"]), g["mus"])
    assert np.array_equal(host(params["seed_intensities"]["sigmas"]), g["sigmas"])
    assert bool(params["deform_params"]["flip"]) == bool(g["flip"])
    m2s = params["selected_seeds"]["mlabel2subclusters"]
    assert [m2s[m] for m in range(1, 5)] == list(g["mlabel2subclusters"])
    assert np.array_equal(host(seg).astype(np.uint8), g["seg_out"]), "labels must be bit-exact"
    if params["deform_params"]["affine"] is not None:
        assert np.array_equal(params["deform_params"]["affine"]["rotations"], g["rotations"])
        assert list(params["deform_params"]["non_rigid"]["size_F_small"]) == list(g["size_F_small"])
    ns = params["noise_params"]["noise_std"]
    assert (ns is None and np.isnan(g["noise_std"])) or ns == float(g["noise_std"])
    if api == "scaled":
        np.testing.assert_allclose(host(out), g["scaled"], rtol=0, atol=2e-5)
    else:
        # y/max when the resampler fired (values in [0,1]); 0..255-scale intensities otherwise
        atol = 2e-5 if not np.isnan(g["spacing"]).any() else ATOL
        np.testing.assert_allclose(host(out), g["out"], rtol=RTOL, atol=atol)


def test_e2e_device_rng_fullsize_vs_oracle(K):
    """BASELINE config 2 shape (256^3), device-Philox RNG: the oracle is fed the very noise fields the
    kernels generate (fsg_randn_f32 with the keys the host drew) and must agree."""
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (256, 256, 256)
    seg, seeds = make_seed_volumes(shape, 0)
    gen = make_generator(shape, DEV, rng="device")
    np.random.seed(7)
    torch.manual_seed(7)
    out, seg_d, _img, params = gen._pipeline(None, t(seg), seeds, {}, scale01=True)
    torch.cuda.synchronize()

    streams = iter((1, 2))

    def philox_field(shp):
        key = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())  # same draw the product makes
        return K.randn(shp, key, next(streams), DEV).cpu()

    cfg = O.Config(shape, prob=1.0)
    np.random.seed(7)
    torch.manual_seed(7)
    r = O.run_sample(cfg, t(seg), seeds, noise_gmm=philox_field, noise_lowres=philox_field)
    assert np.array_equal(host(params["seed_intensities"]["mus"]), r["params"]["mus"].numpy())
    assert np.array_equal(host(seg_d).astype(np.uint8), r["seg"].numpy().astype(np.uint8)), "labels bit-exact at 256^3"
    np.testing.assert_allclose(host(out), r["scaled"].numpy(), rtol=0, atol=2e-5)
    o = host(out)
    assert o.min() == 0.0 and o.max() == 1.0


def test_e2e_reference_rng_fullsize_vs_oracle(K):
    """BASELINE config 2 as stated: a single 256^3 volume, full path, SEED-MATCHED against the CPU path -- `rng="reference"`
    (both large noise fields drawn by torch's CPU generator at the reference's points of the draw order and uploaded), the
    oracle under the same numpy / torch seeds: labels bit-exact, [0,1] image within 2e-5, both generators left at the same
    position."""
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (256, 256, 256)
    seg, seeds = make_seed_volumes(shape, 1)
    gen = make_generator(shape, DEV, rng="reference")
    np.random.seed(11)
    torch.manual_seed(11)
    out, seg_d, _img, params = gen._pipeline(None, t(seg), seeds, {}, scale01=True)
    torch.cuda.synchronize()
    tail_product = (np.random.rand(), float(torch.rand(1)))
    np.random.seed(11)
    torch.manual_seed(11)
    r = O.run_sample(O.Config(shape, prob=1.0), t(seg), seeds)
    tail_oracle = (np.random.rand(), float(torch.rand(1)))
    assert tail_product == tail_oracle
    assert np.array_equal(host(params["seed_intensities"]["mus"]), r["params"]["mus"].numpy())
    assert np.array_equal(host(seg_d).astype(np.uint8), r["seg"].numpy().astype(np.uint8)), "labels bit-exact at 256^3"
    np.testing.assert_allclose(host(out), r["scaled"].numpy(), rtol=0, atol=2e-5)


def test_dataset_contract_and_determinism(K, tmp_path):
    """FetalSynthDataset on a tiny BIDS tree written by the test: output dict contract, CPU tensors,
    int64 labels, params schema; same seeds => identical sample; cache on/off agree."""
    from tests.util_bids import write_tree
    from fetalsyngen_amd.data.datasets import FetalSynthDataset

    shape = (32, 32, 32)
    bids, seed_dir = write_tree(tmp_path, shape, ["sub-a", "sub-b"])
    outs = []
    for cache in (True, False):
        gen = make_generator(shape, DEV, rng="reference", prob=0.9)
        ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None, cache_on_device=cache)
        assert len(ds) == 2
        np.random.seed(11)
        torch.manual_seed(11)
        d = ds.sample_with_meta(1)
        outs.append(d)
        assert d["image"].shape == (1, *shape) and d["image"].dtype == torch.float32 and not d["image"].is_cuda
        assert d["label"].shape == (1, *shape) and d["label"].dtype == torch.int64 and not d["label"].is_cuda
        assert d["name"] == "sub-b"
        assert float(d["image"].min()) == 0.0 and float(d["image"].max()) == 1.0
        for k in ("idx", "img_paths", "segm_paths", "seeds", "selected_seeds", "seed_intensities", "deform_params",
                  "gamma_params", "bf_params", "resample_params", "noise_params", "artifacts", "generation_time"):
            assert k in d["generation_params"], k
    assert torch.equal(outs[0]["image"], outs[1]["image"]) and torch.equal(outs[0]["label"], outs[1]["label"])
    item = ds[0]
    assert set(item) == {"image", "label", "name"} and hasattr(ds, "generation_params")
    with pytest.raises(FileNotFoundError):
        FetalSynthDataset(str(bids), gen, str(tmp_path / "nope"), None)


def test_errors(K):
    gen = make_generator((16, 16, 16), DEV)
    with pytest.raises(ValueError, match="intensity prior"):
        gen.sample(image=None, segmentation=torch.zeros(16, 16, 16), seeds=None)
    with pytest.raises(RuntimeError, match="no CPU"):
        K.gamma(torch.zeros(4, 4, 4), 1.0)
    with pytest.raises(RuntimeError, match="MI355X only"):
        make_generator((16, 16, 16), "cpu")


# ---- tuned kernels vs plain kernels (fsg_set_tuning) --------------------------------------------------
def test_rowwise_kernels_equal_per_voxel_kernels(K, golden):
    """Row-wise LDS kernels (warp, coordinate min/max, zoom family) must reproduce the per-voxel kernels
    bit for bit; the fast gamma/bias epilogue (v_log/v_exp) stays within the stage tolerance."""
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import tables as T

    lib = _lib.load()
    rs = np.random.RandomState(8)
    shape = (72, 60, 100)
    fs = (rs.randn(5, 4, 7, 3) * 2).astype(np.float32)
    ht, new = T.zoom_tables(fs.shape[:3], np.array(shape) / np.array(fs.shape[:3]))
    assert new == shape
    A = np.array([[1.02, 0.05, -0.03], [-0.04, 0.97, 0.06], [0.02, -0.05, 1.04]], dtype=np.float32)
    c = (np.array(shape) - 1) / 2
    img = (rs.rand(*shape) * 255).astype(np.float32)
    lab = rs.randint(0, 8, shape).astype(np.float32)
    bias = (rs.randn(3, 2, 4) * 0.3).astype(np.float32)
    bt, _ = T.zoom_tables(bias.shape, np.array(shape) / np.array(bias.shape))
    res = {}
    for flags in (0, 1, 2, 3):
        prev = lib.fsg_set_tuning(flags)
        try:
            for flip in (False, True):
                spec = K.DeformSpec(shape, A, c, c.astype(np.float32) + np.float32(0.4), flip, dev(fs),
                                    K.DeviceTables(ht, DEV), device=DEV)
                mm6 = K.coords_minmax(spec)
                out, seg = K.warp(spec, mm6, src_lin=dev(img), src_nn=dev(lab), gamma=1.13, bias=dev(bias),
                                  bias_tabs=K.DeviceTables(bt, DEV))
                plain, _ = K.warp(spec, mm6, src_lin=dev(img))
                res[(flags, flip)] = (host(mm6), host(out), host(seg), host(plain))
                if flags in (0, 2):  # the same again through the precomputed row workspace
                    spec.prepare_rows(dev(bias), K.DeviceTables(bt, DEV))
                    mm6w = K.coords_minmax(spec)
                    outw, segw = K.warp(spec, mm6w, src_lin=dev(img), src_nn=dev(lab), gamma=1.13, bias=dev(bias),
                                        bias_tabs=K.DeviceTables(bt, DEV))
                    plainw, _ = K.warp(spec, mm6w, src_lin=dev(img))
                    _, seg8w = K.warp(spec, mm6w, src_nn=dev(lab.astype(np.uint8)))
                    assert np.array_equal(host(mm6w), host(mm6)) and np.array_equal(host(outw), host(out))
                    assert np.array_equal(host(segw), host(seg)) and np.array_equal(host(plainw), host(plain))
                    assert np.array_equal(host(seg8w), host(seg).astype(np.uint8))
        finally:
            lib.fsg_set_tuning(prev)
    for flip in (False, True):
        base = res[(3, flip)]  # per-voxel kernels, OCML math
        for flags in (0, 1, 2):
            mm, out, seg, plain = res[(flags, flip)]
            assert np.array_equal(mm, base[0]) and np.array_equal(seg, base[2]) and np.array_equal(plain, base[3])
            np.testing.assert_allclose(out, base[1], rtol=RTOL, atol=ATOL)
        assert np.array_equal(res[(2, flip)][1], base[1])  # rows + precise math == per-voxel + precise math
    # zoom family
    src = (rs.rand(37, 29, 41) * 255).astype(np.float32)
    tabs, _ = T.zoom_tables(src.shape, np.array((64, 80, 96)) / np.array(src.shape))
    outs = []
    for flags in (0, 4, 8):
        prev = lib.fsg_set_tuning(flags)
        try:
            dt = K.DeviceTables(tabs, DEV)
            z = K.zoom3d(dev(src), dt)
            mm = K.zoom_minmax(dev(src), dt)
            n0 = K.zoom_normalise(dev(src), dt, mm, 0)
            n1 = K.zoom_normalise(dev(src), dt, mm, 1)
            zn = K.resample_noise(dev(src), dt, noise_std=7.5, seed=42, stream_id=2)
            outs.append([host(v) for v in (z, mm, n0, n1, zn)])
        finally:
            lib.fsg_set_tuning(prev)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
    assert np.array_equal(outs[0][0], O.linear_zoom(t(src), np.array((64, 80, 96)) / np.array(src.shape)).numpy())


def test_floormin_shortcut_has_the_exact_floor(K, golden):
    """fsg_coords_floormin_f32 (faces first, conditional full pass) vs the golden margins, including the
    engineered case whose minima are far from 0 (the full pass must then run and be exact)."""
    g = golden("deform_image")
    for i in range(int(g["ncases"])):
        spec, shape = _spec_from_golden(K, g, i)
        for ws in (False, True):
            if ws:
                spec.prepare_rows()
            mm3 = K.coords_floormin(spec)
            lo = [int(np.floor(K.key_to_float(k))) for k in host(mm3)[:3]]
            assert lo == list(g[f"margins_{i}"][:3]), (i, ws)
    assert list(g["margins_1"][:3]) != [0, 0, 0]
    # and the warp fed by it equals the warp fed by the exact min/max
    spec, shape = _spec_from_golden(K, g, 1)
    img = dev((np.random.RandomState(2).rand(*shape) * 255).astype(np.float32))
    a, _ = K.warp(spec, K.coords_floormin(spec), src_lin=img)
    b, _ = K.warp(spec, K.coords_minmax(spec), src_lin=img)
    assert torch.equal(a, b)


def test_gmm_from_label_parts_equals_gmm_of_the_sum(K):
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import combined_seed_labels, make_seed_volumes

    for shape in [(32, 32, 32), (17, 9, 7)]:  # second: size not a multiple of 4 (ragged tail)
        seg, seeds = make_seed_volumes(shape)
        bank = SeedBank(seeds, DEV)
        m2s = {1: 3, 2: 6, 3: 1, 4: 4}
        mus = dev(np.linspace(20, 220, 50, dtype=np.float32))
        sig = dev(np.linspace(5, 25, 50, dtype=np.float32))
        a = K.gmm_sample_parts(bank.parts(m2s), mus, sig, seed=5, stream_id=1)
        comb = dev(combined_seed_labels(seeds, m2s))
        b = K.gmm_sample(comb, mus, sig, seed=5, stream_id=1)
        assert torch.equal(a, b)
        assert torch.equal(bank.combined(m2s), comb)
        z = torch.randn(shape).to(DEV)
        assert torch.equal(K.gmm_sample_parts(bank.parts(m2s), mus, sig, noise=z), K.gmm_sample(comb, mus, sig, noise=z))


@pytest.mark.parametrize("case", ["typical", "extreme", "fine_grid", "no_field"])
def test_brick_kernel_equals_row_kernel(K, case):
    """LDS-brick warp (uint8 labels) vs the row kernel: bit-identical image and labels, with/without flip,
    ragged brick counts, boxes that do not fit LDS (extreme) and coarse windows wider than 4 nodes."""
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import tables as T
    from fetalsyngen_amd.utils.generation import make_affine_matrix

    lib = _lib.load()
    rs = np.random.RandomState(11)
    shape = {"typical": (70, 60, 100), "extreme": (64, 64, 64), "fine_grid": (40, 40, 40), "no_field": (33, 17, 48)}[case]
    rot = {"typical": 0.15, "extreme": 0.7, "fine_grid": 0.1, "no_field": 0.2}[case]
    fsh = {"typical": (5, 4, 7), "extreme": (4, 4, 4), "fine_grid": (20, 20, 20), "no_field": None}[case]
    A = make_affine_matrix([rot, -rot, rot], [0.01, -0.02, 0.015], [1.05, 0.95, 1.02]).astype(np.float32)
    c = (np.array(shape) - 1) / 2
    img = (rs.rand(*shape) * 255).astype(np.float32)
    lab = rs.randint(0, 8, shape).astype(np.uint8)
    bias = (rs.randn(3, 2, 4) * 0.3).astype(np.float32)
    bt = K.DeviceTables(T.zoom_tables(bias.shape, np.array(shape) / np.array(bias.shape))[0], DEV)
    for flip in (False, True):
        fs, ft = None, None
        if fsh is not None:
            fs = dev((rs.randn(*fsh, 3) * (6.0 if case == "extreme" else 2.0)).astype(np.float32))
            ft = K.DeviceTables(T.zoom_tables(fsh, np.array(shape) / np.array(fsh))[0], DEV)
        spec = K.DeformSpec(shape, A, c, c.astype(np.float32) + np.float32(0.3), flip, fs, ft, device=DEV)
        mm = K.coords_floormin(spec)
        outs = {}
        for flags in (16, 0):
            prev = lib.fsg_set_tuning(flags)
            try:
                a, l8 = K.warp(spec, mm, src_lin=dev(img), src_nn=dev(lab), gamma=0.9, bias=dev(bias), bias_tabs=bt)
                b, lf = K.warp(spec, mm, src_lin=dev(img), src_nn=dev(lab), nn_out=torch.float32)
                _, l8o = K.warp(spec, mm, src_nn=dev(lab))
                p, _ = K.warp(spec, mm, src_lin=dev(img))
                outs[flags] = [host(v) for v in (a, l8, b, lf, l8o, p)]
            finally:
                lib.fsg_set_tuning(prev)
        for u, v in zip(outs[16], outs[0]):  # brick kernel (16) vs default kernels
            assert np.array_equal(u, v), (case, flip)
        assert np.array_equal(outs[0][1], outs[0][3].astype(np.uint8)) and np.array_equal(outs[0][1], outs[0][4])
        # and against the float32-label kernels: 16-wave patch kernel (default), row kernel (flag 32), and with
        # the precomputed row workspace
        f32 = {}
        for flags in (0, 32):
            prev = lib.fsg_set_tuning(flags)
            try:
                a, lf32 = K.warp(spec, mm, src_lin=dev(img), src_nn=dev(lab.astype(np.float32)), gamma=0.9,
                                 bias=dev(bias), bias_tabs=bt)
                f32[flags] = (host(a), host(lf32))
            finally:
                lib.fsg_set_tuning(prev)
        spec.prepare_rows(dev(bias), bt)
        a, lf32 = K.warp(spec, mm, src_lin=dev(img), src_nn=dev(lab.astype(np.float32)), gamma=0.9, bias=dev(bias),
                         bias_tabs=bt)
        for got in (f32[0], f32[32], (host(a), host(lf32))):
            assert np.array_equal(got[0], outs[0][0]) and np.array_equal(got[1], outs[0][3]), (case, flip)


# ---- the other entry modes of FetalSynthGen.sample ----------------------------------------------------
@pytest.mark.parametrize("shape", [(50, 37, 70), (64, 64, 64), (33, 72, 130)])
def test_warp_work_shapes_are_bit_identical(K, shape):
    """fsg_warp_set_variant only changes which voxels a wave processes together: every variant must reproduce the
    default kernel bit for bit (image with gamma + bias epilogue, f32 and u8 labels, both flips, ragged shapes)."""
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import tables as T
    from fetalsyngen_amd.utils.generation import make_affine_matrix

    lib = _lib.load()
    rs = np.random.RandomState(11)
    img = (rs.rand(*shape) * 255).astype(np.float32)
    lab = rs.randint(0, 8, shape).astype(np.float32)
    A = make_affine_matrix([0.2, -0.17, 0.22], [0.01, -0.02, 0.015], [1.05, 0.95, 1.02]).astype(np.float32)
    fdim = (5, 4, 6)
    fs = dev((rs.randn(*fdim, 3) * 2.0).astype(np.float32))
    ft, _ = T.zoom_tables(fdim, np.array(shape) / np.array(fdim))
    bdim = (2, 3, 2)
    bias = dev((rs.randn(*bdim) * 0.2).astype(np.float32))
    bt, _ = T.zoom_tables(bdim, np.array(shape) / np.array(bdim))
    c = (np.array(shape) - 1) / 2
    res = {}
    prev = lib.fsg_warp_set_variant(0)
    try:
        for variant in (0, 1, 2, 3, 4):
            lib.fsg_warp_set_variant(variant)
            for flip in (False, True):
                spec = K.DeformSpec(shape, A, c, c.astype(np.float32), flip, fs, K.DeviceTables(ft, DEV), device=DEV)
                spec.prepare_rows(bias, K.DeviceTables(bt, DEV))
                mm6 = K.coords_minmax(spec)
                out, seg = K.warp(spec, mm6, src_lin=dev(img), src_nn=dev(lab), gamma=1.13, bias=bias,
                                  bias_tabs=K.DeviceTables(bt, DEV))
                _, seg8 = K.warp(spec, mm6, src_nn=dev(lab.astype(np.uint8)))
                plain, _ = K.warp(spec, mm6, src_lin=dev(img))
                out2, seg8f = K.warp(spec, mm6, src_lin=dev(img), src_nn=dev(lab.astype(np.uint8)), gamma=1.13, bias=bias,
                                     bias_tabs=K.DeviceTables(bt, DEV), nn_out=torch.float32)
                res[(variant, flip)] = (host(out), host(seg), host(seg8), host(plain), host(out2), host(seg8f))
        # variant 0 = the lean body (fsg_warp_lean.hip); the same with the r01 patch body and precise math
        lib.fsg_warp_set_variant(0)
        for key, flags in (("patch", 4096), ("lean_precise", 2), ("patch_precise", 4096 | 2)):
            pf = lib.fsg_set_tuning(flags)
            try:
                for flip in (False, True):
                    spec = K.DeformSpec(shape, A, c, c.astype(np.float32), flip, fs, K.DeviceTables(ft, DEV), device=DEV)
                    spec.prepare_rows(bias, K.DeviceTables(bt, DEV))
                    mm6 = K.coords_minmax(spec)
                    out, seg = K.warp(spec, mm6, src_lin=dev(img), src_nn=dev(lab), gamma=1.13, bias=bias,
                                      bias_tabs=K.DeviceTables(bt, DEV))
                    res[(key, flip)] = (host(out), host(seg))
            finally:
                lib.fsg_set_tuning(pf)
    finally:
        lib.fsg_warp_set_variant(prev)
    for flip in (False, True):
        base = res[(0, flip)]
        assert np.array_equal(base[1].astype(np.uint8), base[2]) and np.array_equal(base[1], base[5])
        assert np.array_equal(base[0], base[4])
        for variant in (1, 2, 3, 4):
            for a, b in zip(res[(variant, flip)], base):
                assert np.array_equal(a, b), (variant, flip)
        assert np.array_equal(res[("patch", flip)][0], base[0]) and np.array_equal(res[("patch", flip)][1], base[1])
        assert np.array_equal(res[("lean_precise", flip)][0], res[("patch_precise", flip)][0])
        assert np.array_equal(res[("lean_precise", flip)][1], base[1])


def _phantom_image(shape):
    g = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    return (100 * np.exp(-(g[0] ** 2 + 1.5 * g[1] ** 2 + 2 * g[2] ** 2)) + 20 * np.sin(5 * g[0] * g[1]) + 30).astype(np.float32)


def test_image_is_deformed_with_the_same_field(K):
    """load_image=True path (model.py:143-153): the real image is returned warped by the same field."""
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (40, 36, 28)
    seg, seeds = make_seed_volumes(shape, 2)
    img = _phantom_image(shape)
    gen = make_generator(shape, DEV, rng="reference", nonlin_scale=(0.1, 0.2))
    np.random.seed(21)
    torch.manual_seed(21)
    out, seg_d, img_d, params = gen.sample(image=t(img), segmentation=t(seg), seeds=seeds)
    np.random.seed(21)
    torch.manual_seed(21)
    r = O.run_sample(O.Config(shape, prob=1.0, nonlin_scale=(0.1, 0.2)), t(seg), seeds, image=t(img))
    assert np.array_equal(host(img_d), r["image"].numpy())
    assert np.array_equal(host(seg_d), r["seg"].numpy())
    np.testing.assert_allclose(host(out), r["out"].numpy(), rtol=RTOL, atol=2e-5)
    # stage-wise API gives the same
    np.random.seed(21)
    torch.manual_seed(21)
    o2, s2, i2, _ = gen.generate(image=t(img), segmentation=t(seg), seeds=seeds)
    assert np.array_equal(host(i2), r["image"].numpy()) and np.array_equal(host(s2), r["seg"].numpy())


def test_image_as_intensity_prior(K):
    """seeds=None path (model.py:131-140): the image, scaled to 0..255, replaces the GMM draw."""
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (32, 32, 32)
    seg, _ = make_seed_volumes(shape)
    img = _phantom_image(shape)
    gen = make_generator(shape, DEV, rng="reference", nonlin_scale=(0.1, 0.3))
    for api in ("sample", "stagewise"):
        np.random.seed(4)
        torch.manual_seed(4)
        if api == "sample":
            out, seg_d, img_d, params = gen.sample(image=t(img), segmentation=t(seg), seeds=None)
        else:
            o, seg_d, img_d, p1 = gen.generate(image=t(img), segmentation=t(seg), seeds=None)
            out, p2 = gen.augment(image=o, segmentation=seg_d)
            params = {**p1, **p2}
        assert params["selected_seeds"] == {} and params["seed_intensities"] == {}
        np.random.seed(4)
        torch.manual_seed(4)
        r = O.run_sample(O.Config(shape, prob=1.0, nonlin_scale=(0.1, 0.3)), t(seg), None, image=t(img))
        assert np.array_equal(host(seg_d), r["seg"].numpy())
        np.testing.assert_allclose(host(out), r["out"].numpy(), rtol=RTOL, atol=2e-5)
        np.testing.assert_allclose(host(img_d), r["image"].numpy(), rtol=0, atol=1e-4)


def test_genparams_replay_fixes_the_augmentation_strengths(K):
    """A returned `synth_params` dict passed back as `genparams` forces every gate and reproduces every
    recorded parameter (docs/datasets.md of the reference: strengths are fixed, voxel noise is re-drawn)."""
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (32, 32, 32)
    seg, seeds = make_seed_volumes(shape)
    gen = make_generator(shape, DEV, rng="device", nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
    np.random.seed(8)
    torch.manual_seed(8)
    out1, seg1, _, p1 = gen.sample(image=None, segmentation=t(seg), seeds=seeds)
    lazy = make_generator(shape, DEV, rng="device", prob=0.0, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
    np.random.seed(99)
    torch.manual_seed(99)
    out2, seg2, _, p2 = lazy.sample(image=None, segmentation=t(seg), seeds=seeds, genparams=p1)
    assert p2["selected_seeds"] == p1["selected_seeds"]
    # sigmas are taken as given; the class-tied means are re-perturbed on top of the given ones, exactly as
    # the reference does (rand_gmm.py:139-145 runs even when "mus" is supplied); untied entries stay
    assert torch.equal(p2["seed_intensities"]["sigmas"], p1["seed_intensities"]["sigmas"])
    assert torch.equal(p2["seed_intensities"]["mus"][1:10], p1["seed_intensities"]["mus"][1:10])
    for k in ("rotations", "shears", "scalings"):
        assert np.array_equal(p2["deform_params"]["affine"][k], p1["deform_params"]["affine"][k])
    assert p2["deform_params"]["flip"] == p1["deform_params"]["flip"]
    assert p2["deform_params"]["non_rigid"]["size_F_small"] == p1["deform_params"]["non_rigid"]["size_F_small"]
    assert p2["gamma_params"] == p1["gamma_params"] and p2["noise_params"] == p1["noise_params"]
    assert p2["resample_params"] == p1["resample_params"] and p2["bf_params"]["bf_size"] == p1["bf_params"]["bf_size"]
    assert out2.shape == out1.shape and not torch.equal(out2, out1)  # fields are re-drawn, as in the reference
    # with no genparams and prob 0 nothing fires
    np.random.seed(99)
    torch.manual_seed(99)
    out3, seg3, _, p3 = lazy.sample(image=None, segmentation=t(seg), seeds=seeds)
    assert p3["deform_params"] == {"affine": None, "non_rigid": None, "flip": False}
    assert p3["gamma_params"]["gamma"] is None and p3["resample_params"]["spacing"] is None
    assert np.array_equal(host(seg3), seg)


@pytest.mark.parametrize("rng_mode", ["reference", "device"])
@pytest.mark.parametrize("prob", [1.0, 0.5, 0.0])
def test_native_pipeline_equals_stagewise(K, rng_mode, prob):
    """fsg_sample_run (one native call per sample, device-resident SeedBank) must give bit-identical images
    and labels to launching the entry points one by one from Python, for every gate combination."""
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (48, 40, 32)
    seg, seeds = make_seed_volumes(shape, 1)
    bank = SeedBank(seeds, DEV)
    segd = dev(seg)
    for seed in range(6):
        res = {}
        for native in (True, False):
            gen = make_generator(shape, DEV, rng=rng_mode, prob=prob, nonlin_scale=(0.1, 0.25), bf_scale=(0.05, 0.15))
            gen.native_pipeline = native
            for scale01 in (True, False):
                np.random.seed(seed)
                torch.manual_seed(seed)
                out, lab, img, params = gen._pipeline(None, segd, bank, {}, scale01=scale01)
                res[(native, scale01)] = (host(out), host(lab), params)
        for scale01 in (True, False):
            a, b = res[(True, scale01)], res[(False, scale01)]
            assert np.array_equal(a[0], b[0]), (seed, scale01)
            assert np.array_equal(a[1], b[1]), (seed, scale01)
            assert a[2]["gamma_params"] == b[2]["gamma_params"] and a[2]["noise_params"] == b[2]["noise_params"]
            assert a[2]["resample_params"] == b[2]["resample_params"]
    # and the host-array seeds path (reference-style dict of volumes) agrees with the SeedBank path
    gen = make_generator(shape, DEV, rng=rng_mode, prob=prob, nonlin_scale=(0.1, 0.25), bf_scale=(0.05, 0.15))
    np.random.seed(3)
    torch.manual_seed(3)
    o1, l1, _, _ = gen._pipeline(None, t(seg), seeds, {}, scale01=True)
    np.random.seed(3)
    torch.manual_seed(3)
    o2, l2, _, _ = gen._pipeline(None, segd, bank, {}, scale01=True)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)


def test_sample_head_entry_equals_its_three_parts(K, golden):
    """fsg_sample_head_f32 through the C ABI against fsg_gmm_sample_u8x4 + fsg_deform_rows_f32 + fsg_coords_floormin_f32:
    image and rows bit-identical; keys have the golden floors once fsg_coords_floormin_rest_f32 has run (including the
    engineered golden case whose minima are far from 0, where that pass is the one that settles them)."""
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes
    from fetalsyngen_amd import tables as T

    g = golden("deform_image")
    rs = np.random.RandomState(8)
    for i in range(int(g["ncases"])):
        spec, shape = _spec_from_golden(K, g, i)
        if int(spec.c.field_dims[2]) == 0:
            continue  # the head needs at least one row entry
        seg, seeds = make_seed_volumes(shape, 3)
        parts = SeedBank(seeds, DEV).parts({1: 1, 2: 1, 3: 1, 4: 1})
        mus = dev((25 + 200 * rs.rand(50)).astype(np.float32))
        sig = dev((5 + 20 * rs.rand(50)).astype(np.float32))
        bsz = (2, 3, 2)
        bias = dev(rs.randn(*bsz).astype(np.float32))
        bt, _ = T.zoom_tables_between(bsz, shape)
        btabs = K.DeviceTables(bt, DEV)
        img, mm3 = K.sample_head(parts, mus, sig, spec, bias, btabs, seed=5, stream_id=1)
        rows_head = spec._keep[-1].clone()
        K.coords_floormin_rest(spec, mm3)
        assert [int(np.floor(K.key_to_float(k))) for k in host(mm3)[:3]] == list(g[f"margins_{i}"][:3]), i
        ref_img = K.gmm_sample_parts(parts, mus, sig, seed=5, stream_id=1)
        spec2, _ = _spec_from_golden(K, g, i)
        spec2.prepare_rows(bias, btabs)
        need = 3 * int(spec.c.field_dims[2]) + bsz[2]
        a = rows_head.view(shape[0] * shape[1], -1)[:, :need]
        b = spec2._keep[-1].view(shape[0] * shape[1], -1)[:, :need]
        assert torch.equal(img, ref_img) and torch.equal(a, b), i
        mm_ref = K.coords_floormin(spec2)
        assert [int(np.floor(K.key_to_float(k))) for k in host(mm_ref)[:3]] == list(g[f"margins_{i}"][:3]), i


def test_fused_sample_head_equals_split_launches(K):
    """The head of a sample as one launch (GMM draw + per-row coarse values + six-face minimum, keys uploaded initialised)
    against the same three jobs as separate launches with a launch-side key reset (FSG_TUNE_SPLIT_HEAD): images, labels
    bit-identical, for cubic / ragged shapes, both RNG modes, with and without the bias field."""
    from fetalsyngen_amd import _lib
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    lib = _lib.load()
    for shape in ((48, 40, 32), (33, 50, 44), (64, 64, 64)):
        seg, seeds = make_seed_volumes(shape, 2)
        bank, segd = SeedBank(seeds, DEV), dev(seg)
        for rng_mode in ("device", "reference"):
            gen = make_generator(shape, DEV, rng=rng_mode, prob=0.7, nonlin_scale=(0.1, 0.25), bf_scale=(0.05, 0.15))
            for seed in range(5):
                res = []
                for flag in (0, 2048):
                    prev = lib.fsg_set_tuning(flag)
                    try:
                        np.random.seed(seed)
                        torch.manual_seed(seed)
                        out, lab, _, _ = gen._pipeline(None, segd, bank, {}, scale01=True)
                        res.append((out.clone(), lab.clone()))
                    finally:
                        lib.fsg_set_tuning(prev)
                assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (shape, rng_mode, seed)


# ---- BASELINE.json configs as parity cases ------------------------------------------------------------
def test_config1_sta21_128_on_gpu(K, golden):
    """BASELINE config 1 (sub-sta21 at 128^3, seed 0, YAML probabilities): the HIP path against what the real
    reference produced on the CPU (fixture holds the decimated inputs and summaries of the output)."""
    g = golden("config1_sta21_128")
    comb = g["seeds_in"].astype(np.int8)
    zero = np.zeros_like(comb)
    seeds = {n: {m: (comb if m == 1 else zero) for m in range(1, 5)} for n in range(1, 7)}
    shape = (128, 128, 128)
    for bank in (False, True):
        gen = make_generator(shape, DEV, rng="reference", prob=0.9, resolution=(1.0, 1.0, 1.0), res_range=(1.0, 3.0))
        src = seeds
        if bank:
            from fetalsyngen_amd.data.datasets import SeedBank

            src = SeedBank(seeds, DEV)
        np.random.seed(0)
        torch.manual_seed(0)
        out, seg, _img, params = gen._pipeline(None, dev(g["seg_in"].astype(np.float32)), src, {}, scale01=True)
        m2s = params["selected_seeds"]["mlabel2subclusters"]
        assert [m2s[m] for m in range(1, 5)] == list(g["mlabel2subclusters"])
        so = host(seg).astype(np.uint8)
        assert np.array_equal(np.bincount(so.reshape(-1), minlength=8), g["label_counts"])
        assert np.array_equal(so[::4, ::4, ::4], g["seg_sub4"])
        assert np.array_equal(so[64], g["seg_slice_x"]) and np.array_equal(so[:, :, 64], g["seg_slice_z"])
        sc = host(out)
        np.testing.assert_allclose(sc[::8, ::8, ::8], g["sub8"], rtol=0, atol=2e-5)
        st = np.array([sc.min(), sc.max(), sc.mean(dtype=np.float64), sc.std(dtype=np.float64)])
        np.testing.assert_allclose(st, g["stats"], rtol=1e-5, atol=1e-6)


def test_config3_batch_is_independent_of_the_sharding(K):
    """BASELINE config 3 in miniature: a batch of volumes produced under (base_seed, index) keys gives the
    same volumes whether one rank makes all of them or they are dealt round-robin to 2 / 4 ranks."""
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (32, 32, 32)
    n_items = 8
    banks, segs = [], []
    for v in range(4):
        seg, seeds = make_seed_volumes(shape, v)
        banks.append(SeedBank(seeds, DEV))
        segs.append(dev(seg))

    def produce(i):
        gen = make_generator(shape, DEV, rng="device", prob=0.9, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
        sharding.seed_for_sample(2024, i)
        out, seg, _, _ = gen._pipeline(None, segs[i % 4], banks[i % 4], {}, scale01=True)
        return host(out), host(seg)

    ref = [produce(i) for i in range(n_items)]
    for world in (2, 4):
        got = {}
        for rank in range(world):
            for i in sharding.shard(n_items, rank, world):
                got[i] = produce(i)
        for i in range(n_items):
            assert np.array_equal(got[i][0], ref[i][0]) and np.array_equal(got[i][1], ref[i][1])
    assert not np.array_equal(ref[0][0], ref[4][0])  # same label volume, different key -> different sample


def test_config4_384_hot_path_properties(K):
    """BASELINE config 4 grid (384^3, 0.5 mm): the hot path beyond the Infinity-Cache-resident size.  Checked
    through size-independent properties (a full oracle run at 384^3 takes minutes on the host):
    labels are a pure gather of the input labels, the [0,1] image spans exactly [0,1], the no-deformation /
    no-augmentation run reproduces the GMM draw, and the result is deterministic."""
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (384, 384, 384)
    seg, seeds = make_seed_volumes(shape)
    bank = SeedBank(seeds, DEV)
    segd = dev(seg)
    gen = make_generator(shape, DEV, rng="device", prob=1.0)
    outs = []
    for rep in range(2):
        np.random.seed(5)
        torch.manual_seed(5)
        out, lab, _, params = gen._pipeline(None, segd, bank, {}, scale01=True)
        outs.append((out, lab))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    out, lab = outs[0]
    assert float(out.min()) == 0.0 and float(out.max()) == 1.0 and bool(torch.isfinite(out).all())
    assert set(torch.unique(lab).tolist()) <= set(np.unique(seg).tolist())
    frac = float((lab > 0).float().mean())
    assert 0.5 * float((segd > 0).float().mean()) < frac < 1.6 * float((segd > 0).float().mean())
    assert params["deform_params"]["non_rigid"]["size_F_small"][0] in range(11, 25)
    # the oracle on a coarse sub-lattice of the SAME deformation: labels at every 16th voxel must match exactly.
    # The host plan is replayed (same seeds -> same draws, in the reference's order) to recover A, c2 and the
    # coarse field; the oracle then evaluates zoom + coordinates only at the lattice points.
    from fetalsyngen_amd import rng as _rng

    np.random.seed(5)
    torch.manual_seed(5)
    with _rng.use("device"):
        gen.intensity_generator.draw_subclusters({})
        gen.intensity_generator.plan_intensities(shape, {})
        dplan = gen.spatial_deform.plan(shape, random_shift=True, genparams={})
    dp = params["deform_params"]
    assert dplan.active and np.array_equal(dplan.params["affine"]["rotations"], dp["affine"]["rotations"])
    assert dplan.params["non_rigid"]["size_F_small"] == dp["non_rigid"]["size_F_small"] and dplan.flip == dp["flip"]
    A64 = O.affine_matrix(dp["affine"]["rotations"], dp["affine"]["shears"], dp["affine"]["scalings"])
    assert np.array_equal(A64.astype(np.float32), dplan.A.numpy())
    # every 16th voxel plus the last index of each axis (the corners are where the clamps bite)
    index = [np.unique(np.r_[np.arange(0, n, 16), n - 1]) for n in shape]
    fs = dplan.field_small
    f_sub = O.linear_zoom(fs, np.array(shape) / np.array(fs.shape[:3]), index=index)
    ii, jj, kk, margins = O.deformation_coords(shape, gen.spatial_deform.size, dplan.A, dplan.c2, f_sub, index=index)
    assert tuple(margins[:3]) == (0, 0, 0), "sub-lattice must already contain a coordinate < 1 per axis (floor(min) = 0)"
    seg_t = t(seg)
    want = O.sample_nearest(torch.flip(seg_t, [0]) if dplan.flip else seg_t, ii, jj, kk)
    got = lab.cpu()[np.ix_(*index)]
    assert got.shape == want.shape and torch.equal(got.float(), want.float()), "labels on the 16-voxel lattice, exact"
    assert int((want > 0).sum()) > 500  # the lattice does see the labelled region
    lazy = make_generator(shape, DEV, rng="device", prob=0.0)
    np.random.seed(5)
    torch.manual_seed(5)
    raw, lab0, _, p0 = lazy._pipeline(None, segd, bank, {}, scale01=False)
    assert torch.equal(lab0, segd)
    m2s = p0["selected_seeds"]["mlabel2subclusters"]
    mus, sig = p0["seed_intensities"]["mus"], p0["seed_intensities"]["sigmas"]
    cnt, mean, var = K.label_stats(bank.combined(m2s), raw, 50)
    for l in torch.nonzero(cnt > 200000).flatten().tolist():
        if float(mus[l]) > 4 * float(sig[l]):
            assert abs(float(mean[l]) - float(mus[l])) < 0.05 and abs(float(var[l]) ** 0.5 - float(sig[l])) < 0.05


@pytest.mark.parametrize("prob", [1.0, 0.5, 0.0])
def test_uint8_label_output_equals_the_float_labels(K, prob):
    """labels_u8=True (the device-resident stream's hand-over): the fused warp writes uint8 labels itself.  Same image bit
    for bit, labels equal to the float32 ones, same generator states afterwards -- with every gate on, with gates failing
    at random (no deformation: the labels pass through) and with all of them off."""
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (56, 48, 64)
    seg, seeds = make_seed_volumes(shape, 1)
    segd, bank = dev(seg), SeedBank(seeds, DEV)
    gen = make_generator(shape, DEV, rng="device", prob=prob, nonlin_scale=(0.08, 0.2), bf_scale=(0.03, 0.12))
    for i in range(6):
        np.random.seed(40 + i)
        torch.manual_seed(40 + i)
        o1, l1, _, p1 = gen._pipeline(None, segd, bank, {}, scale01=True)
        tail1 = (float(np.random.rand()), float(torch.rand(1)))
        np.random.seed(40 + i)
        torch.manual_seed(40 + i)
        o2, l2, _, p2 = gen._pipeline(None, segd, bank, {}, scale01=True, labels_u8=True)
        tail2 = (float(np.random.rand()), float(torch.rand(1)))
        assert l2.dtype == torch.uint8 and l1.dtype == torch.float32
        assert torch.equal(o1, o2) and torch.equal(l1, l2.float()) and tail1 == tail2
        assert p1["resample_params"] == p2["resample_params"]
    # the batched form
    np.random.seed(77)
    torch.manual_seed(77)
    want = [gen._pipeline(None, segd, bank, {}, scale01=True) for _ in range(3)]
    for streams in (1, 2):
        np.random.seed(77)
        torch.manual_seed(77)
        out, lab, _, _ = gen.sample_batch([(None, segd, bank)] * 3, scale01=True, streams=streams, labels_u8=True)
        assert lab.dtype == torch.uint8
        for b in range(3):
            assert torch.equal(out[b], want[b][0]) and torch.equal(lab[b].float(), want[b][1])


@pytest.mark.parametrize("shape", [(64, 56, 72), (256, 256, 256)])
def test_head_overlap_reproduces_the_in_order_stream(K, shape):
    """fsg_sample_plan::overlap: the parameter upload and the head of sample n+1 run on the library's side stream beside the
    resampling tail of sample n.  The results must be those of the plain in-order launch (FSG_HEAD_OVERLAP=0: upload on the
    launch stream before the call) bit for bit, for a sequence that keeps invalidating the ordering point: seed banks the
    generator has not used yet, a bank rewritten in place, a `sample_batch` in between, a second launch stream, and user
    work on the outputs enqueued between the calls (256^3: kernels long enough for a missing dependency to be hit)."""
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.generator import model as M
    from fetalsyngen_amd.phantom import make_seed_volumes

    nsub = 3
    subjects = []
    for v in range(nsub):
        seg, seeds = make_seed_volumes(shape, v)
        subjects.append((dev(seg), SeedBank(seeds, DEV)))
    small = shape[0] < 128
    gen = make_generator(shape, DEV, rng="device", prob=1.0, nonlin_scale=(0.08, 0.2) if small else (0.03, 0.06),
                         bf_scale=(0.03, 0.12) if small else (0.004, 0.02))
    other = torch.cuda.Stream()

    def run(mode):
        prev = M._HEAD_OVERLAP
        M._HEAD_OVERLAP = mode
        try:
            sums = []
            keep = None
            for i in range(14):
                np.random.seed(100 + i)
                torch.manual_seed(100 + i)
                seg, bank = subjects[i % nsub]
                if i == 6:  # a bank the generator has never seen, uploaded right before the call
                    seg2, seeds2 = make_seed_volumes(shape, 7)
                    seg, bank = dev(seg2), SeedBank(seeds2, DEV)
                if i == 8:  # rewritten in place (same storage, new _version)
                    for d in bank.vol.values():
                        for t in d.values():
                            t.copy_(t.flip(0).contiguous())
                if i == 9:
                    out, lab, _, _ = gen.sample_batch([(None, seg, bank)], streams=1)
                    out, lab = out[0], lab[0]
                elif i == 11:
                    with torch.cuda.stream(other):
                        other.wait_stream(torch.cuda.current_stream())
                        out, lab, _, _ = gen.sample(None, seg, bank)
                    torch.cuda.current_stream().wait_stream(other)
                else:
                    out, lab, _, _ = gen.sample(None, seg, bank)
                keep = (out * 2.0).sum()  # user work on the launch stream between the calls; its block is freed next turn
                sums.append((out.double().sum().item(), lab.double().sum().item(), out[3, 5, 7].item(), keep.item()))
                if i == 8:
                    for d in bank.vol.values():
                        for t in d.values():
                            t.copy_(t.flip(0).contiguous())
            torch.cuda.synchronize()
            return sums
        finally:
            M._HEAD_OVERLAP = prev

    want = run("0")
    got = run("2")
    assert got == want
    assert run("1") == want  # upload inside the call, launch stream only


@pytest.mark.parametrize("prob", [1.0, 0.6])
def test_sample_batch_equals_consecutive_samples(K, tmp_path, prob):
    """B volumes per call.  `sample_batch` (one parameter upload, one fsg_sample_run_batch call, 1 or 2 HIP
    streams) must reproduce B consecutive `sample` calls bit for bit -- images, labels, synth_params and the position of
    both host generators afterwards -- for B in {1, 4}, with every gate on and with gates failing at random."""
    from tests.util_bids import write_tree
    from fetalsyngen_amd.data.datasets import FetalSynthDataset, SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (48, 40, 56)
    subjects = []
    for v in range(4):
        seg, seeds = make_seed_volumes(shape, v)
        subjects.append((dev(seg), SeedBank(seeds, DEV)))
    gen = make_generator(shape, DEV, rng="device", prob=prob, nonlin_scale=(0.08, 0.2), bf_scale=(0.03, 0.12))

    def run_single(order):
        np.random.seed(21)
        torch.manual_seed(21)
        res = [gen.sample(None, subjects[k][0], subjects[k][1]) for k in order]
        return res, next_draws_pair()

    def next_draws_pair():
        return float(np.random.rand()), float(torch.rand(1))

    for order in ([2], [0, 3, 1, 2]):
        want, tail = run_single(order)
        for streams in (1, 2):
            np.random.seed(21)
            torch.manual_seed(21)
            out, lab, imgs, params = gen.sample_batch([(None, subjects[k][0], subjects[k][1]) for k in order], streams=streams)
            assert next_draws_pair() == tail
            assert tuple(out.shape) == (len(order), *shape) and imgs == [None] * len(order)
            for b, (wo, wl, _wi, wp) in enumerate(want):
                assert torch.equal(out[b], wo) and torch.equal(lab[b], wl), (order, streams, b)
                assert wp["resample_params"] == params[b]["resample_params"] and wp["noise_params"] == params[b]["noise_params"]
                assert torch.equal(wp["seed_intensities"]["mus"], params[b]["seed_intensities"]["mus"])
    # dataset level: the collated contract, and scale01 like __getitem__
    bids, seed_dir = write_tree(tmp_path, (32, 32, 32), ["sub-a", "sub-b", "sub-c"])
    g2 = make_generator((32, 32, 32), DEV, rng="device", prob=prob, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
    ds = FetalSynthDataset(str(bids), g2, str(seed_dir), None)
    np.random.seed(5)
    torch.manual_seed(5)
    singles = [ds[i] for i in (2, 0, 1)]
    np.random.seed(5)
    torch.manual_seed(5)
    batch, gps = ds.sample_batch([2, 0, 1], streams=2)
    assert batch["image"].shape == (3, 1, 32, 32, 32) and batch["label"].dtype == torch.int64 and not batch["image"].is_cuda
    assert batch["name"] == [s_["name"] for s_ in singles] and len(gps) == 3 and gps[1]["idx"] == 0
    for b, s_ in enumerate(singles):
        assert torch.equal(batch["image"][b], s_["image"]) and torch.equal(batch["label"][b], s_["label"])


def test_prefetching_stream_batched_and_half_precision(K, tmp_path):
    """Batched streaming (B samples per native call, one D2H copy per tensor and batch) yields exactly the per-sample
    stream's samples whatever the batch size (ragged last batch included); the optional float16 image is the float32 image
    rounded to nearest even."""
    from tests.util_bids import write_tree
    from fetalsyngen_amd.data.datasets import FetalSynthDataset
    from fetalsyngen_amd.data.staging import PrefetchingStream

    shape = (32, 32, 32)
    bids, seed_dir = write_tree(tmp_path, shape, ["sub-a", "sub-b"])
    gen = make_generator(shape, DEV, rng="device", prob=0.9, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
    ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None)
    ds.sample(0), ds.sample(1)
    n = 11
    want = [(d["image"].clone(), d["label"].clone(), d["name"]) for d in PrefetchingStream(ds, range(n), base_seed=4, depth=2)]
    for B, streams in ((4, 1), (3, 2), (16, 2)):
        got = []
        for item in PrefetchingStream(ds, range(n), base_seed=4, depth=2, batch_size=B, batch_streams=streams):
            assert item["image"].dim() == 5 and item["image"].shape[1] == 1 and len(item["name"]) == item["image"].shape[0]
            for b in range(item["image"].shape[0]):
                got.append((item["image"][b].clone(), item["label"][b].clone(), item["name"][b]))
        assert len(got) == n
        for (wi, wl, wn), (gi, gl, gn) in zip(want, got):
            assert wn == gn and torch.equal(wi, gi) and torch.equal(wl, gl)
    dev_items = list(PrefetchingStream(ds, range(n), base_seed=4, to_host=False, batch_size=4))
    assert dev_items[0]["image"].is_cuda and dev_items[0]["label"].dtype == torch.uint8 and dev_items[-1]["image"].shape[0] == 3
    assert torch.equal(dev_items[0]["image"][1].cpu(), want[1][0])
    half = [d for d in PrefetchingStream(ds, range(n), base_seed=4, depth=2, batch_size=4, image_dtype=torch.float16,
                                         label_dtype=torch.uint8)]
    assert half[0]["image"].dtype == torch.float16 and half[0]["label"].dtype == torch.uint8
    assert torch.equal(half[0]["image"][2], want[2][0].to(torch.float16)) and torch.equal(half[0]["label"][2].long(), want[2][1])
    x = torch.rand(1000003, device=DEV) * 3 - 1
    assert torch.equal(K.cast_f16(x), x.to(torch.float16))


@pytest.mark.parametrize("prob", [1.0, 0.5])
def test_fast_plan_equals_field_by_field_plan(K, prob):
    """The per-sample plan built as two flat arrays + fsg_sample_plan_pack (what `_pipeline` uses) against the plan filled
    field by field through ctypes: byte-identical structs for random samples (gates on and off, reference and device RNG),
    and identical outputs from the two paths."""
    import ctypes as C

    from fetalsyngen_amd import _lib
    from fetalsyngen_amd import rng as _rng
    from fetalsyngen_amd import tables as T
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.generator import model as M
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (40, 48, 56)
    seg, seeds = make_seed_volumes(shape, 2)
    bank, segd = SeedBank(seeds, DEV), dev(seg)
    lib = _lib.load()
    for mode in ("device", "reference"):
        gen = make_generator(shape, DEV, rng=mode, prob=prob, nonlin_scale=(0.08, 0.2), bf_scale=(0.03, 0.12))
        gen.register_label_twin(segd, segd.to(torch.uint8))  # both plans then see the same uint8 label source
        for seed in range(6):
            np.random.seed(seed)
            torch.manual_seed(seed)
            with _rng.use(gen.rng):
                arena = T.Arena()
                c = gen._prepare(None, segd, bank, {}, arena)
                arena.upload(DEV)
                gen._resolve(c)
                gen._native_operands(c)
                ws = gen._workspace(c.shape, gen._rows_needed(c))
                out = torch.empty(c.shape, dtype=torch.float32, device=DEV)
                seg_out = torch.empty_like(c.seg)
                slow = _lib.SamplePlan()
                assert gen._fill_native_plan(slow, c, True, ws, out, seg_out)
                assert gen._fast_operands(c) and gen._flat_plan(c, True, out, seg_out if c.dplan.active else c.seg, ws)
                fb = gen._flat
                fast = _lib.SamplePlan()
                _lib.check(lib.fsg_sample_plan_pack(C.byref(fast), fb["ivp"], gen._I["COUNT"], fb["fvp"], 17, fb["tbp"]), "pack")
                if not c.dplan.active:  # the field-by-field plan leaves seg pointers unset without a deformation; so does pack
                    assert fast.seg_in is None and fast.seg_out is None
                if mode == "reference":  # host-tape noise is uploaded per plan: fresh device tensors, so the pointers differ
                    for pl in (slow, fast):
                        assert pl.gmm_noise and (pl.noise or not c.nplan.active)
                        pl.gmm_noise, pl.noise = None, None
                a, b = bytes(slow), bytes(fast)
                assert a == b, [i for i in range(len(a)) if a[i] != b[i]][:16]
    # and end to end: both paths give the same sample
    gen = make_generator(shape, DEV, rng="device", prob=prob, nonlin_scale=(0.08, 0.2), bf_scale=(0.03, 0.12))
    res = {}
    for slow_plan in (False, True):
        M._SLOW_PLAN = slow_plan
        try:
            np.random.seed(3)
            torch.manual_seed(3)
            res[slow_plan] = [gen._pipeline(None, segd, bank, {}, scale01=True)[:2] for _ in range(4)]
        finally:
            M._SLOW_PLAN = False
    for (o1, l1), (o2, l2) in zip(res[False], res[True]):
        assert torch.equal(o1, o2) and torch.equal(l1, l2)


def test_workspace_eviction_across_streams(K):
    """More (shape, stream) keys than `FetalSynthGen._ws` keeps (4): samples interleaved over 6 streams evict
    each other's scratch volumes while kernels are still in flight; every result must equal the same sample
    made alone on the default stream."""
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (96, 96, 96)
    seg, seeds = make_seed_volumes(shape, 1)
    bank, segd = SeedBank(seeds, DEV), dev(seg)
    gen = make_generator(shape, DEV, rng="device", prob=1.0, nonlin_scale=(0.05, 0.12), bf_scale=(0.02, 0.08))

    def produce(i):
        sharding.seed_for_sample(77, i)
        out, lab, _, _ = gen._pipeline(None, segd, bank, {}, scale01=True)
        return out, lab

    want = [tuple(x.clone() for x in produce(i)) for i in range(24)]
    torch.cuda.synchronize()
    gen._ws.clear()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(6)]
    got, keys_seen = [], set()
    for i in range(24):
        with torch.cuda.stream(streams[i % 6]):
            got.append(produce(i))
        keys_seen.update(gen._ws)
        assert len(gen._ws) <= 4
    assert len(keys_seen) == 6  # six keys went through a cache of four: evictions happened mid-flight
    torch.cuda.synchronize()
    for (wo, wl), (go, gl) in zip(want, got):
        assert torch.equal(wo, go) and torch.equal(wl, gl)


def test_run_native_rejects_mismatched_operands(K):
    """The fused path only hands pointers to the C side: a segmentation (or seed volume) of another shape must be
    refused on the host, not gathered out of bounds on the device."""
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (32, 32, 32)
    seg, seeds = make_seed_volumes(shape, 0)
    bank = SeedBank(seeds, DEV)
    gen = make_generator(shape, DEV, rng="device", prob=1.0, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
    with pytest.raises(ValueError, match="segmentation shape"):
        gen._pipeline(None, dev(seg[:24]), bank, {}, scale01=True)
    small = make_seed_volumes((24, 32, 32), 0)[1]
    mixed = {n: {m: (small[n][m] if m == 2 else v) for m, v in d.items()} for n, d in seeds.items()}

    class Bank(SeedBank):
        pass

    with pytest.raises(ValueError, match="seed label volume"):
        gen._pipeline(None, dev(seg), Bank(mixed, DEV), {}, scale01=True)
    out, lab, _, _ = gen._pipeline(None, dev(seg), bank, {}, scale01=True)  # and the matching call still works
    assert tuple(out.shape) == shape and tuple(lab.shape) == shape


def test_config5_streaming_epoch_through_a_dataloader(K, tmp_path):
    """BASELINE config 5 in miniature: an epoch streamed through a torch DataLoader consumer (main-process
    producer, device-resident outputs and reference-contract CPU outputs)."""
    from tests.util_bids import write_tree
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import FetalSynthDataset

    shape = (32, 32, 32)
    bids, seed_dir = write_tree(tmp_path, shape, ["sub-a", "sub-b", "sub-c"])
    for return_device in (False, True):
        gen = make_generator(shape, DEV, rng="device", prob=0.9, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
        ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None, return_device=return_device)
        stream = sharding.ShardedSynthStream(lambda i: ds[i % len(ds)], n_items=12, base_seed=3, rank=0, world=1)
        loader = torch.utils.data.DataLoader(stream, batch_size=4, num_workers=0)
        seen, total = 0, 0.0
        for batch in loader:
            assert batch["image"].shape == (4, 1, *shape)
            assert batch["image"].is_cuda == return_device
            assert batch["label"].dtype == (torch.uint8 if return_device else torch.int64)
            total += float(batch["image"].sum())
            seen += batch["image"].shape[0]
        assert seen == 12 and np.isfinite(total)
        # restartable: the same (base_seed, index) gives the same sample
        sharding.seed_for_sample(3, 7)
        a = ds[7 % len(ds)]["image"]
        sharding.seed_for_sample(3, 7)
        b = ds[7 % len(ds)]["image"]
        assert torch.equal(a, b)


def test_host_stager_matches_synchronous_copy(K, tmp_path):
    """Pinned double-buffered D2H staging returns exactly what `.cpu()` would, in order, for more samples than
    ring slots; overrunning the ring is an error."""
    from tests.util_bids import write_tree
    from fetalsyngen_amd.data.datasets import FetalSynthDataset
    from fetalsyngen_amd.data.staging import HostStager, PrefetchingStream

    shape = (32, 32, 32)
    bids, seed_dir = write_tree(tmp_path, shape, ["sub-a", "sub-b"])
    gen = make_generator(shape, DEV, rng="device", prob=0.9, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
    ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None)
    ds.sample(0), ds.sample(1)  # fill the device caches
    want = [(d["image"].cpu().clone(), d["label"].cpu().clone())
            for d in PrefetchingStream(ds, range(7), base_seed=5, to_host=False)]
    got = [(d["image"].clone(), d["label"].clone()) for d in PrefetchingStream(ds, range(7), base_seed=5, depth=2)]
    assert len(got) == 7
    for (wi, wl), (gi, gl) in zip(want, got):
        assert not gi.is_cuda and not gl.is_cuda
        assert torch.equal(gi, wi) and gl.dtype == torch.int64 and torch.equal(gl, wl.long())
    st = HostStager(shape, DEV, depth=1, keep=1)
    x, y = torch.rand(shape, device=DEV), torch.ones(shape, device=DEV)
    tk = st.submit(x, y)
    tk2 = st.submit(x, y)
    with pytest.raises(RuntimeError, match="overrun"):
        st.submit(x, y)
    img, lab = st.collect(tk)
    assert torch.equal(img.view(shape), x.cpu()) and int(lab.sum()) == int(np.prod(shape))
    st.collect(tk2)


def test_host_stager_keeps_yielded_samples_alive_and_labels_intact(K, tmp_path):
    """Several 128^3 samples in flight (48 MiB of D2H each, so copies lag the generator), items consumed WITHOUT
    cloning in a `prev, cur` pattern: a yielded sample must stay intact while the next one is being used, and the
    label conversion must not read a recycled device block (the source of the int64 conversion is freed by the
    producer right after submit)."""
    from tests.util_bids import write_tree
    from fetalsyngen_amd.data.datasets import FetalSynthDataset
    from fetalsyngen_amd.data.staging import PrefetchingStream

    shape = (128, 128, 128)
    bids, seed_dir = write_tree(tmp_path, shape, ["sub-a"])
    gen = make_generator(shape, DEV, rng="device", prob=1.0)
    ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None)
    ds.sample(0)
    n = 10
    want = [(d["image"].cpu().clone(), d["label"].cpu().clone().long())
            for d in PrefetchingStream(ds, range(n), base_seed=9, to_host=False)]
    for label_dtype in (torch.int64, torch.uint8):
        prev, seen = None, 0
        for k, cur in enumerate(PrefetchingStream(ds, range(n), base_seed=9, depth=3, label_dtype=label_dtype)):
            assert cur["image"].is_pinned() and cur["label"].dtype == label_dtype
            if prev is not None:  # the previous item is still what it was when it was yielded
                assert torch.equal(prev["image"], want[k - 1][0]) and torch.equal(prev["label"].long(), want[k - 1][1])
            assert torch.equal(cur["image"], want[k][0]) and torch.equal(cur["label"].long(), want[k][1])
            prev, seen = cur, seen + 1
        assert seen == n
