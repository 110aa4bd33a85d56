#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite
never import the reference -- they read the .npz files this script wrote.  The reference
has no test-suite of its own, so these vectors are the parity pins.

What it does
  * installs `sys.modules` stand-ins for the reference's absent third-party deps
    (monai, SimpleITK, nibabel, hydra, skimage) -- none of them is on the hot path's
    arithmetic -- and replaces `torch.utils.cpp_extension.load` with a dummy BEFORE any
    reference import, so the reference's CUDA extension is never hipified/JIT-built;
  * `sys.dont_write_bytecode = True` so importing does not write into /root/reference;
  * records an RNG tape (numpy global + torch global draws, in order);
  * calls the reference functions on seeded inputs and stores inputs + outputs.

Usage:  python tests/golden/make_golden.py [--ref /root/reference] [--only NAME]
"""
from __future__ import annotations

import argparse
import gzip
import struct
import sys
import types
from pathlib import Path

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))


# --------------------------------------------------------------------------------------
# stand-ins for absent third-party modules
# --------------------------------------------------------------------------------------
def install_stubs():
    import torch.utils.cpp_extension as cpp_ext

    cpp_ext.load = lambda *a, **k: types.SimpleNamespace()  # never JIT the CUDA extension

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Transform:  # nominal base class only
        pass

    class _Identity:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x, *a, **k):
            return x

    class ScaleIntensity:
        def __init__(self, minv=0.0, maxv=1.0):
            self.minv, self.maxv = minv, maxv

        def __call__(self, x):
            mn, mx = x.min(), x.max()
            if mn == mx:
                return x * self.minv
            return (x - mn) / (mx - mn) * (self.maxv - self.minv) + self.minv

    class MetaTensor:  # must be a class (used in a `Tensor | MetaTensor` annotation)
        pass

    tr = mod(
        "monai.transforms",
        Transform=Transform,
        Spacing=_Identity,
        Orientation=_Identity,
        ScaleIntensity=ScaleIntensity,
        Compose=_Identity,
    )
    da = mod("monai.data", MetaTensor=MetaTensor)
    mod("monai", transforms=tr, data=da)
    mod("SimpleITK", ReadImage=None, GetArrayFromImage=None)
    mod("nibabel")
    hu = mod("hydra.utils", instantiate=None)
    mod("hydra", utils=hu)
    def ball(radius, dtype=np.uint8):
        """skimage.morphology.ball: voxels within `radius` of the centre of a (2r+1)^3 cube."""
        n = 2 * radius + 1
        Z, Y, X = np.mgrid[-radius : radius : n * 1j, -radius : radius : n * 1j, -radius : radius : n * 1j]
        return np.array(X**2 + Y**2 + Z**2 <= radius * radius, dtype=dtype)

    sm = mod("skimage.morphology", ball=ball)
    mod("skimage", morphology=sm)


# --------------------------------------------------------------------------------------
# RNG tape
# --------------------------------------------------------------------------------------
class Tape:
    """Records every numpy-global / torch-global draw in call order."""

    BIG = 4096

    def __init__(self):
        self.entries = []
        self._orig = {}

    def _rec(self, name, args, value):
        v = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
        self.entries.append((name, args, v))

    def __enter__(self):
        tape = self

        def wrap(owner, attr, name):
            orig = getattr(owner, attr)
            self._orig[(owner, attr)] = orig

            def f(*a, **k):
                out = orig(*a, **k)
                tape._rec(name, repr((a, {kk: str(vv) for kk, vv in k.items()})), out)
                return out

            setattr(owner, attr, f)

        for a in ("rand", "randn", "randint", "uniform"):
            wrap(np.random, a, "np." + a)
        for a in ("rand", "randn"):
            wrap(torch, a, "torch." + a)
        return self

    def __exit__(self, *exc):
        for (owner, attr), orig in self._orig.items():
            setattr(owner, attr, orig)

    def pack(self, prefix="tape"):
        """dict of arrays: small draws stored whole, big ones as (shape, head, sum)."""
        out = {f"{prefix}_n": np.int64(len(self.entries))}
        names = []
        for i, (name, _args, v) in enumerate(self.entries):
            names.append(name)
            v = np.asarray(v)
            if v.size <= self.BIG:
                out[f"{prefix}_{i}"] = v
            else:
                out[f"{prefix}_{i}_shape"] = np.asarray(v.shape, dtype=np.int64)
                out[f"{prefix}_{i}_head"] = v.reshape(-1)[:16].copy()
                out[f"{prefix}_{i}_sum"] = np.float64(v.astype(np.float64).sum())
        out[f"{prefix}_names"] = np.asarray(names)
        return out


def read_nifti(path):
    """Minimal NIfTI-1 (.nii.gz) reader: returns array in (x,y,z) order + pixdim."""
    raw = gzip.open(path, "rb").read()
    dim = struct.unpack("<8h", raw[40:56])
    datatype = struct.unpack("<h", raw[70:72])[0]
    pixdim = struct.unpack("<8f", raw[76:108])
    vox_offset = int(struct.unpack("<f", raw[108:112])[0])
    dt = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16}[datatype]
    n = dim[1] * dim[2] * dim[3]
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=vox_offset).reshape(dim[3], dim[2], dim[1])
    return np.ascontiguousarray(arr.transpose(2, 1, 0)), pixdim[1:4]


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path.name}: {path.stat().st_size/1024:.1f} KiB, {len(arrays)} arrays")


# --------------------------------------------------------------------------------------
# individual golden sets
# --------------------------------------------------------------------------------------
def g_affine(R):
    rng = np.random.default_rng(11)
    rots = np.concatenate([np.zeros((1, 3)), rng.uniform(-0.35, 0.35, (4, 3))])
    shs = np.concatenate([np.zeros((1, 3)), rng.uniform(-0.02, 0.02, (4, 3))])
    scs = np.concatenate([np.ones((1, 3)), rng.uniform(0.9, 1.1, (4, 3))])
    A = np.stack([R.gen.make_affine_matrix(r, s, c) for r, s, c in zip(rots, shs, scs)])
    save("affine", rot=rots, shear=shs, scale=scs, A=A)


def g_gauss(R):
    sig = np.array([0.3, 0.44, 0.87, 1.0, 1.77, 5.0])
    out = {"sigma": sig}
    for i, s in enumerate(sig):
        out[f"taps_{i}"] = R.gen.make_gaussian_kernel(float(s), "cpu").numpy()
    save("gauss_taps", **out)


def g_blur(R):
    rng = np.random.default_rng(21)
    out = {}
    stds = np.array([[1.2, 0.0, 0.7], [1.77, 1.77, 1.77], [0.3, 0.3, 0.3], [5.0, 5.0, 5.0], [0.0, 0.9, 0.0]])
    out["stds"] = stds
    for si, shape in enumerate([(24, 20, 16), (48, 48, 48), (7, 33, 10)]):
        x = (rng.random(shape, dtype=np.float32) * 255).astype(np.float32)
        out[f"x_{si}"] = x
        for ti, st in enumerate(stds):
            y = R.gen.gaussian_blur_3d(torch.from_numpy(x), st, "cpu")
            out[f"y_{si}_{ti}"] = y.numpy()
    save("blur", **out)


def g_zoom(R):
    rng = np.random.default_rng(31)
    out = {}
    cases = [((3, 3, 2, 3), (24, 20, 16)), ((5, 5, 5), (48, 48, 48)), ((40, 40, 40), (48, 48, 48)),
             ((1, 2, 1), (32, 32, 32)), ((20, 17, 29), (32, 32, 32)), ((9, 8, 7, 3), (40, 36, 28))]
    out["ncases"] = np.int64(len(cases))
    for i, (s, d) in enumerate(cases):
        x = rng.standard_normal(s).astype(np.float32)
        factor = np.array(d) / np.array(s[:3])
        y = R.gen.myzoom_torch(torch.from_numpy(x), factor)
        out[f"x_{i}"] = x
        out[f"factor_{i}"] = factor
        out[f"y_{i}"] = y.numpy()
    save("zoom", **out)


def g_interp(R):
    rng = np.random.default_rng(41)
    out = {}
    for ci, shape in enumerate([(16, 16, 16), (12, 10, 8)]):
        x = (rng.random(shape, dtype=np.float32) * 255).astype(np.float32)
        npts = (9, 8, 7)
        hi = np.array(shape, dtype=np.float32) - 1
        c = (rng.random((3,) + npts, dtype=np.float32) * (hi[:, None, None, None] + 2) - 1).astype(np.float32)
        flat = c.reshape(3, -1)
        # engineered edge coordinates
        flat[:, 0] = 0.0
        flat[:, 1] = hi
        flat[:, 2] = [2.5, 3.5, 4.5]
        flat[:, 3] = [0.5, 1.5, 0.5]
        flat[:, 4] = [-0.0, 1.0, 1.0]
        flat[:, 5] = [1.0, 0.0, 1.0]
        flat[:, 6] = [hi[0], 1.25, hi[2] - 0.5]
        flat[:, 7] = np.nextafter(hi, np.float32(0))
        flat[:, 8] = np.nextafter(np.float32(0), np.float32(1))
        flat[:, 9] = [5.5, 6.5, 3.49999]
        c = flat.reshape((3,) + npts)
        cc = np.clip(c, 0, hi[:, None, None, None]).astype(np.float32)  # deform_image clamps first
        for tag, co in (("raw", c), ("clamped", cc)):
            II, JJ, KK = (torch.from_numpy(co[a].copy()) for a in range(3))
            out[f"lin_{ci}_{tag}"] = R.gen.fast_3D_interp_torch(torch.from_numpy(x), II, JJ, KK, "linear").numpy()
            out[f"nn_{ci}_{tag}"] = R.gen.fast_3D_interp_torch(torch.from_numpy(x), II, JJ, KK, "nearest").numpy()
            out[f"coords_{ci}_{tag}"] = co
        out[f"x_{ci}"] = x
    save("interp", **out)


def g_deform_image(R):
    rng = np.random.default_rng(51)
    out = {}
    cases = []
    # (shape, size, rot, shear, scale, F?)
    cases.append(((24, 20, 16), (24, 20, 16), (0.2, -0.1, 0.3), (0.01, -0.02, 0.015), (1.05, 0.95, 1.0), True))
    cases.append(((32, 32, 32), (32, 32, 32), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.9, 0.9, 0.9), False))  # non-zero margins
    cases.append(((20, 24, 28), (16, 16, 16), (-0.3, 0.25, 0.1), (0.0, 0.02, -0.01), (1.1, 1.0, 0.92), True))
    out["ncases"] = np.int64(len(cases))
    for i, (shape, size, rot, sh, sc, useF) in enumerate(cases):
        sd = R.SpatialDeformation(20, 0.02, 0.1, list(size), 1.0, True, 0.03, 0.06, 4, 0.5, "cpu")
        sd._prepare_grid(shape)
        A = torch.tensor(R.gen.make_affine_matrix(np.array(rot), np.array(sh), np.array(sc)), dtype=torch.float)
        c2 = torch.tensor((np.array(shape) - 1) / 2, dtype=torch.float) + torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64)
        F = None
        if useF:
            fs = rng.standard_normal((4, 3, 5, 3)).astype(np.float32) * 2.0
            F = R.gen.myzoom_torch(torch.from_numpy(fs), np.array(shape) / np.array(fs.shape[:3]))
            out[f"Fsmall_{i}"] = fs
        xx2, yy2, zz2, x1, y1, z1, x2, y2, z2 = sd.deform_image(shape, A, c2, F)
        out[f"shape_{i}"] = np.array(shape)
        out[f"size_{i}"] = np.array(size)
        out[f"A_{i}"] = A.numpy()
        out[f"c2_{i}"] = c2.numpy()
        out[f"coords_{i}"] = np.stack([xx2.numpy(), yy2.numpy(), zz2.numpy()])
        out[f"margins_{i}"] = np.array([x1, y1, z1, x2, y2, z2]).astype(np.int64)
    save("deform_image", **out)


DEFAULT_SEED_LABELS = [0] + list(range(10, 50))
DEFAULT_GEN_CLASSES = [0] + [10] * 10 + [20] * 10 + [30] * 10 + list(range(40, 50))


def g_gmm(R):
    rng = np.random.default_rng(61)
    labs = np.array([0, 10, 11, 20, 30, 31, 32, 40, 49])
    seeds = labs[rng.integers(0, len(labs), (16, 16, 16))].astype(np.int64)
    out = {"seeds": seeds.astype(np.uint8)}
    for k, gc in enumerate([DEFAULT_GEN_CLASSES, DEFAULT_SEED_LABELS]):
        ifs = R.ImageFromSeeds(1, 6, DEFAULT_SEED_LABELS, gc)
        torch.manual_seed(0)
        with Tape() as tape:
            img, p = ifs.sample_intensities(torch.from_numpy(seeds), "cpu")
        out[f"img_{k}"] = img.numpy()
        out[f"mus_{k}"] = p["mus"].numpy()
        out[f"sigmas_{k}"] = p["sigmas"].numpy()
        out.update(tape.pack(f"tape{k}"))
    # fixed genparams
    ifs = R.ImageFromSeeds(1, 6, DEFAULT_SEED_LABELS, DEFAULT_GEN_CLASSES)
    torch.manual_seed(3)
    mus = 25 + 200 * torch.rand(50)
    sig = 5 + 20 * torch.rand(50)
    out["fixed_mus_in"] = mus.numpy().copy()
    out["fixed_sigmas_in"] = sig.numpy().copy()
    torch.manual_seed(4)
    img, p = ifs.sample_intensities(torch.from_numpy(seeds), "cpu", genparams={"mus": mus, "sigmas": sig})
    out["fixed_img"] = img.numpy()
    out["fixed_mus_out"] = p["mus"].numpy()
    save("gmm", **out)


def g_stages(R):
    """RandGamma / RandBiasField / RandResample / RandNoise / resize_back on 32^3."""
    rng = np.random.default_rng(71)
    x = (rng.random((32, 32, 32), dtype=np.float32) * 255).astype(np.float32)
    x[:4] = 0.0
    out = {"x": x}
    for seed in range(3):
        for gates in ("on", "off"):
            p = 1.0 if gates == "on" else 0.0
            gam = R.RandGamma(p, 0.1)
            bf = R.RandBiasField(p, 0.05, 0.2, 0.01, 0.3)
            rs = R.RandResample(p, 0.5, 1.5)
            nz = R.RandNoise(p, 5, 15)
            np.random.seed(seed)
            torch.manual_seed(seed)
            key = f"s{seed}_{gates}"
            with Tape() as tape:
                a, pa = gam(torch.from_numpy(x), "cpu")
                b, pb = bf(a, "cpu")
                c, factors, pc = rs(b, np.array([0.5, 0.5, 0.5]), "cpu")
                d, pd = nz(c, "cpu")
                e = rs.resize_back(d, factors)
            out[f"{key}_gamma"] = a.numpy()
            out[f"{key}_bias"] = b.numpy()
            out[f"{key}_resampled"] = c.numpy()
            out[f"{key}_noisy"] = d.numpy()
            out[f"{key}_back"] = e.numpy()
            if gates == "on":
                out[f"{key}_p_gamma"] = np.float64(pa["gamma"])
                out[f"{key}_p_bf_scale"] = np.asarray(pb["bf_scale"], dtype=np.float64)
                out[f"{key}_p_bf_std"] = np.asarray(pb["bf_std"], dtype=np.float64)
                out[f"{key}_p_bf_size"] = np.asarray(pb["bf_size"], dtype=np.int64)
                out[f"{key}_p_spacing"] = np.asarray(pc["spacing"], dtype=np.float64)
                out[f"{key}_p_factors"] = np.asarray(factors, dtype=np.float64)
                out[f"{key}_p_noise_std"] = np.float64(pd["noise_std"])
            out.update(tape.pack(f"{key}_tape"))
    # anisotropic fixed spacing through genparams (only some axes blurred)
    rs = R.RandResample(0.0, 0.5, 1.5)
    np.random.seed(5)
    torch.manual_seed(5)
    c, factors, pc = rs(torch.from_numpy(x), np.array([0.5, 0.5, 0.5]), "cpu", genparams={"spacing": [0.5, 0.8, 1.3]})
    out["aniso_resampled"] = c.numpy()
    out["aniso_factors"] = np.asarray(factors)
    out["aniso_back"] = rs.resize_back(c, factors).numpy()
    save("stages", **out)


class _Recorder:
    """Wraps a callable stage object and stores its first return value."""

    def __init__(self, inner, store, key):
        self.__dict__["_inner"] = inner
        self.__dict__["_store"] = store
        self.__dict__["_key"] = key

    def __call__(self, *a, **k):
        r = self._inner(*a, **k)
        self._store[self._key] = r[0].detach().clone()
        return r

    def __getattr__(self, n):
        return getattr(self._inner, n)


def build_generator(R, shape, prob=1.0, nonlin=(0.03, 0.06), size=None, bf_scale=(0.004, 0.02),
                    res=(0.5, 0.5, 0.5), rmin=0.5, rmax=1.5):
    size = list(shape) if size is None else list(size)
    return R.FetalSynthGen(
        shape=list(shape), resolution=list(res), device="cpu",
        intensity_generator=R.ImageFromSeeds(1, 6, DEFAULT_SEED_LABELS, DEFAULT_GEN_CLASSES),
        spatial_deform=R.SpatialDeformation(20, 0.02, 0.1, size, prob, True, nonlin[0], nonlin[1], 4, 0.5, "cpu"),
        resampler=R.RandResample(prob, rmin, rmax),
        bias_field=R.RandBiasField(prob, bf_scale[0], bf_scale[1], 0.01, 0.3),
        noise=R.RandNoise(prob, 5, 15),
        gamma=R.RandGamma(prob, 0.1),
    )


def run_e2e(R, shape, seed, variant=0, keep_stages=True, **gen_kw):
    from fetalsyngen_amd.phantom import make_seed_volumes

    seg, seeds = make_seed_volumes(shape, variant)
    table = {}
    paths = {}
    for n_sub, d in seeds.items():
        paths[n_sub] = {}
        for m, vol in d.items():
            key = f"seed_{n_sub}_{m}"
            table[key] = torch.from_numpy(vol.copy())
            paths[n_sub][m] = key
    gen = build_generator(R, shape, **gen_kw)
    gen.intensity_generator.loader = lambda p: table[p].clone()
    stages = {}
    ig = gen.intensity_generator
    orig_si = ig.sample_intensities

    def si(*a, **k):
        r = orig_si(*a, **k)
        stages["gmm"] = r[0].detach().clone()
        return r

    ig.sample_intensities = si
    sd = gen.spatial_deform
    orig_g = sd.generate_deformation_and_flip

    def gd(*a, **k):
        r = orig_g(*a, **k)
        if r[0] is not None:
            stages["coords"] = torch.stack([r[0], r[1], r[2]]).detach().clone()
        return r

    sd.generate_deformation_and_flip = gd
    orig_a = sd.apply_deformation_and_flip

    def ad(*a, **k):
        r = orig_a(*a, **k)
        stages["warped"] = r[2].detach().clone()
        return r

    sd.apply_deformation_and_flip = ad
    gen.gamma = _Recorder(gen.gamma, stages, "gamma")
    gen.biasfield = _Recorder(gen.biasfield, stages, "bias")
    gen.noise = _Recorder(gen.noise, stages, "noisy")
    gen.resampled = _Recorder(gen.resampled, stages, "resampled")

    np.random.seed(seed)
    torch.manual_seed(seed)
    with Tape() as tape:
        out, seg_out, _img, params = gen.sample(image=None, segmentation=torch.from_numpy(seg), seeds=paths)
    scaled = sys.modules["monai.transforms"].ScaleIntensity(0, 1)(out)
    res = {
        "shape": np.array(shape), "seed": np.int64(seed), "variant": np.int64(variant),
        "out": out.numpy(), "seg_out": seg_out.numpy().astype(np.uint8), "scaled": scaled.numpy(),
        "mus": params["seed_intensities"]["mus"].numpy(), "sigmas": params["seed_intensities"]["sigmas"].numpy(),
        "mlabel2subclusters": np.array([params["selected_seeds"]["mlabel2subclusters"][m] for m in range(1, 5)]),
        "flip": np.bool_(params["deform_params"]["flip"]),
    }
    assert np.array_equal(seg_out.numpy(), seg_out.numpy().astype(np.uint8))
    dp = params["deform_params"]
    if dp["affine"] is not None:
        res["rotations"] = np.asarray(dp["affine"]["rotations"])
        res["shears"] = np.asarray(dp["affine"]["shears"])
        res["scalings"] = np.asarray(dp["affine"]["scalings"])
        res["nonlin_scale"] = np.asarray(dp["non_rigid"]["nonlin_scale"], dtype=np.float64)
        res["nonlin_std"] = np.float64(dp["non_rigid"]["nonlin_std"])
        res["size_F_small"] = np.asarray(dp["non_rigid"]["size_F_small"])
    g = params["gamma_params"]["gamma"]
    res["gamma"] = np.float64(np.nan if g is None else g)
    sp = params["resample_params"]["spacing"]
    res["spacing"] = np.asarray([np.nan] * 3 if sp is None else sp, dtype=np.float64)
    ns = params["noise_params"]["noise_std"]
    res["noise_std"] = np.float64(np.nan if ns is None else ns)
    if params["bf_params"]["bf_size"] is not None:
        res["bf_size"] = np.asarray(params["bf_params"]["bf_size"])
        res["bf_std"] = np.asarray(params["bf_params"]["bf_std"], dtype=np.float64)
    if keep_stages:
        for k, v in stages.items():
            res["stage_" + k] = v.numpy()
    res.update(tape.pack("tape"))
    return res


def g_e2e(R):
    save("e2e_32_s0", **run_e2e(R, (32, 32, 32), 0))
    save("e2e_32_s1", **run_e2e(R, (32, 32, 32), 1, nonlin=(0.1, 0.3), bf_scale=(0.05, 0.2)))
    save("e2e_32_s2", **run_e2e(R, (32, 32, 32), 2, prob=0.5))
    save("e2e_48_s0", **run_e2e(R, (48, 48, 48), 0, keep_stages=False, nonlin=(0.08, 0.2)))
    save("e2e_nc_s1", **run_e2e(R, (40, 36, 28), 1, keep_stages=False, nonlin=(0.1, 0.2), variant=3))
    save("e2e_sz_s4", **run_e2e(R, (40, 40, 40), 4, keep_stages=False, nonlin=(0.1, 0.2), size=(32, 32, 32)))


def g_config1(R, ref):
    """BASELINE config 1: sub-sta21 decimated to 128^3, CPU reference, seed 0 (summary stats)."""
    data = Path(ref) / "data"
    seg, _ = read_nifti(data / "sub-sta21/anat/sub-sta21_rec-irtk_T2w_dseg.nii.gz")
    seg = np.ascontiguousarray(seg[::2, ::2, ::2]).astype(np.float32)
    table, paths = {}, {}
    for n_sub in range(1, 7):
        paths[n_sub] = {}
        for m in range(1, 5):
            p = data / f"derivatives/seeds/subclasses_{n_sub}/sub-sta21/anat/sub-sta21_rec-irtk_T2w_dseg_mlabel_{m}.nii.gz"
            paths[n_sub][m] = str(p)
    loaded = {}

    def loader(p):
        if p not in loaded:
            v, _ = read_nifti(p)
            loaded[p] = torch.from_numpy(np.ascontiguousarray(v[::2, ::2, ::2]))
        return loaded[p].clone()

    gen = build_generator(R, (128, 128, 128), prob=0.9, res=(1.0, 1.0, 1.0), rmin=1.0, rmax=3.0)
    gen.intensity_generator.loader = loader
    np.random.seed(0)
    torch.manual_seed(0)
    with Tape() as tape:
        out, seg_out, _img, params = gen.sample(image=None, segmentation=torch.from_numpy(seg), seeds=paths)
    scaled = sys.modules["monai.transforms"].ScaleIntensity(0, 1)(out).numpy()
    so = seg_out.numpy().astype(np.uint8)
    m2s = params["selected_seeds"]["mlabel2subclusters"]
    comb = sum(loaded[paths[m2s[m]][m]].numpy().astype(np.int16) for m in range(1, 5)).astype(np.uint8)
    res = {
        "seg_in": seg.astype(np.uint8), "seeds_in": comb,
        "mlabel2subclusters": np.array([m2s[m] for m in range(1, 5)]),
        "stats": np.array([scaled.min(), scaled.max(), scaled.mean(dtype=np.float64), scaled.std(dtype=np.float64)]),
        "label_counts": np.bincount(so.reshape(-1), minlength=8),
        "slice_x": scaled[64].astype(np.float16), "slice_y": scaled[:, 64].astype(np.float16),
        "slice_z": scaled[:, :, 64].astype(np.float16),
        "seg_slice_x": so[64], "seg_slice_y": so[:, 64], "seg_slice_z": so[:, :, 64],
        "sub8": scaled[::8, ::8, ::8].copy(), "seg_sub4": so[::4, ::4, ::4].copy(),
    }
    res.update(tape.pack("tape"))
    save("config1_sta21_128", **res)


def g_inputs(ref):
    """Input side: what the reference's bundled label files hold, read with THIS script's own
    reader (the reference reads them through SimpleITK + monai Orientation("RAS"), both absent here; the files
    carry diagonal positive sforms, so RAS orientation is the identity and the voxel array in (x,y,z) order is
    the expected tensor).  Stored: header fields, label histogram, float64 voxel sum, every 4th voxel."""
    data = Path(ref) / "data"
    files = {
        "dseg": data / "sub-sta21/anat/sub-sta21_rec-irtk_T2w_dseg.nii.gz",
        "seed": data / "derivatives/seeds/subclasses_3/sub-sta21/anat/sub-sta21_rec-irtk_T2w_dseg_mlabel_2.nii.gz",
    }
    res = {}
    for key, path in files.items():
        raw = gzip.open(path, "rb").read()
        arr, pixdim = read_nifti(path)
        res[f"{key}_relpath"] = np.asarray(str(path.relative_to(ref)))
        res[f"{key}_dim"] = np.array(struct.unpack("<8h", raw[40:56]))
        res[f"{key}_datatype"] = np.array(struct.unpack("<h", raw[70:72])[0])
        res[f"{key}_pixdim"] = np.array(struct.unpack("<8f", raw[76:108]), dtype=np.float32)
        res[f"{key}_codes"] = np.array(struct.unpack("<2h", raw[252:256]))  # qform_code, sform_code
        res[f"{key}_quatern"] = np.array(struct.unpack("<6f", raw[256:280]), dtype=np.float32)
        res[f"{key}_srow"] = np.array(struct.unpack("<12f", raw[280:328]), dtype=np.float32).reshape(3, 4)
        res[f"{key}_sum"] = np.float64(arr.astype(np.float64).sum())
        res[f"{key}_hist"] = np.bincount(arr.astype(np.int64).reshape(-1), minlength=64)
        res[f"{key}_sub4"] = arr[::4, ::4, ::4].astype(np.uint8)
        res[f"{key}_centre_x"] = arr[arr.shape[0] // 2].astype(np.uint8)
    save("inputs_sta21", **res)


def _rigid_transforms(rng, n, max_rot=0.5, max_t=3.0):
    """n random 3x4 [R|t] (rotation about a random axis, translation), float32."""
    out = np.zeros((n, 3, 4), dtype=np.float32)
    for i in range(n):
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        th = rng.uniform(-max_rot, max_rot)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        out[i, :, :3] = R
        out[i, :, 3] = rng.uniform(-max_t, max_t, 3)
    return out


def g_slice_acq(R):
    """SVoRT slice acquisition / adjoint: the reference's torch (CPU) fallback
    (svort/slice_acquisition/slice_acq.py:266-546); its CUDA kernels cannot run here."""
    import importlib

    # (the package's __init__ rebinds the name `slice_acquisition` to a function, so go through sys.modules)
    SA = importlib.import_module("fetalsyngen.generator.artifacts.svort.slice_acquisition.slice_acq")
    get_PSF = importlib.import_module("fetalsyngen.generator.artifacts.svort.data.utils").get_PSF

    rng = np.random.default_rng(91)
    out = {}
    psfs = {"aniso": get_PSF(res_ratio=(1, 1, 3)), "iso": get_PSF(res_ratio=(1.2, 1.2, 1.2)), "delta": get_PSF(0)}
    for k, v in psfs.items():
        out[f"psf_{k}"] = v.numpy()
    vol = rng.random((1, 1, 20, 24, 28), dtype=np.float32) * 100
    tr = _rigid_transforms(rng, 5)
    vmask = rng.random(vol.shape) > 0.2
    smask = rng.random((5, 1, 14, 18)) > 0.3
    out["vol"], out["transforms"], out["vol_mask"], out["slices_mask"] = vol[0, 0], tr, vmask[0, 0], smask[:, 0]
    T = torch.from_numpy
    for pk in ("aniso", "iso"):
        for mk, (vm, sm) in {"nomask": (None, None), "masks": (T(vmask), T(smask))}.items():
            s, w = SA.slice_acquisition_torch(T(tr), T(vol), vm, sm, psfs[pk], (14, 18), 1.3, True)
            out[f"fwd_{pk}_{mk}"] = s.numpy()[:, 0]
            out[f"fwdw_{pk}_{mk}"] = w.numpy()[:, 0]
            for eq in (False, True):
                v = SA.slice_acquisition_adjoint_torch(T(tr), psfs[pk], s, sm, vm, (20, 24, 28), 1.3, eq)
                out[f"adj_{pk}_{mk}_eq{int(eq)}"] = v.numpy()[0, 0]
    # 1x1x1 PSF without weights takes the grid_sample path (slice_acq.py:381-384, :445-480)
    s = SA.slice_acquisition_torch(T(tr), T(vol), None, None, psfs["delta"], (14, 18), 1.3, False)
    out["fwd_delta_nomask"] = s.numpy()[:, 0]
    out["res_slice"] = np.float32(1.3)
    save("slice_acq", **out)


# --------------------------------------------------------------------------------------
# SR-artifact stages: reference run on CPU (torch fallbacks of the CUDA extension)
# --------------------------------------------------------------------------------------
FIXED_CLOCK = 1700000000  # generate_fractal_noise_3d re-seeds numpy from int(time.time()) (artifacts/utils.py:365-367)


def _ref_sr():
    import importlib

    M = types.SimpleNamespace()
    M.AU = importlib.import_module("fetalsyngen.generator.artifacts.utils")
    M.SR = importlib.import_module("fetalsyngen.generator.artifacts.simulate_reco")
    M.FM = importlib.import_module("fetalsyngen.generator.artifacts.svort.data.fetal_motion")
    M.TR = importlib.import_module("fetalsyngen.generator.artifacts.svort.transform.transform")
    M.ART = importlib.import_module("fetalsyngen.generator.augmentation.artifacts")
    # the shipped trajectories (svort/data/traj.npy) are a pickle and are not loaded; the reference's sample_motion
    # code runs on the synthetic bank this repo uses in their place
    from fetalsyngen_amd.generator.artifacts.svort.scan import synthetic_trajectory_bank

    bank = synthetic_trajectory_bank()
    M.FM.get_trajectory = lambda: bank
    M.AU.time = types.SimpleNamespace(time=lambda: FIXED_CLOCK)
    return M


def _phantom(n, rng):
    """[0,1] image + label map 0..7 (nested shells; 2 = 'cortex', 3 = 'white matter') on an n^3 grid."""
    g = np.stack(np.meshgrid(*[np.arange(n, dtype=np.float32) - (n - 1) / 2] * 3, indexing="ij"))
    r = np.sqrt((g[0] / 1.0) ** 2 + (g[1] / 0.9) ** 2 + (g[2] / 0.8) ** 2) / (n / 2)
    seg = np.zeros((n, n, n), np.float32)
    for lab, rad in ((1, 0.85), (2, 0.75), (3, 0.62), (4, 0.40), (5, 0.30), (6, 0.2), (7, 0.1)):
        seg[r < rad] = lab
    img = (0.15 + 0.1 * seg) * (seg > 0) + 0.03 * rng.standard_normal((n, n, n)).astype(np.float32) * (seg > 0)
    img = np.clip(img, 0, None)
    return (img / img.max()).astype(np.float32), seg


def _next_draws():
    return np.array([np.random.rand(), float(torch.rand(1))])


def g_sr_units(R):
    """mog_3d_tensor, Perlin noise, axis-angle algebra, random stacks, scanner corruptions."""
    M = _ref_sr()
    out = {}
    rng = np.random.default_rng(3)
    # -- MoG (artifacts/utils.py:125-160)
    centers = [(3, 10, 20), (15.5, 2, 7), (22, 17, 1)]
    out["mog_centers"] = np.array(centers, np.float32)
    out["mog_a"] = M.AU.mog_3d_tensor((24, 20, 28), centers, 4.0, "cpu").numpy()
    sig = np.array([[2.0, 3.0, 4.0], [5.0, 1.5, 2.5], [3.0, 3.0, 9.0]])
    out["mog_sig"] = sig
    out["mog_b"] = M.AU.mog_3d_tensor((24, 20, 28), centers, sig, "cpu").numpy()
    out["mog_c"] = M.AU.mog_3d_tensor((24, 20, 28), centers, [torch.tensor([6.0]), torch.tensor([2.0]), torch.tensor([11.0])], "cpu").numpy()
    # -- Perlin (artifacts/utils.py:224-388)
    for tag, (shape, res, octv, inc) in {"p1": ((32, 32, 32), 2, 2, 0.1), "p2": ((48, 32, 16), 1, 4, 0.25),
                                         "p3": ((24, 24, 24), 2, 1, 0.0)}.items():
        torch.manual_seed(17)
        np.random.seed(5)
        n = M.AU.generate_fractal_noise_3d(shape, (res, res, res), octaves=octv, persistence=0.5, lacunarity=2,
                                           increase=inc, device="cpu")
        out[f"perlin_{tag}"] = n.numpy()
        out[f"perlin_{tag}_cfg"] = np.array([*shape, res, octv, inc], np.float64)
        out[f"perlin_{tag}_next"] = _next_draws()
    # -- rigid algebra (transform/transform_convert.py, transform/transform.py)
    ax = np.concatenate([rng.uniform(-3, 3, (40, 3)), rng.uniform(-30, 30, (40, 3))], 1).astype(np.float32)
    ax[0, :3] = 0
    ax[1, :3] = 2e-4
    ax[2, :3] = [np.pi * 0.999, 0, 0]
    ax[3, :3] = [0, 3.1, 0.01]
    TC = sys.modules["fetalsyngen.generator.artifacts.svort.transform.transform_convert"]
    m = TC.axisangle2mat_cpu(torch.from_numpy(ax))
    out["ax"], out["ax_mat"], out["ax_back"] = ax, m.numpy(), TC.mat2axisangle_cpu(m).numpy()
    a, b = M.TR.RigidTransform(torch.from_numpy(ax[:20])), M.TR.RigidTransform(m[20:], trans_first=False)
    out["compose"] = a.compose(b).matrix().numpy()
    out["inv"] = a.inv().matrix().numpy()
    out["b_ax_first"] = b.axisangle(trans_first=True).numpy()
    np.random.seed(11)
    st = M.TR.random_init_stack_transforms(9, 2.5, False, 3.0, "cpu")
    out["stack_ax"] = st.axisangle().numpy()
    out["stack_reset"] = M.TR.reset_transform(st[torch.tensor([False, True, True, True, True, False, False, False, False])]).axisangle().numpy()
    out["stack_upd"] = M.TR.mat_update_resolution(st.matrix(), 0.8, 0.5).numpy()
    mo = M.FM.sample_motion(np.arange(9) * 1.3, "cpu", True)
    out["motion"] = mo.matrix().numpy()
    out["units_next"] = _next_draws()
    # -- scanner corruptions (simulate_reco.py:210-298)
    sc = M.SR.Scanner(0.5, 2, 1.5, 1.5, 3.5, 1.5, 5.5, 2, 6, 250, 0.0, 0.1, 1, 2, prob_gamma=1.0, gamma_std=0.05,
                      prob_void=0.5, slice_size=None, restrict_transform=False, txy=3.0)
    s0 = (rng.random((7, 1, 20, 24), dtype=np.float32) * (rng.random((7, 1, 20, 24)) > 0.3)).astype(np.float32)
    out["slices_in"] = s0
    np.random.seed(21)
    torch.manual_seed(22)
    s1 = sc.random_gamma(torch.from_numpy(s0.copy()))
    out["slices_gamma"] = s1.numpy()
    s2 = sc.add_noise(s1.clone())
    out["slices_noise"] = s2.numpy()
    s3 = sc.signal_void(s2.clone())
    out["slices_void"] = s3.numpy()
    out["corrupt_next"] = _next_draws()
    save("sr_units", **out)


SCANNER_KW = dict(resolution_slice_fac_min=0.5, resolution_slice_fac_max=2, resolution_slice_max=1.5, slice_thickness_min=1.5,
                  slice_thickness_max=3.5, gap_min=1.5, gap_max=5.5, min_num_stack=2, max_num_stack=6, max_num_slices=250,
                  noise_sigma_min=0, noise_sigma_max=0.1, TR_min=1, TR_max=2, prob_void=0.2, prob_gamma=0.1, gamma_std=0.05,
                  slice_size=None, restrict_transform=False, txy=3.0)
MERGE_KW = dict(perlin_res_list=[1, 2], perlin_octaves_list=[1, 2, 4], perlin_persistence=0.5, perlin_lacunarity=2,
                gauss_ngaussians_min=2, gauss_ngaussians_max=4, perlin_increase_size=0.25)
RECON_KW = dict(prob_misreg_slice=0.1, slices_misreg_ratio=0.1, prob_misreg_stack=0.1, txy=3.0, prob_merge=1.0, prob_smooth=0.2,
                prob_rm_slices=0.3, rm_slices_min=0.1, rm_slices_max=0.4)


def g_sr_motion(R):
    """Scanner.scan + PSFReconstructor.recon_psf and SimulateMotion.__call__ on a 32^3 phantom, CPU."""
    M = _ref_sr()
    rng = np.random.default_rng(8)
    img, seg = _phantom(32, rng)
    out = {"img": img, "seg": seg}
    for case, (seed, merge_type, over) in {"a": (1, "perlin", {}), "b": (4, "gaussian", dict(prob_smooth=1.0, prob_rm_slices=1.0,
                                                                                           prob_misreg_stack=1.0, prob_misreg_slice=1.0)),
                                           "c": (6, "perlin", dict(prob_merge=0.0))}.items():
        np.random.seed(seed)
        torch.manual_seed(seed)
        sp = M.AU.ScannerParams(**{**SCANNER_KW, "prob_gamma": 0.5, "prob_void": 0.5})
        rp = M.AU.ReconParams(**{**RECON_KW, **over}, merge_params=M.AU.ReconMergeParams(merge_type=merge_type, **MERGE_KW))
        sm = M.ART.SimulateMotion(prob=1.0, scanner_params=sp, recon_params=rp)
        y, meta = sm(torch.from_numpy(img.copy()), torch.from_numpy(seg.copy()), "cpu", {}, resolution=[0.5, 0.5, 0.5])
        out[f"{case}_out"] = y.numpy()
        out[f"{case}_next"] = _next_draws()
        for k, v in meta.items():
            if k == "misreg_stack_on":
                v = np.array(v, dtype=np.int64)
            elif isinstance(v, str):
                v = np.array(v)
            elif v is None:
                v = np.array(np.nan)
            out[f"{case}_meta_{k}"] = np.asarray(v)
    # one explicit scan, to pin the intermediate slice stacks and transforms
    np.random.seed(13)
    torch.manual_seed(13)
    d = {"resolution": np.float64(0.5), "volume": torch.from_numpy(img.copy())[None, None],
         "mask": torch.from_numpy((seg > 0).astype(np.float32))[None, None], "seg": torch.from_numpy(seg.copy())[None, None],
         "threshold": 0.1}
    sc = M.SR.Scanner(**{**SCANNER_KW, "prob_gamma": 0.5, "prob_void": 0.5, "resolution_recon": np.float64(0.5)})
    ds = sc.scan(d)
    out["scan_stacks"] = ds["stacks"].numpy()[:, 0]
    out["scan_stacks_no_psf"] = ds["stacks_no_psf"].numpy()[:, 0]
    out["scan_positions"] = ds["positions"].numpy()
    out["scan_transforms"] = ds["transforms"].numpy()
    out["scan_transforms_gt"] = ds["transforms_gt"].numpy()
    out["scan_psf_rec"] = ds["psf_rec"].numpy()
    out["scan_meta"] = np.array([ds["resolution_slice"], ds["slice_thickness"], ds["gap"]])
    out["scan_next"] = _next_draws()
    # reconstruction grid coarser than the input grid (resolution_recon=None: drawn between the two resolutions), which
    # resamples the ground truth with grid_sample (simulate_reco.py:319-328); then the reconstruction on that grid
    np.random.seed(29)
    torch.manual_seed(29)
    d = {"resolution": np.float64(0.5), "volume": torch.from_numpy(img.copy())[None, None],
         "mask": torch.from_numpy((seg > 0).astype(np.float32))[None, None], "seg": torch.from_numpy(seg.copy())[None, None],
         "threshold": 0.1}
    sc = M.SR.Scanner(**{**SCANNER_KW, "resolution_slice_fac_min": 1.6, "resolution_recon": None})
    ds = sc.scan(d)
    rp = M.AU.ReconParams(**{**RECON_KW, "prob_merge": 1.0}, merge_params=M.AU.ReconMergeParams(merge_type="perlin", **MERGE_KW))
    rec = M.SR.PSFReconstructor(**{f: getattr(rp, f) for f in rp.__dataclass_fields__})
    vol, w = rec.recon_psf(ds)
    out["scanr_meta"] = np.array([ds["resolution_recon"], ds["resolution_slice"], ds["slice_thickness"], ds["gap"]])
    out["scanr_volume_gt"] = ds["volume_gt"].numpy()[0, 0]
    out["scanr_seg_gt"] = ds["seg_gt"].numpy()[0, 0].astype(np.uint8)
    out["scanr_stacks"] = ds["stacks"].numpy()[:, 0]
    out["scanr_recon"] = vol.numpy()[0, 0]
    out["scanr_weight"] = w.numpy()
    out["scanr_next"] = _next_draws()
    save("sr_motion", **out)


def g_sr_volumetric(R):
    """BlurCortex and StructNoise (augmentation/artifacts.py:24-342) on a 48^3 phantom, CPU."""
    M = _ref_sr()
    rng = np.random.default_rng(9)
    img, seg = _phantom(48, rng)
    out = {"img": img, "seg": seg}
    for case, seed in {"a": 2, "b": 5}.items():
        np.random.seed(seed)
        torch.manual_seed(seed)
        bc = M.ART.BlurCortex(prob=1.0, cortex_label=2, nblur_min=4, nblur_max=12)
        y, meta = bc(torch.from_numpy(img.copy()), torch.from_numpy(seg.copy()), "cpu", {})
        out[f"blur_{case}"], out[f"blur_{case}_nblur"], out[f"blur_{case}_next"] = y.numpy(), np.int64(meta["nblur"]), _next_draws()
    for case, (seed, mt) in {"a": (3, "perlin"), "b": (7, "gaussian"), "c": (12, "perlin")}.items():
        np.random.seed(seed)
        torch.manual_seed(seed)
        mp = M.AU.StructNoiseMergeParams(merge_type=mt, gauss_nloc_min=5, gauss_nloc_max=15, gauss_sigma_mu=25, gauss_sigma_std=5,
                                         perlin_res_list=[1, 2], perlin_octaves_list=[1, 2, 4], perlin_persistence=0.5,
                                         perlin_lacunarity=2, perlin_increase_size=0.1)
        sn = M.ART.StructNoise(prob=1.0, wm_label=3, std_min=0.2, std_max=0.4, merge_params=mp, nstages_min=1, nstages_max=5)
        y, meta = sn(torch.from_numpy(img.copy()), torch.from_numpy(seg.copy()), "cpu", {})
        out[f"sn_{case}"], out[f"sn_{case}_next"] = y.numpy(), _next_draws()
        out[f"sn_{case}_meta"] = np.array([meta["nstages"], meta["noise_std"]])
    save("sr_volumetric", **out)


def g_sr_boundaries(R):
    """SimulatedBoundaries (augmentation/artifacts.py:428-604) on a 40^3 phantom, CPU; skimage's ball() is a stand-in
    with the same definition (voxels within `radius` of the centre)."""
    M = _ref_sr()
    rng = np.random.default_rng(10)
    img, seg = _phantom(40, rng)
    seg[r_ := (np.indices(seg.shape).sum(0) % 7 == 0) & (seg == 1)] = 0  # ragged outer surface
    img = img * (seg > 0) + 0.2 * (seg == 0)  # background signal, so the mask is visible in the output
    out = {"img": img.astype(np.float32), "seg": seg}
    for case, (seed, ph, pf) in {"halo": (3, 1.0, 0.0), "fuzzy": (4, 0.0, 1.0), "both": (6, 1.0, 1.0), "plain": (7, 0.0, 0.0),
                                 "none": (8, None, None)}.items():
        np.random.seed(seed)
        torch.manual_seed(seed)
        sb = M.ART.SimulatedBoundaries(prob_no_mask=1.0 if ph is None else 0.0, prob_if_mask_halo=ph or 0.0,
                                       prob_if_mask_fuzzy=pf or 0.0)
        y, meta = sb(torch.from_numpy(out["img"].copy()), torch.from_numpy(seg.copy()), "cpu", {})
        out[f"{case}_out"] = y.numpy().astype(np.float32)
        out[f"{case}_next"] = _next_draws()
        out[f"{case}_seeds"] = np.array([-1 if v is None else int(v) for v in
                                         (sb.halo_radius, sb.n_generate_fuzzy, sb.n_centers, sb.base_sigma)])
    # building blocks
    m = torch.from_numpy((seg > 0).astype(np.int32))
    for r in (1, 5, 9):
        out[f"halo_r{r}"] = sb.build_halo(m, r).numpy().astype(np.uint8)
    out["dilate7"] = M.AU.dilate(m, 7).numpy().astype(np.uint8)
    out["erode5"] = M.AU.erode(m, 5).numpy().astype(np.uint8)
    out["boxsum3"] = M.AU.apply_kernel(m, 3).numpy()[0, 0]
    save("sr_boundaries", **out)


def g_e2e_art(R):
    """FetalSynthGen.sample with all four SR-artifact stages on (default YAML values, gates forced), 48^3, CPU."""
    from fetalsyngen_amd.phantom import make_seed_volumes

    M = _ref_sr()
    shape = (48, 48, 48)
    seg, seeds = make_seed_volumes(shape, 0)
    table, paths = {}, {}
    for n_sub, d in seeds.items():
        paths[n_sub] = {}
        for m, vol in d.items():
            table[f"seed_{n_sub}_{m}"] = torch.from_numpy(vol.copy())
            paths[n_sub][m] = f"seed_{n_sub}_{m}"
    out = {}
    for case, seed in {"a": 3, "b": 10}.items():
        sn_merge = M.AU.StructNoiseMergeParams(merge_type="perlin", gauss_nloc_min=5, gauss_nloc_max=15, gauss_sigma_mu=25,
                                               gauss_sigma_std=5, perlin_res_list=[1, 2], perlin_octaves_list=[1, 2, 4],
                                               perlin_persistence=0.5, perlin_lacunarity=2, perlin_increase_size=0.1)
        arts = dict(
            blur_cortex=M.ART.BlurCortex(prob=1.0, cortex_label=2, nblur_min=50, nblur_max=200),
            struct_noise=M.ART.StructNoise(prob=1.0, wm_label=3, std_min=0.2, std_max=0.4, merge_params=sn_merge),
            simulate_motion=M.ART.SimulateMotion(prob=1.0, scanner_params=M.AU.ScannerParams(**SCANNER_KW),
                                                 recon_params=M.AU.ReconParams(**RECON_KW, merge_params=M.AU.ReconMergeParams(
                                                     merge_type="perlin", **MERGE_KW))),
            boundaries=M.ART.SimulatedBoundaries(prob_no_mask=0.0, prob_if_mask_halo=0.5, prob_if_mask_fuzzy=0.5),
        )
        gen = build_generator(R, shape, nonlin=(0.08, 0.2))
        gen.artifacts = arts
        gen.intensity_generator.loader = lambda p: table[p].clone()
        np.random.seed(seed)
        torch.manual_seed(seed)
        y, seg_out, _img, params = gen.sample(image=None, segmentation=torch.from_numpy(seg), seeds=paths)
        out[f"{case}_out"] = y.numpy()
        out[f"{case}_seg"] = seg_out.numpy().astype(np.uint8)
        out[f"{case}_next"] = _next_draws()
        a = params["artifacts"]
        out[f"{case}_meta"] = np.array([a["blur_cortex"]["nblur"], a["struct_noise"]["nstages"], a["simulate_motion"]["nstacks"],
                                        int(bool(a["boundaries"]["halo_on"])), int(bool(a["boundaries"]["fuzzy_on"]))])
    save("e2e_art_48", **out)


ALL = {
    "e2e_art": g_e2e_art,
    "sr_boundaries": g_sr_boundaries,
    "sr_units": g_sr_units, "sr_motion": g_sr_motion, "sr_volumetric": g_sr_volumetric,
    "slice_acq": g_slice_acq,
    "affine": g_affine, "gauss": g_gauss, "blur": g_blur, "zoom": g_zoom, "interp": g_interp,
    "deform_image": g_deform_image, "gmm": g_gmm, "stages": g_stages, "e2e": g_e2e,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()

    install_stubs()
    sys.path.insert(0, args.ref)
    R = types.SimpleNamespace()
    import fetalsyngen.utils.generation as gen
    from fetalsyngen.generator.deformation.affine_nonrigid import SpatialDeformation
    from fetalsyngen.generator.intensity.rand_gmm import ImageFromSeeds
    from fetalsyngen.generator.augmentation.synthseg import RandGamma, RandBiasField, RandResample, RandNoise
    from fetalsyngen.generator.model import FetalSynthGen

    R.gen, R.SpatialDeformation, R.ImageFromSeeds = gen, SpatialDeformation, ImageFromSeeds
    R.RandGamma, R.RandBiasField, R.RandResample, R.RandNoise = RandGamma, RandBiasField, RandResample, RandNoise
    R.FetalSynthGen = FetalSynthGen
    torch.set_num_threads(8)
    for name, fn in ALL.items():
        if args.only and name != args.only:
            continue
        print(name)
        fn(R)
    if not args.only or args.only == "config1":
        print("config1")
        g_config1(R, args.ref)
    if not args.only or args.only == "inputs":
        print("inputs")
        g_inputs(args.ref)


if __name__ == "__main__":
    main()
