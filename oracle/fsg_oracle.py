"""ORACLE -- CPU restatement of fetalsyngen's per-volume synthesis hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; `fetalsyngen_amd/` never does
(tests/test_no_oracle_in_product.py enforces that).

Parity status: PINNED -- every function below is checked bit-for-bit / to the stated
tolerance against golden vectors captured from the real reference imported in the build
container (`tests/golden/make_golden.py`, run against /root/reference @ 2025-06-20).  The
reference itself has no tests or fixtures for this path.

Written from the behaviour of the reference (file:line cited per function; paths relative
to /root/reference/fetalsyngen/), in torch-CPU / numpy ops with the same arithmetic per
element (same operand order, no fused multiply-add, fp32 unless noted) so results are
bit-identical to the reference's CPU path wherever the reference itself is deterministic.
Every stochastic input is an explicit argument ("injected noise"), so the same functions
check the HIP path both in host-tape RNG mode and in device-Philox RNG mode; `Replay`
draws them from the numpy/torch global generators in the reference's order.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as TF

F32 = torch.float32


# --------------------------------------------------------------------------------------
# L0 helpers (utils/generation.py)
# --------------------------------------------------------------------------------------
def affine_matrix(rot, shear, scale) -> np.ndarray:
    """3x3 float64 `SHx @ SHy @ SHz @ Rx @ Ry @ Rz`, rows scaled (utils/generation.py:39-71)."""
    cx, cy, cz = np.cos(rot[0]), np.cos(rot[1]), np.cos(rot[2])
    sx, sy, sz = np.sin(rot[0]), np.sin(rot[1]), np.sin(rot[2])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    shx = np.array([[1, 0, 0], [shear[1], 1, 0], [shear[2], 0, 1]])
    shy = np.array([[1, shear[0], 0], [0, 1, 0], [0, shear[2], 1]])
    shz = np.array([[1, 0, shear[0]], [0, 1, shear[1]], [0, 0, 1]])
    m = shx @ shy @ shz @ rx @ ry @ rz
    return m * np.asarray(scale, dtype=np.float64).reshape(3, 1)


def gaussian_taps(sigma: float) -> torch.Tensor:
    """Normalised 1-D taps, half-width ceil(3 sigma) (utils/generation.py:74-81)."""
    half = int(np.ceil(3 * sigma))
    t = torch.linspace(-half, half, 2 * half + 1, dtype=F32)
    g = torch.exp(-((t / sigma) ** 2) / 2)
    return g / g.sum()


def blur3d(x: torch.Tensor, stds) -> torch.Tensor:
    """Separable zero-padded blur, axis 0 then 1 then 2, axes with std<=0 skipped
    (utils/generation.py:84-110)."""
    y = x[None, None]
    for axis in range(3):
        if stds[axis] > 0:
            k = gaussian_taps(stds[axis])
            shape = [1, 1, 1, 1, 1]
            shape[2 + axis] = len(k)
            pad = [0, 0, 0]
            pad[axis] = len(k) // 2
            y = TF.conv3d(y, k.reshape(shape), stride=1, padding=tuple(pad))
    return torch.squeeze(y)


def zoom_axis_table(n_src: int, factor: float, n_dst: int):
    """Per-axis sample table of the separable linear zoom (utils/generation.py:315-363):
    positions `arange(delta, delta + n_dst/f, 1/f)[:n_dst]` in fp32, clamped to
    [0, n_src-1]; returns (floor idx int64, ceil idx int64, w_floor f32, w_ceil f32)."""
    delta = (1.0 - factor) / (2.0 * factor)
    v = torch.arange(delta, delta + n_dst / factor, 1 / factor, dtype=F32)[:n_dst]
    v[v < 0] = 0
    v[v > (n_src - 1)] = n_src - 1
    lo = torch.floor(v).int()
    hi = lo + 1
    hi[hi > (n_src - 1)] = n_src - 1
    w_hi = v - lo
    w_lo = 1 - w_hi
    return lo.long(), hi.long(), w_lo, w_hi


# The reference evaluates the zoom as three Python loops over output slices (utils/generation.py:374-386): ~770 iterations of
# five tensor ops each at 256^3, which is where a third of its CPU time goes (K2b 0.6 s, K5b 0.2 s, K9 0.43 s).
# REFERENCE_LOOPS = True makes `linear_zoom` take that form (same values element for element): `bench.py`'s cpu_baseline
# times the oracle this way so that the CPU figure has the reference's cost structure, not that of a vectorised port.
REFERENCE_LOOPS = False


def _linear_zoom_loops(y: torch.Tensor, factor, new) -> torch.Tensor:
    """The loop form of utils/generation.py:364-386 (y: (H,W,D,C))."""
    tabs = [zoom_axis_table(y.shape[a], float(factor[a]), int(new[a])) for a in range(3)]
    lo, hi, wl, wh = tabs[0]
    t1 = torch.zeros([int(new[0]), y.shape[1], y.shape[2], y.shape[3]], dtype=F32)
    for i in range(int(new[0])):
        t1[i, :, :] = wl[i] * y[lo[i], :, :] + wh[i] * y[hi[i], :, :]
    lo, hi, wl, wh = tabs[1]
    t2 = torch.zeros([int(new[0]), int(new[1]), y.shape[2], y.shape[3]], dtype=F32)
    for j in range(int(new[1])):
        t2[:, j, :] = wl[j] * t1[:, lo[j], :] + wh[j] * t1[:, hi[j], :]
    lo, hi, wl, wh = tabs[2]
    out = torch.zeros([int(new[0]), int(new[1]), int(new[2]), y.shape[3]], dtype=F32)
    for k in range(int(new[2])):
        out[:, :, k] = wl[k] * t2[:, :, lo[k]] + wh[k] * t2[:, :, hi[k]]
    return out


def linear_zoom(x: torch.Tensor, factor, index=None) -> torch.Tensor:
    """Separable linear resize, x then y then z (utils/generation.py:310-397).
    Vectorised over slices; per element the arithmetic is the reference's
    `w_lo*X[lo] + w_hi*X[hi]` (two fp32 products, one fp32 sum).
    `index` (three int64 index vectors) restricts the OUTPUT to that sub-lattice: the zoom is separable,
    so the selected elements are computed exactly as in the full result (used at sizes where the full
    192 MiB x 3 field is not worth materialising on the host)."""
    squeeze = x.dim() == 3
    y = x[..., None] if squeeze else x
    factor = np.asarray(factor, dtype=np.float64)
    new = np.round(np.array(y.shape[:3]) * factor).astype(int)
    if REFERENCE_LOOPS and index is None:
        out = _linear_zoom_loops(y, factor, new)
        return out[..., 0] if squeeze else out
    for axis in range(3):
        lo, hi, w_lo, w_hi = zoom_axis_table(y.shape[axis], float(factor[axis]), int(new[axis]))
        if index is not None:
            sel = torch.as_tensor(index[axis], dtype=torch.long)
            lo, hi, w_lo, w_hi = lo[sel], hi[sel], w_lo[sel], w_hi[sel]
        bshape = [1, 1, 1, 1]
        bshape[axis] = -1
        y = w_lo.reshape(bshape) * y.index_select(axis, lo) + w_hi.reshape(bshape) * y.index_select(axis, hi)
    return y[..., 0] if squeeze else y


def sample_nearest(x: torch.Tensor, ii, jj, kk) -> torch.Tensor:
    """Round-half-even nearest gather with index clamp (utils/generation.py:211-225)."""
    idx = []
    for c, n in zip((ii, jj, kk), x.shape[:3]):
        r = torch.round(c).long()
        idx.append(torch.clamp(r, 0, n - 1))
    return x[idx[0], idx[1], idx[2]]


def sample_linear(x: torch.Tensor, ii, jj, kk, default: float = 0.0) -> torch.Tensor:
    """Trilinear gather (utils/generation.py:227-285): valid iff every coordinate is
    in (0, n-1] (STRICT > 0); blend x, then y, then z; invalid voxels <- `default`."""
    n0, n1, n2 = x.shape[:3]
    ok = (ii > 0) & (jj > 0) & (kk > 0) & (ii <= n0 - 1) & (jj <= n1 - 1) & (kk <= n2 - 1)
    out = torch.full(ii.shape, float(default), dtype=F32)
    ci, cj, ck = ii[ok], jj[ok], kk[ok]

    def split(c, n):
        lo = torch.floor(c).long()
        hi = torch.clamp(lo + 1, max=n - 1)
        w_hi = c - lo
        return lo, hi, 1 - w_hi, w_hi

    x0, x1, ax, bx = split(ci, n0)
    y0, y1, ay, by = split(cj, n1)
    z0, z1, az, bz = split(ck, n2)
    c00 = x[x0, y0, z0] * ax + x[x1, y0, z0] * bx
    c01 = x[x0, y0, z1] * ax + x[x1, y0, z1] * bx
    c10 = x[x0, y1, z0] * ax + x[x1, y1, z0] * bx
    c11 = x[x0, y1, z1] * ax + x[x1, y1, z1] * bx
    c0 = c00 * ay + c10 * by
    c1 = c01 * ay + c11 * by
    out[ok] = (c0 * az + c1 * bz).float()
    return out


# --------------------------------------------------------------------------------------
# K1  GMM intensity draw (generator/intensity/rand_gmm.py)
# --------------------------------------------------------------------------------------
def gmm_tables(u_mu, u_sigma, z_class, seed_labels, generation_classes):
    """mus = 25+200 U, sigmas = 5+20 U; class-tied means
    `mus[seed_labels] = clamp(mus[generation_classes] + 25 z, 0, 225)` when the two
    label lists differ (rand_gmm.py:116-145).  u_*/z_class are the raw torch draws."""
    mus = 25 + 200 * u_mu
    sigmas = 5 + 20 * u_sigma
    if list(generation_classes) != list(seed_labels):
        mus[list(seed_labels)] = torch.clamp(mus[list(generation_classes)] + 25 * z_class, 0, 225)
    return mus, sigmas


def gmm_image(seeds: torch.Tensor, mus, sigmas, z: torch.Tensor) -> torch.Tensor:
    """`mus[l] + sigmas[l]*z`, negatives clamped to 0 (rand_gmm.py:146-149)."""
    s = seeds.long()
    img = mus[s] + sigmas[s] * z
    img[img < 0] = 0
    return img


# --------------------------------------------------------------------------------------
# K2/K3  deformation field (generator/deformation/affine_nonrigid.py)
# --------------------------------------------------------------------------------------
def centre_with_shift(shape, size, u_shift64=None) -> torch.Tensor:
    """c2 = (shape-1)/2 [+ 2*max_shift*U - max_shift], max_shift = max((shape-size)/2, 0)
    (affine_nonrigid.py:271-290).  With the shift the result is float64 (fp32 + fp64)."""
    base = torch.tensor((np.array(shape[:3]) - 1) / 2, dtype=F32)
    if u_shift64 is None:
        return base
    ms = torch.tensor(np.array(shape[:3]) - np.array(size), dtype=F32) / 2
    ms[ms < 0] = 0
    return base + (2 * (ms * u_shift64) - ms)


def deformation_coords(shape, size, A: torch.Tensor, c2: torch.Tensor, field, index=None):
    """Sampling coordinates of the deformed grid + the six margins
    (affine_nonrigid.py:64-84 grid/centre, :327-366 compose/clamp/margin-subtract).
    `field` is the full-resolution (H,W,D,3) displacement or None.
    With `index` (three int64 index vectors) only that sub-lattice of the grid is evaluated (`field` then
    has the sub-lattice's shape); the margins are those of the sub-lattice -- the caller has to make sure
    they equal the full grid's (e.g. floor(min) == 0 already on the sub-lattice, coordinates being >= 0)."""
    axes = [torch.arange(n, dtype=F32) for n in shape[:3]]
    if index is not None:
        axes = [a[torch.as_tensor(ix, dtype=torch.long)] for a, ix in zip(axes, index)]
    centre = torch.tensor((np.array(size) - 1) / 2, dtype=F32)
    g = torch.meshgrid(*axes, indexing="ij")
    p = [g[a] - centre[a] for a in range(3)]
    if field is not None:
        p = [p[a] + field[..., a] for a in range(3)]
    q = []
    for r in range(3):
        t = A[r, 0] * p[0] + A[r, 1] * p[1] + A[r, 2] * p[2] + c2[r]
        t[t < 0] = 0
        t[t > (shape[r] - 1)] = shape[r] - 1
        q.append(t)
    lo = [torch.floor(torch.min(t)) for t in q]
    hi = [1 + torch.ceil(torch.max(t)) for t in q]
    for r in range(3):
        q[r] -= lo[r]
    margins = np.array([int(v) for v in lo] + [int(v) for v in hi])
    return q[0], q[1], q[2], margins


def nonlinear_field(f_small: torch.Tensor, shape) -> torch.Tensor:
    """Upsample the coarse displacement grid to `shape` (affine_nonrigid.py:319)."""
    return linear_zoom(f_small, np.array(shape[:3]) / np.array(f_small.shape[:3]))


def apply_deformation(output, segmentation, coords, flip: bool):
    """flip along axis 0, then linear (intensity) / nearest (labels) sampling
    (affine_nonrigid.py:164-193)."""
    if flip:
        output = torch.flip(output, [0])
        segmentation = torch.flip(segmentation, [0])
    if coords is None:
        return output, segmentation
    ii, jj, kk = coords
    return sample_linear(output, ii, jj, kk), sample_nearest(segmentation, ii, jj, kk)


# --------------------------------------------------------------------------------------
# K5..K10  intensity augmentations (generator/augmentation/synthseg.py)
# --------------------------------------------------------------------------------------
def gamma_transform(x: torch.Tensor, gamma: float) -> torch.Tensor:
    """300*(x/300)**gamma with gamma a 0-dim float64 tensor (synthseg.py:269-274)."""
    return 300.0 * (x / 300.0) ** torch.tensor(gamma, dtype=torch.float64)


def bias_multiply(x: torch.Tensor, b_small: torch.Tensor) -> torch.Tensor:
    """x * exp(zoom(b_small)) (synthseg.py:178-182)."""
    bf = linear_zoom(b_small, np.array(x.shape) / np.array(b_small.shape))
    return x * torch.exp(bf)


def resample_plan(in_shape, resolution, spacing, u_std: float):
    """Blur stds, low-res size, factors and float64 per-axis sample positions
    (synthseg.py:64-98)."""
    spacing = np.array(spacing, dtype=np.float64)
    resolution = np.array(resolution, dtype=np.float64)
    size = np.array(in_shape)
    stds = (0.85 + 0.3 * u_std) * np.log(5) / np.pi * spacing / resolution
    stds[spacing <= resolution] = 0.0
    new_size = (size * resolution / spacing).astype(int)
    factors = new_size / size
    delta = (1.0 - factors) / (2.0 * factors)
    pos = [np.arange(delta[a], delta[a] + new_size[a] / factors[a], 1 / factors[a])[: new_size[a]] for a in range(3)]
    return stds, new_size, factors, pos


def resample_down(x: torch.Tensor, resolution, spacing, u_std: float):
    """Blur then trilinear resample at the axis-aligned positions (synthseg.py:78-105)."""
    stds, _new, factors, pos = resample_plan(x.shape, resolution, spacing, u_std)
    blurred = blur3d(x, stds)
    g = np.meshgrid(*pos, sparse=False, indexing="ij")
    ii, jj, kk = (torch.tensor(a, dtype=F32) for a in g)
    return sample_linear(blurred, ii, jj, kk), factors


def add_noise(x: torch.Tensor, noise_std, z: torch.Tensor) -> torch.Tensor:
    """x + std*z with std an fp32 tensor, negatives clamped (synthseg.py:225-233)."""
    y = x + torch.tensor(noise_std, dtype=F32) * z
    y[y < 0] = 0
    return y


def resize_back(x: torch.Tensor, factors):
    """zoom by 1/factors then divide by the global max (synthseg.py:109-114)."""
    if factors is None:
        return x
    y = linear_zoom(x, 1 / np.asarray(factors))
    return y / torch.max(y)


def scale01(x: torch.Tensor) -> torch.Tensor:
    """(x-min)/(max-min): monai ScaleIntensity(minv=0,maxv=1) as used at
    data/datasets.py:40,:311.  monai==1.4.0 is not under /root/reference; its published
    `rescale_array` is `(arr-mina)/(maxa-mina)*(maxv-minv)+minv` (arr*minv if flat)."""
    mn, mx = x.min(), x.max()
    if mn == mx:
        return x * 0.0
    return (x - mn) / (mx - mn) * 1.0 + 0.0


# --------------------------------------------------------------------------------------
# Replay: the whole path with the reference's RNG draw order
# --------------------------------------------------------------------------------------
DEFAULT_SEED_LABELS = [0] + list(range(10, 50))
DEFAULT_GEN_CLASSES = [0] + [10] * 10 + [20] * 10 + [30] * 10 + list(range(40, 50))


class Config:
    """Generator hyper-parameters (configs/dataset/generator/default.yaml:1-56)."""

    def __init__(self, shape, resolution=(0.5, 0.5, 0.5), size=None, prob=0.9, flip_prb=0.5,
                 max_rotation=20, max_shear=0.02, max_scaling=0.1, nonlinear=True,
                 nonlin_scale=(0.03, 0.06), nonlin_std_max=4, res_range=(0.5, 1.5),
                 bf_scale=(0.004, 0.02), bf_std=(0.01, 0.3), noise_std=(5, 15), gamma_std=0.1,
                 subclusters=(1, 6), seed_labels=None, generation_classes=None,
                 deform_prob=None, gamma_prob=None, bias_prob=None, resample_prob=None, noise_prob=None):
        self.shape = tuple(shape)
        self.size = tuple(shape) if size is None else tuple(size)
        self.resolution = tuple(resolution)
        p = prob
        self.deform_prob = p if deform_prob is None else deform_prob
        self.gamma_prob = p if gamma_prob is None else gamma_prob
        self.bias_prob = p if bias_prob is None else bias_prob
        self.resample_prob = p if resample_prob is None else resample_prob
        self.noise_prob = p if noise_prob is None else noise_prob
        self.flip_prb = flip_prb
        self.max_rotation, self.max_shear, self.max_scaling = max_rotation, max_shear, max_scaling
        self.nonlinear, self.nonlin_scale, self.nonlin_std_max = nonlinear, nonlin_scale, nonlin_std_max
        self.res_range, self.bf_scale, self.bf_std = res_range, bf_scale, bf_std
        self.noise_std, self.gamma_std, self.subclusters = noise_std, gamma_std, subclusters
        self.seed_labels = DEFAULT_SEED_LABELS if seed_labels is None else list(seed_labels)
        self.generation_classes = DEFAULT_GEN_CLASSES if generation_classes is None else list(generation_classes)


def run_sample(cfg: Config, segmentation: torch.Tensor, seed_volumes, *, noise_gmm=None, noise_lowres=None,
               keep_stages=False, image=None, draws=None):
    """One `FetalSynthGen.sample` + the dataset's final [0,1] scaling, drawing from the
    numpy / torch GLOBAL generators in the reference's order
    (model.py:231-276 -> rand_gmm.py:82-85,:120-148 -> affine_nonrigid.py:140-145,:248-263,
    :284,:303-318 -> synthseg.py:263-265,:157-176,:63-78,:218-232 -> datasets.py:311).

    seed_volumes[n_sub][mlabel] -> integer array (the decoded seed file).  `noise_gmm` /
    `noise_lowres`: when given, used INSTEAD of drawing the two large torch.randn fields
    (device-RNG parity: the HIP path's Philox noise is injected here); either may be a
    callable (shape)->tensor, invoked at the point of the draw order where the field is
    needed (the low-res shape is only known mid-way, and a callable may itself consume
    the torch generator exactly like the product's key draw does).
    `draws`: when given, NOTHING is drawn from the global generators: every stochastic input of the sample is taken from
    this dict (keyed mode of the HIP path: the product exports what it drew, the oracle runs the reference's arithmetic on
    it) -- "m2s" {mlabel: n_sub}, "mus" / "sigmas" (float32 tensors), "deform" None | {"flip", "A" (3,3) float32, "c2" (3,)
    float64, "f_small" float32 | None}, "gamma" None | float, "bias" None | float32 grid (already scaled by its std),
    "resample" None | {"spacing", "u_std"}, "noise_std" None | float64; the two large fields through noise_gmm /
    noise_lowres, which are then mandatory.
    Returns dict(out, seg, scaled, params, stages)."""
    if draws is not None:
        return _run_sample_injected(cfg, segmentation, seed_volumes, draws, noise_gmm, noise_lowres, keep_stages)
    st = {}
    if seed_volumes is not None:
        lo_s, hi_s = cfg.subclusters
        m2s = {m: int(np.random.randint(lo_s, hi_s + 1)) for m in range(1, 5)}
        seeds = None
        for m in range(1, 5):
            v = torch.as_tensor(np.asarray(seed_volumes[m2s[m]][m])).clone()
            seeds = v if seeds is None else seeds + v
        seeds = seeds.long()

        nlab = max(cfg.seed_labels) + 1
        u_mu = torch.rand(nlab, dtype=F32)
        u_sg = torch.rand(nlab, dtype=F32)
        z_cls = None
        if cfg.generation_classes != cfg.seed_labels:
            z_cls = torch.randn(len(cfg.seed_labels), dtype=F32)
        mus, sigmas = gmm_tables(u_mu, u_sg, z_cls, cfg.seed_labels, cfg.generation_classes)
        if noise_gmm is None:
            z = torch.randn(seeds.shape, dtype=F32)
        else:
            z = noise_gmm(tuple(seeds.shape)) if callable(noise_gmm) else noise_gmm
        out = gmm_image(seeds, mus, sigmas, z)
        params = {"mlabel2subclusters": m2s, "mus": mus, "sigmas": sigmas}
    else:
        # image as intensity prior, scaled to 0..255 (generator/model.py:131-140)
        if image is None:
            raise ValueError("If no seeds are passed, an image must be loaded to be used as intensity prior!")
        out = (image - image.min()) / (image.max() - image.min()) * 255
        params = {}
    st["gmm"] = out

    shape = tuple(out.shape)
    seg = segmentation
    coords, flip = None, False
    if np.random.rand() < cfg.deform_prob:
        flip = bool(np.random.rand() < cfg.flip_prb)
        rot = (2 * cfg.max_rotation * np.random.rand(3) - cfg.max_rotation) / 180.0 * np.pi
        sh = 2 * cfg.max_shear * np.random.rand(3) - cfg.max_shear
        sc = 1 + (2 * cfg.max_scaling * np.random.rand(3) - cfg.max_scaling)
        A = torch.tensor(affine_matrix(rot, sh, sc), dtype=F32)
        c2 = centre_with_shift(shape, cfg.size, torch.rand(3, dtype=torch.float64))
        field = None
        params.update(rotations=rot, shears=sh, scalings=sc)
        if cfg.nonlinear:
            nscale = cfg.nonlin_scale[0] + np.random.rand(1) * (cfg.nonlin_scale[1] - cfg.nonlin_scale[0])
            small = np.round(nscale * np.array(shape)).astype(int).tolist()
            nstd = cfg.nonlin_std_max * np.random.rand()
            f_small = nstd * torch.randn([*small, 3], dtype=F32)
            field = nonlinear_field(f_small, shape)
            params.update(nonlin_scale=nscale, nonlin_std=nstd, size_F_small=small, f_small=f_small)
        ii, jj, kk, margins = deformation_coords(shape, cfg.size, A, c2, field)
        coords = (ii, jj, kk)
        params.update(A=A, c2=c2, margins=margins)
        if keep_stages:
            st["coords"] = torch.stack(coords)
    params["flip"] = flip
    out, seg = apply_deformation(out, seg, coords, flip)
    st["warped"] = out
    image_def = None
    if image is not None:  # the real image follows the same field, linear (affine_nonrigid.py:183,:190-191)
        image_def, _ = apply_deformation(image, segmentation, coords, flip)

    gamma = None
    if np.random.rand() < cfg.gamma_prob:
        gamma = float(np.exp(cfg.gamma_std * np.random.randn(1)[0]))
        out = gamma_transform(out, gamma)
    params["gamma"] = gamma
    st["gamma"] = out

    params["bf_size"] = None
    if np.random.rand() < cfg.bias_prob:
        bscale = cfg.bf_scale[0] + np.random.rand(1) * (cfg.bf_scale[1] - cfg.bf_scale[0])
        bsize = np.maximum(np.round(bscale * np.array(out.shape)).astype(int), 1).tolist()
        bstd = cfg.bf_std[0] + (cfg.bf_std[1] - cfg.bf_std[0]) * np.random.rand(1)
        b_small = torch.tensor(bstd, dtype=F32) * torch.randn(bsize, dtype=F32)
        out = bias_multiply(out, b_small)
        params.update(bf_scale=bscale, bf_size=bsize, bf_std=bstd, b_small=b_small)
    st["bias"] = out

    factors, spacing = None, None
    if np.random.rand() < cfg.resample_prob:
        spacing = np.array([1.0, 1.0, 1.0]) * np.random.uniform(cfg.res_range[0], cfg.res_range[1])
        u_std = np.random.rand()
        out, factors = resample_down(out, cfg.resolution, spacing, u_std)
        params["u_std"] = u_std
    params["spacing"], params["factors"] = spacing, factors
    st["resampled"] = out

    nstd = None
    if np.random.rand() < cfg.noise_prob:
        nstd = cfg.noise_std[0] + (cfg.noise_std[1] - cfg.noise_std[0]) * np.random.rand(1)
        if noise_lowres is None:
            zl = torch.randn(out.shape, dtype=F32)
        else:
            zl = noise_lowres(tuple(out.shape)) if callable(noise_lowres) else noise_lowres
        out = add_noise(out, nstd, zl)
        nstd = float(torch.tensor(nstd, dtype=F32).item())
    params["noise_std"] = nstd
    st["noisy"] = out

    out = resize_back(out, factors)
    return {"out": out, "seg": seg, "scaled": scale01(out), "params": params, "stages": st if keep_stages else None,
            "image": image_def}



def _run_sample_injected(cfg: Config, segmentation, seed_volumes, draws, noise_gmm, noise_lowres, keep_stages):
    """`run_sample` with every draw supplied (see its `draws` argument): the same stage functions in the same order."""
    st = {}
    m2s = {int(m): int(n) for m, n in draws["m2s"].items()}
    seeds = None
    for m in sorted(m2s):
        v = torch.as_tensor(np.asarray(seed_volumes[m2s[m]][m])).clone()
        seeds = v if seeds is None else seeds + v
    seeds = seeds.long()
    mus, sigmas = draws["mus"], draws["sigmas"]
    z = noise_gmm(tuple(seeds.shape)) if callable(noise_gmm) else noise_gmm
    out = gmm_image(seeds, mus, sigmas, z)
    params = {"mlabel2subclusters": m2s, "mus": mus, "sigmas": sigmas}
    st["gmm"] = out

    shape = tuple(out.shape)
    seg, coords, flip = segmentation, None, False
    dd = draws.get("deform")
    if dd is not None:
        flip = bool(dd["flip"])
        field = nonlinear_field(dd["f_small"], shape) if dd.get("f_small") is not None else None
        ii, jj, kk, margins = deformation_coords(shape, cfg.size, dd["A"], dd["c2"], field)
        coords = (ii, jj, kk)
        params.update(A=dd["A"], c2=dd["c2"], margins=margins)
    params["flip"] = flip
    out, seg = apply_deformation(out, seg, coords, flip)
    st["warped"] = out

    gamma = draws.get("gamma")
    if gamma is not None:
        out = gamma_transform(out, float(gamma))
    params["gamma"] = gamma
    st["gamma"] = out
    if draws.get("bias") is not None:
        out = bias_multiply(out, draws["bias"])
    st["bias"] = out

    factors, rs = None, draws.get("resample")
    if rs is not None:
        spacing = np.array([1.0, 1.0, 1.0]) * float(rs["spacing"])
        out, factors = resample_down(out, cfg.resolution, spacing, float(rs["u_std"]))
    st["resampled"] = out
    nstd = draws.get("noise_std")
    if nstd is not None:
        zl = noise_lowres(tuple(out.shape)) if callable(noise_lowres) else noise_lowres
        out = add_noise(out, np.array([nstd], dtype=np.float64), zl)
    st["noisy"] = out
    out = resize_back(out, factors)
    return {"out": out, "seg": seg, "scaled": scale01(out), "params": params, "stages": st if keep_stages else None,
            "image": None}
