"""Property tests for the sampler edge cases: coordinates exactly 0, exactly n-1, k+0.5 (even and
odd k), 1 ulp either side of those, out-of-range values -- `fast_3D_interp_torch` semantics, reference
utils/generation.py:204-288.

CPU part: the vectorised oracle against a scalar, voxel-by-voxel statement of the same rules (its own
transcription of the reference lines, so a slip in the oracle's masking / indexing shows).
GPU part: the HIP gather (`fsg_interp3d_f32` through `fast_3D_interp_torch`) against the oracle, bit for bit.
"""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import fsg_oracle as O

F = np.float32


def _edge_values(n):
    """Coordinates worth probing on an axis of length n (float32)."""
    vals = [0.0, n - 1.0, -0.0, -1.0, -0.25, n - 0.5, float(n), n + 3.0, 1e-30, 1e-45]
    for k in range(0, n):
        vals += [k + 0.5, float(k)]
    out = []
    for v in vals:
        v = F(v)
        out += [v, np.nextafter(v, F(np.inf), dtype=F), np.nextafter(v, F(-np.inf), dtype=F)]
    return np.array(out, dtype=F)


def _coords(draw, shape, npts):
    cols = []
    for n in shape:
        pool = _edge_values(n)
        pick = draw(st.lists(st.integers(0, len(pool) - 1), min_size=npts, max_size=npts))
        frac = draw(st.lists(st.floats(-1.5, n + 0.5, width=32), min_size=npts, max_size=npts))
        use_edge = draw(st.lists(st.booleans(), min_size=npts, max_size=npts))
        cols.append(np.where(use_edge, pool[pick], np.array(frac, dtype=F)).astype(F))
    return cols


@st.composite
def cases(draw):
    shape = tuple(draw(st.integers(1, 6)) for _ in range(3))
    npts = draw(st.integers(1, 24))
    seed = draw(st.integers(0, 2**31 - 1))
    vol = np.random.RandomState(seed).uniform(-50, 250, shape).astype(F)
    return vol, _coords(draw, shape, npts)


def _scalar_linear(x, ii, jj, kk, default=0.0):
    """One voxel at a time, float32 arithmetic in the reference's order (generation.py:227-285)."""
    n0, n1, n2 = x.shape
    out = np.full(ii.shape, default, dtype=F)
    for p in range(ii.size):
        i, j, k = F(ii[p]), F(jj[p]), F(kk[p])
        if not (i > 0 and j > 0 and k > 0 and i <= n0 - 1 and j <= n1 - 1 and k <= n2 - 1):
            continue
        fx, fy, fz = int(np.floor(i)), int(np.floor(j)), int(np.floor(k))
        cx, cy, cz = min(fx + 1, n0 - 1), min(fy + 1, n1 - 1), min(fz + 1, n2 - 1)
        wcx, wcy, wcz = F(i - F(fx)), F(j - F(fy)), F(k - F(fz))
        wfx, wfy, wfz = F(F(1) - wcx), F(F(1) - wcy), F(F(1) - wcz)
        c00 = F(F(x[fx, fy, fz] * wfx) + F(x[cx, fy, fz] * wcx))
        c01 = F(F(x[fx, fy, cz] * wfx) + F(x[cx, fy, cz] * wcx))
        c10 = F(F(x[fx, cy, fz] * wfx) + F(x[cx, cy, fz] * wcx))
        c11 = F(F(x[fx, cy, cz] * wfx) + F(x[cx, cy, cz] * wcx))
        c0 = F(F(c00 * wfy) + F(c10 * wcy))
        c1 = F(F(c01 * wfy) + F(c11 * wcy))
        out[p] = F(F(c0 * wfz) + F(c1 * wcz))
    return out


def _scalar_nearest(x, ii, jj, kk):
    """round-half-even -> clamp -> gather (generation.py:211-225); no validity mask."""
    out = np.empty(ii.shape, dtype=F)
    for p in range(ii.size):
        idx = []
        for c, n in zip((ii[p], jj[p], kk[p]), x.shape):
            r = np.rint(F(c))  # IEEE round-half-even, like torch.round
            idx.append(int(min(max(r, 0), n - 1)))
        out[p] = x[tuple(idx)]
    return out


@settings(max_examples=150, deadline=None)
@given(case=cases())
def test_oracle_sampler_equals_the_scalar_statement(case):
    vol, (ii, jj, kk) = case
    x = torch.from_numpy(vol)
    ti, tj, tk = (torch.from_numpy(c) for c in (ii, jj, kk))
    assert np.array_equal(O.sample_linear(x, ti, tj, tk).numpy(), _scalar_linear(vol, ii, jj, kk))
    assert np.array_equal(O.sample_nearest(x, ti, tj, tk).numpy(), _scalar_nearest(vol, ii, jj, kk))


def test_named_edge_cases():
    """The cases SURVEY 7.3-7 spells out, as plain assertions on the oracle."""
    x = torch.arange(4 * 3 * 5, dtype=torch.float32).reshape(4, 3, 5) + 1
    c = lambda *v: torch.tensor(v, dtype=torch.float32)
    # a coordinate exactly 0 on any axis -> default (strict > 0); exactly n-1 on every axis is valid
    assert float(O.sample_linear(x, c(0.0), c(1.0), c(1.0))) == 0.0
    assert float(O.sample_linear(x, c(1.0), c(0.0), c(1.0))) == 0.0
    assert float(O.sample_linear(x, c(3.0), c(2.0), c(4.0))) == float(x[3, 2, 4])
    assert float(O.sample_linear(x, c(np.nextafter(F(3.0), F(4.0))), c(2.0), c(4.0))) == 0.0
    tiny = float(np.nextafter(F(0), F(1)))
    assert float(O.sample_linear(x, c(tiny), c(tiny), c(tiny))) > 0.0  # the smallest positive coordinate is valid
    # nearest: half-to-even, clamped, no mask
    assert float(O.sample_nearest(x, c(0.5), c(0.0), c(0.0))) == float(x[0, 0, 0])   # 0.5 -> 0
    assert float(O.sample_nearest(x, c(1.5), c(0.0), c(0.0))) == float(x[2, 0, 0])   # 1.5 -> 2
    assert float(O.sample_nearest(x, c(2.5), c(0.0), c(0.0))) == float(x[2, 0, 0])   # 2.5 -> 2
    assert float(O.sample_nearest(x, c(-7.0), c(9.0), c(4.49))) == float(x[0, 2, 4])


@pytest.mark.gpu
@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(case=cases())
def test_hip_sampler_equals_oracle_on_edge_coordinates(case):
    if not torch.cuda.is_available():
        pytest.fail("GPU test without a GPU")
    from fetalsyngen_amd.utils.generation import fast_3D_interp_torch

    vol, (ii, jj, kk) = case
    x = torch.from_numpy(vol)
    ti, tj, tk = (torch.from_numpy(c) for c in (ii, jj, kk))
    d = lambda a: a.to("cuda:0")
    lin = fast_3D_interp_torch(d(x), d(ti), d(tj), d(tk), "linear").cpu()
    nn = fast_3D_interp_torch(d(x), d(ti), d(tj), d(tk), "nearest").cpu()
    assert torch.equal(lin, O.sample_linear(x, ti, tj, tk))
    assert torch.equal(nn, O.sample_nearest(x, ti, tj, tk))
