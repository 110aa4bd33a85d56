"""Deterministic synthetic label phantoms (no files, no RNG state touched).

Stand-in for the reference's bundled sample data (`/root/reference/data/sub-sta*`:
a FeTA-style 0..7 tissue segmentation plus per-meta-label sub-cluster "seed" volumes,
see `scripts/generate_seeds.py` and `rand_gmm.py:51-99` in the reference) for tests,
golden-vector generation and `bench.py`.  Nothing here mirrors reference code: the
phantom is nested ellipsoids with a quantised smooth field splitting every meta-label
into sub-clusters.

A seed volume for (n_sub, mlabel) holds values `mlabel*10 + j` (j < n_sub) inside the
meta-label's region and 0 elsewhere, so that summing the four meta-label volumes gives
the seed label map with values in {0, 10..19, 20..29, 30..39, 40..49}.
"""

from __future__ import annotations

import numpy as np

META_LABELS = 4
MAX_SUBCLUSTERS = 6


def _unit_grid(shape):
    axes = [(np.arange(n, dtype=np.float64) - (n - 1) / 2.0) / (n / 2.0) for n in shape]
    return np.meshgrid(*axes, indexing="ij", sparse=True)


def _ellipsoid(g, centre, radii):
    x, y, z = g
    return (
        ((x - centre[0]) / radii[0]) ** 2
        + ((y - centre[1]) / radii[1]) ** 2
        + ((z - centre[2]) / radii[2]) ** 2
    )


def make_segmentation(shape, variant: int = 0):
    """(segmentation uint8 0..7, meta uint8 0..4).

    `variant` perturbs the ellipsoid radii/centres deterministically so a batch of
    distinct volumes can be produced (BASELINE config 3).
    """
    rs = np.random.RandomState(1234 + int(variant))  # local generator, global RNG untouched
    j = rs.uniform(-0.04, 0.04, size=16) if variant else np.zeros(16)
    g = _unit_grid(shape)
    seg = np.zeros(shape, dtype=np.uint8)
    meta = np.zeros(shape, dtype=np.uint8)

    head = _ellipsoid(g, (j[0], j[1], j[2]), (0.80 + j[3], 0.72 + j[4], 0.66 + j[5]))
    brain = _ellipsoid(g, (j[0], j[1], j[2]), (0.64 + j[6], 0.58 + j[7], 0.52 + j[8]))
    gm_in = _ellipsoid(g, (j[0], j[1], j[2]), (0.56 + j[6], 0.50 + j[7], 0.44 + j[8]))
    wm_in = _ellipsoid(g, (j[0], j[1], j[2]), (0.30 + j[9], 0.26 + j[10], 0.22 + j[11]))
    vent_l = _ellipsoid(g, (j[0] - 0.12, j[1], j[2] + 0.04), (0.07, 0.16, 0.06))
    vent_r = _ellipsoid(g, (j[0] + 0.12, j[1], j[2] + 0.04), (0.07, 0.16, 0.06))
    cereb = _ellipsoid(g, (j[0], j[1] - 0.34, j[2] - 0.30), (0.22, 0.14, 0.12))
    stem = _ellipsoid(g, (j[0], j[1] - 0.16, j[2] - 0.34), (0.07, 0.07, 0.16))

    meta[head <= 1.0] = 4  # extra-cerebral shell (no segmentation label)
    seg[brain <= 1.0] = 1
    meta[brain <= 1.0] = 1  # external CSF
    seg[gm_in <= 1.0] = 2
    meta[gm_in <= 1.0] = 2  # cortical GM
    inner = _ellipsoid(g, (j[0], j[1], j[2]), (0.50 + j[6], 0.44 + j[7], 0.38 + j[8])) <= 1.0
    seg[inner] = 3
    meta[inner] = 3  # WM
    seg[wm_in <= 1.0] = 6
    meta[wm_in <= 1.0] = 2  # deep GM
    v = (vent_l <= 1.0) | (vent_r <= 1.0)
    seg[v] = 4
    meta[v] = 1  # ventricles (CSF)
    cb = (cereb <= 1.0) & (brain <= 1.0)
    seg[cb] = 5
    meta[cb] = 3
    st = (stem <= 1.0) & (brain <= 1.0)
    seg[st] = 7
    meta[st] = 3
    return seg, meta


def _subcluster_field(shape, mlabel):
    x, y, z = _unit_grid(shape)
    f = (
        np.sin(3.1 * x + 0.7 * mlabel)
        + np.sin(2.3 * y - 1.1 * mlabel)
        + np.sin(2.9 * z + 0.3 * mlabel)
    )
    return (f + 3.0) / 6.0  # in [0, 1]


def make_seed_volumes(shape, variant: int = 0):
    """Return (segmentation f32 (H,W,D), seeds) where
    `seeds[n_sub][mlabel]` is an int8 array like the reference's
    `subclasses_{n_sub}/..._mlabel_{mlabel}.nii.gz` files."""
    seg, meta = make_segmentation(shape, variant)
    seeds = {}
    fields = {m: _subcluster_field(shape, m) for m in range(1, META_LABELS + 1)}
    for n_sub in range(1, MAX_SUBCLUSTERS + 1):
        seeds[n_sub] = {}
        for m in range(1, META_LABELS + 1):
            sub = np.minimum((fields[m] * n_sub).astype(np.int64), n_sub - 1)
            vol = np.where(meta == m, m * 10 + sub, 0).astype(np.int8)
            seeds[n_sub][m] = vol
    return seg.astype(np.float32), seeds


def combined_seed_labels(seeds, mlabel2subclusters):
    """Sum of the four selected meta-label volumes as uint8 (values 0, 10..49)."""
    out = None
    for m in range(1, META_LABELS + 1):
        v = seeds[mlabel2subclusters[m]][m].astype(np.int16)
        out = v if out is None else out + v
    return out.astype(np.uint8)
