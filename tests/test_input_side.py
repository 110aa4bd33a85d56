"""Input side of the path: NIfTI reader + RAS reorientation.

Pins:
  * tests/golden/inputs_sta21.npz -- header fields, label histogram, voxel sum and a decimated copy of two of the
    reference's bundled volumes (`sub-sta21` dseg, one seed file), written by tests/golden/make_golden.py with
    its own independent reader.  Where /root/reference is present (the build container) `NiftiReader` reads
    the very files and must reproduce the fixture.
  * hand-built NIfTI-1 headers (LPS-coded, permuted axes, qform-only with qfac = -1, oblique) for
    `io_orientation` / `ras_reorient`, with the expected result written out by hand.
SimpleITK / monai / nibabel are not importable here: anything beyond these pins is "parity unpinned"
(reference utils/image_reading.py:32-55, data/datasets.py:280-284).
"""
import gzip
import itertools
import struct
from pathlib import Path

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from fetalsyngen_amd.utils.image_reading import NiftiReader, io_orientation, ras_reorient, read_nifti

GOLD = Path(__file__).parent / "golden"
REF = Path("/root/reference")


def _write(path, arr, *, sform=None, qform=None, pixdim=(1.0, 1.0, 1.0), qfac=1.0, gz=True):
    """Hand-built NIfTI-1 single file.  sform: (3,4) or None; qform: (b,c,d,ox,oy,oz) or None."""
    arr = np.asarray(arr)
    code, bits = {np.dtype(np.float32): (16, 32), np.dtype(np.uint8): (2, 8), np.dtype(np.int16): (4, 16)}[arr.dtype]
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, *arr.shape, 1, 1, 1, 1)
    struct.pack_into("<h", hdr, 70, code)
    struct.pack_into("<h", hdr, 72, bits)
    struct.pack_into("<8f", hdr, 76, qfac, *pixdim, 1, 1, 1, 1)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    if qform is not None:
        struct.pack_into("<h", hdr, 252, 1)
        struct.pack_into("<6f", hdr, 256, *qform)
    if sform is not None:
        struct.pack_into("<h", hdr, 254, 1)
        struct.pack_into("<12f", hdr, 280, *np.asarray(sform, dtype=np.float32).reshape(-1))
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + np.asfortranarray(arr).tobytes(order="F")
    with (gzip.open(path, "wb", compresslevel=1) if gz else open(path, "wb")) as fh:
        fh.write(payload)


def _vol(shape=(5, 4, 3)):
    return np.arange(int(np.prod(shape)), dtype=np.float32).reshape(shape)


# ---------------------------------------------------------------------------------------------------
def test_fixture_of_the_reference_bundled_files():
    g = np.load(GOLD / "inputs_sta21.npz")
    for key in ("dseg", "seed"):
        assert tuple(g[f"{key}_dim"][1:4]) == (256, 256, 256)
        assert tuple(g[f"{key}_codes"]) == (1, 1)
        srow = g[f"{key}_srow"]
        assert np.all(np.diag(srow[:, :3]) == 0.5) and np.count_nonzero(srow[:, :3]) == 3  # already RAS
        assert int(g[f"{key}_hist"].sum()) == 256 ** 3
    assert set(np.nonzero(g["dseg_hist"])[0]) <= set(range(8))
    assert set(np.nonzero(g["seed_hist"])[0]) <= {0} | set(range(20, 30))  # meta label 2 -> sub-clusters 20..29


@pytest.mark.skipif(not REF.exists(), reason="the reference's bundled data only exists in the build container")
def test_reader_reproduces_the_bundled_files():
    g = np.load(GOLD / "inputs_sta21.npz")
    for key in ("dseg", "seed"):
        path = REF / str(g[f"{key}_relpath"])
        arr, affine, pixdim = read_nifti(path)
        np.testing.assert_array_equal(affine[:3, :], g[f"{key}_srow"].astype(np.float64))
        assert tuple(pixdim) == (0.5, 0.5, 0.5)
        t = NiftiReader()(path)
        assert tuple(t.shape) == (256, 256, 256)
        v = t.numpy()
        assert float(v.astype(np.float64).sum()) == float(g[f"{key}_sum"])
        np.testing.assert_array_equal(np.bincount(v.astype(np.int64).reshape(-1), minlength=64), g[f"{key}_hist"])
        np.testing.assert_array_equal(v[::4, ::4, ::4].astype(np.uint8), g[f"{key}_sub4"])
        np.testing.assert_array_equal(v[128].astype(np.uint8), g[f"{key}_centre_x"])


# ---------------------------------------------------------------------------------------------------
def test_ras_file_is_left_alone(tmp_path):
    v = _vol()
    _write(tmp_path / "a.nii.gz", v, sform=[[0.5, 0, 0, -3], [0, 0.5, 0, -4], [0, 0, 0.5, -5]])
    np.testing.assert_array_equal(NiftiReader()(tmp_path / "a.nii.gz").numpy(), v)


def test_lps_coded_file_is_flipped_on_x_and_y(tmp_path):
    v = _vol()
    _write(tmp_path / "a.nii.gz", v, sform=[[-0.5, 0, 0, 3], [0, -0.5, 0, 4], [0, 0, 0.5, -5]])
    np.testing.assert_array_equal(NiftiReader()(tmp_path / "a.nii.gz").numpy(), v[::-1, ::-1, :])


def test_permuted_axes(tmp_path):
    # voxel axis 0 runs Superior, axis 1 runs Right, axis 2 runs Posterior
    v = _vol((5, 4, 3))
    sform = [[0, 1.0, 0, 0], [0, 0, -1.0, 0], [1.0, 0, 0, 0]]
    _write(tmp_path / "a.nii", v, sform=sform, gz=False)
    out = NiftiReader()(tmp_path / "a.nii").numpy()
    assert out.shape == (4, 3, 5)  # (R, A, S) <- voxel axes (1, 2 flipped, 0)
    np.testing.assert_array_equal(out, v[:, :, ::-1].transpose(1, 2, 0))
    np.testing.assert_array_equal(io_orientation(np.vstack([sform, [0, 0, 0, 1]])), [[2, 1], [0, 1], [1, -1]])


def test_qform_only_with_negative_qfac(tmp_path):
    # identity quaternion, qfac = -1: the third voxel axis runs Inferior -> flipped to run Superior
    v = _vol()
    _write(tmp_path / "a.nii.gz", v, qform=(0, 0, 0, 1, 2, 3), qfac=-1.0, pixdim=(0.5, 0.5, 0.5))
    arr, affine, _ = read_nifti(tmp_path / "a.nii.gz")
    np.testing.assert_allclose(affine, [[0.5, 0, 0, 1], [0, 0.5, 0, 2], [0, 0, -0.5, 3], [0, 0, 0, 1]])
    np.testing.assert_array_equal(NiftiReader()(tmp_path / "a.nii.gz").numpy(), v[:, :, ::-1])


def test_qform_only_half_turn_about_z(tmp_path):
    # quaternion (a,b,c,d) = (0,0,0,1): rotation by 180 degrees about z = an LPS-coded file
    v = _vol()
    _write(tmp_path / "a.nii.gz", v, qform=(0, 0, 1.0, 0, 0, 0))
    _arr, affine, _ = read_nifti(tmp_path / "a.nii.gz")
    np.testing.assert_allclose(affine[:3, :3], np.diag([-1.0, -1.0, 1.0]), atol=1e-12)
    np.testing.assert_array_equal(NiftiReader()(tmp_path / "a.nii.gz").numpy(), v[::-1, ::-1, :])


def test_sform_wins_over_qform(tmp_path):
    v = _vol()
    _write(tmp_path / "a.nii.gz", v, sform=[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], qform=(0, 0, 1.0, 0, 0, 0))
    np.testing.assert_array_equal(NiftiReader()(tmp_path / "a.nii.gz").numpy(), v)


def test_oblique_affine_picks_the_closest_canonical_axes(tmp_path):
    # 25 degrees about z on top of an LPS coding: still "x -> Left, y -> Posterior"; used to raise
    c, s = np.cos(np.deg2rad(25)), np.sin(np.deg2rad(25))
    rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ np.diag([-0.8, -0.8, 2.0])
    sform = np.hstack([rot, [[1], [2], [3]]])
    v = _vol()
    _write(tmp_path / "a.nii.gz", v, sform=sform)
    np.testing.assert_array_equal(io_orientation(np.vstack([sform, [0, 0, 0, 1]])), [[0, -1], [1, -1], [2, 1]])
    np.testing.assert_array_equal(NiftiReader()(tmp_path / "a.nii.gz").numpy(), v[::-1, ::-1, :])


def test_near_45_degree_tie_is_resolved_without_reusing_a_world_axis():
    # both in-plane voxel axes lean on R more than on A: the first takes R, the second must take A
    A = np.eye(4)
    A[:3, :3] = [[0.80, 0.75, 0], [0.60, -0.66, 0], [0, 0, 1]]
    o = io_orientation(A)
    assert sorted(o[:, 0]) == [0, 1, 2] and tuple(o[0]) == (0, 1) and tuple(o[1]) == (1, -1)


@settings(max_examples=60, deadline=None)
@given(perm=st.permutations([0, 1, 2]), signs=st.tuples(*[st.sampled_from([-1, 1])] * 3),
       zooms=st.tuples(*[st.floats(0.2, 3.0)] * 3), tilt=st.floats(-0.3, 0.3))
def test_signed_permutations_property(perm, signs, zooms, tilt):
    """For any signed axis permutation (optionally tilted by < 0.3 rad about a world axis) the reoriented
    array satisfies out[r, a, s] == in[voxel index that world (r,a,s) maps to]."""
    v = _vol((4, 3, 2))
    M = np.zeros((3, 3))
    for in_ax in range(3):
        M[perm[in_ax], in_ax] = signs[in_ax] * zooms[in_ax]
    c, s = np.cos(tilt), np.sin(tilt)
    M = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ M
    A = np.eye(4)
    A[:3, :3] = M
    out = ras_reorient(v, A)
    inv = np.argsort(perm)  # world axis -> voxel axis
    assert out.shape == tuple(v.shape[inv[w]] for w in range(3))
    for idx in itertools.product(*[range(n) for n in out.shape]):
        src = [0, 0, 0]
        for w in range(3):
            ax = inv[w]
            src[ax] = idx[w] if signs[ax] > 0 else v.shape[ax] - 1 - idx[w]
        assert out[idx] == v[tuple(src)]


def test_big_endian_and_unknown_datatype_raise(tmp_path):
    p = tmp_path / "be.nii"
    p.write_bytes(struct.pack(">i", 348) + bytes(348))
    with pytest.raises(ValueError):
        read_nifti(p)
    v = _vol()
    _write(tmp_path / "a.nii", v, sform=np.eye(4)[:3], gz=False)
    raw = bytearray((tmp_path / "a.nii").read_bytes())
    struct.pack_into("<h", raw, 70, 1792)  # complex128
    (tmp_path / "b.nii").write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        read_nifti(tmp_path / "b.nii")
