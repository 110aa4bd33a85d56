"""Minimal NIfTI-1 reader (gzip + struct) standing in for the reference's `SimpleITKReader`
(fetalsyngen/utils/image_reading.py:8-55), which needs SimpleITK and monai.

Returns the voxel array as a torch tensor indexed (x, y, z) -- what the reference produces after its
`(z,y,x) -> (x,y,z)` permute -- and, on request, the RAS affine from the sform/qform.  File I/O is
outside the hot path; this exists so `FetalSynthDataset` can read the bundled
sample volumes and test fixtures.
"""
from __future__ import annotations

import gzip
import struct
from pathlib import Path

import numpy as np
import torch

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32}


def _qform_affine(raw, pixdim):
    """NIfTI-1 method 2 (quaternion + offsets; the NIfTI-1 standard's nifti_quatern_to_mat44): qfac = pixdim[0]
    (-1 flips the third axis), columns scaled by pixdim[1..3]."""
    b, c, d = struct.unpack("<3f", raw[256:268])
    off = struct.unpack("<3f", raw[268:280])
    b, c, d = float(b), float(c), float(d)
    a2 = 1.0 - (b * b + c * c + d * d)
    if a2 < 1e-7:  # 180 degree rotation: renormalise (b,c,d), a = 0
        nrm = 1.0 / np.sqrt(b * b + c * c + d * d)
        b, c, d, a = b * nrm, c * nrm, d * nrm, 0.0
    else:
        a = float(np.sqrt(a2))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
                  [2 * b * c + 2 * a * d, a * a + c * c - b * b - d * d, 2 * c * d - 2 * a * b],
                  [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a + d * d - c * c - b * b]])
    zooms = np.array([pixdim[1] if pixdim[1] > 0 else 1.0, pixdim[2] if pixdim[2] > 0 else 1.0,
                      pixdim[3] if pixdim[3] > 0 else 1.0], dtype=np.float64)
    qfac = -1.0 if pixdim[0] < 0 else 1.0
    zooms[2] *= qfac
    affine = np.eye(4)
    affine[:3, :3] = R * zooms[None, :]
    affine[:3, 3] = off
    return affine


def read_nifti(path):
    """-> (array indexed (x,y,z[,t]), 4x4 voxel->RAS affine, pixdim[1:4]).

    Affine choice: sform when sform_code > 0, else qform when qform_code > 0, else diag(pixdim) -- nibabel's
    `get_best_affine` order.  (The reference reads through SimpleITK/ITK, which is absent here; when a file
    carries BOTH forms and they disagree, ITK's choice between them is version dependent: parity unpinned for
    such files.  The reference's bundled volumes carry consistent, diagonal forms.)"""
    path = str(path)
    with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as fh:
        raw = fh.read()
    if struct.unpack("<i", raw[0:4])[0] != 348:
        raise ValueError(f"{path}: not a little-endian NIfTI-1 file")
    dim = struct.unpack("<8h", raw[40:56])
    datatype = struct.unpack("<h", raw[70:72])[0]
    pixdim = struct.unpack("<8f", raw[76:108])
    vox_offset = int(struct.unpack("<f", raw[108:112])[0])
    slope, inter = struct.unpack("<2f", raw[112:120])
    qform_code, sform_code = struct.unpack("<2h", raw[252:256])
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    shape = tuple(int(d) for d in dim[1 : 1 + max(dim[0], 3)])
    shape = shape + (1,) * (3 - len(shape))
    n = int(np.prod(shape))
    arr = np.frombuffer(raw, dtype=_DTYPES[datatype], count=n, offset=vox_offset)
    arr = arr.reshape(shape[::-1]).transpose(*range(len(shape) - 1, -1, -1))  # Fortran order on disk
    if slope not in (0.0, 1.0) or inter != 0.0:
        arr = arr.astype(np.float32) * slope + inter
    affine = np.eye(4)
    if sform_code > 0:
        affine[:3, :] = np.array(struct.unpack("<12f", raw[280:328])).reshape(3, 4)
    elif qform_code > 0:
        affine = _qform_affine(raw, pixdim)
    else:
        affine[0, 0], affine[1, 1], affine[2, 2] = [v if v > 0 else 1.0 for v in pixdim[1:4]]
    return np.ascontiguousarray(arr), affine, pixdim[1:4]


def io_orientation(affine: np.ndarray) -> np.ndarray:
    """(3,2) array: for every voxel axis the world axis (0=R,1=A,2=S) it is closest to and the direction (+1/-1).

    Restatement of the published algorithm of nibabel 5.3.2 `orientations.io_orientation` (the routine monai's
    `Orientation("RAS")` -- reference data/datasets.py:280-284 -- calls; pinned in the reference's
    environment.yml:110, source not under /root/reference): normalise the columns, take the closest
    orthogonal matrix (SVD), then walk the voxel axes in order, giving each the not-yet-taken world axis with
    the largest absolute component.  Works for oblique affines."""
    rzs = np.asarray(affine, dtype=np.float64)[:3, :3]
    zooms = np.sqrt(np.sum(rzs * rzs, axis=0))
    zooms[zooms == 0] = 1
    rs = rzs / zooms
    P, S, Qs = np.linalg.svd(rs, full_matrices=False)
    tol = S.max() * 3 * np.finfo(S.dtype).eps
    keep = S > tol
    R = P[:, keep] @ Qs[keep]
    ornt = np.full((3, 2), np.nan)
    for in_ax in range(3):
        col = R[:, in_ax]
        if not np.allclose(col, 0):
            out_ax = int(np.argmax(np.abs(col)))
            ornt[in_ax, 0] = out_ax
            ornt[in_ax, 1] = -1 if col[out_ax] < 0 else 1
            R[out_ax, :] = 0  # this world axis is taken
    if np.isnan(ornt).any():
        raise ValueError("degenerate affine: a voxel axis has no direction")
    return ornt


def ras_reorient(arr: np.ndarray, affine: np.ndarray):
    """Flip / permute the three leading axes so that they run R, A, S (closest canonical, as monai's
    `Orientation("RAS")`)."""
    ornt = io_orientation(affine)
    out = arr
    for in_ax in range(3):
        if ornt[in_ax, 1] < 0:
            out = np.flip(out, axis=in_ax)
    order = [int(v) for v in np.argsort(ornt[:, 0])] + list(range(3, arr.ndim))
    return np.ascontiguousarray(out.transpose(order))


class NiftiReader:
    """`reader(path) -> torch.Tensor` (x,y,z), oriented to RAS."""

    def __call__(self, img_path: str | Path, as_meta: bool = True) -> torch.Tensor:
        arr, affine, _ = read_nifti(img_path)
        return torch.from_numpy(ras_reorient(arr, affine))


# name used by the reference's call sites
SimpleITKReader = NiftiReader
