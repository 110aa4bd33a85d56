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


def read_nifti(path):
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
    sform_code = struct.unpack("<h", raw[254:256])[0]
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
    else:
        affine[0, 0], affine[1, 1], affine[2, 2] = pixdim[1:4]
    return np.ascontiguousarray(arr), affine, pixdim[1:4]


def ras_reorient(arr: np.ndarray, affine: np.ndarray):
    """Permute/flip axes so that the voxel axes run R, A, S (closest-canonical)."""
    rot = affine[:3, :3]
    order = [int(np.argmax(np.abs(rot[r, :]))) for r in range(3)]
    if sorted(order) != [0, 1, 2]:
        raise ValueError("oblique affine: cannot pick a closest canonical orientation")
    out = arr.transpose(order)
    for r in range(3):
        if rot[r, order[r]] < 0:
            out = np.flip(out, axis=r)
    return np.ascontiguousarray(out)


class NiftiReader:
    """`reader(path) -> torch.Tensor` (x,y,z), oriented to RAS."""

    def __call__(self, img_path: str | Path, as_meta: bool = True) -> torch.Tensor:
        arr, affine, _ = read_nifti(img_path)
        return torch.from_numpy(ras_reorient(arr, affine))


# name used by the reference's call sites
SimpleITKReader = NiftiReader
