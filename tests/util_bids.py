"""Writes a tiny BIDS tree (segmentation + seed volumes as .nii.gz) from the synthetic phantom."""
import gzip
import struct

import numpy as np

from fetalsyngen_amd.phantom import make_seed_volumes

_CODES = {np.dtype(np.float32): (16, 32), np.dtype(np.int8): (256, 8), np.dtype(np.uint8): (2, 8)}


def write_nifti(path, arr, voxel=0.5):
    arr = np.asarray(arr)
    code, bits = _CODES[arr.dtype]
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, *arr.shape, 1, 1, 1, 1)
    struct.pack_into("<h", hdr, 70, code)
    struct.pack_into("<h", hdr, 72, bits)
    struct.pack_into("<8f", hdr, 76, 1.0, voxel, voxel, voxel, 1, 1, 1, 1)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    struct.pack_into("<h", hdr, 254, 1)  # sform_code
    struct.pack_into("<12f", hdr, 280, voxel, 0, 0, -10, 0, voxel, 0, -10, 0, 0, voxel, -10)
    hdr[344:348] = b"n+1\0"
    path.parent.mkdir(parents=True, exist_ok=True)
    with gzip.open(path, "wb", compresslevel=1) as fh:
        fh.write(bytes(hdr))
        fh.write(np.asfortranarray(arr).tobytes(order="F"))


def write_tree(root, shape, subjects):
    bids, seeds_root = root / "bids", root / "seeds"
    for vi, sub in enumerate(subjects):
        seg, seeds = make_seed_volumes(shape, vi)
        write_nifti(bids / sub / "anat" / f"{sub}_rec-x_T2w.nii.gz", (seg * 30).astype(np.float32))
        write_nifti(bids / sub / "anat" / f"{sub}_rec-x_T2w_dseg.nii.gz", seg.astype(np.float32))
        for n_sub, d in seeds.items():
            for m, vol in d.items():
                write_nifti(seeds_root / f"subclasses_{n_sub}" / sub / "anat" / f"{sub}_rec-x_T2w_dseg_mlabel_{m}.nii.gz",
                            vol)
    return bids, seeds_root
