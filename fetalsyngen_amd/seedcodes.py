"""A subject's seed volumes as ONE uint16 code volume (`fsg_sample_head_codes_f32`, include/fsg_hip.h).

The reference adds up four label volumes per sample -- the sub-cluster map chosen for each meta label (`rand_gmm.py:91-99`) --
so the GMM draw reads 4 bytes of labels per voxel.  Per subject the stacked seed volumes (every sub-cluster count x every meta
label) take few distinct columns (a voxel's meta label and its sub-cluster under each count: tens, not thousands); the column
index of every voxel is a 2-byte code, and a sample's label is a look-up of that code in a table the draw's workgroups build
from the `tuples` rows.  Same labels, 6 instead of 8 bytes per voxel in the head kernel.
"""
from __future__ import annotations

import torch

CODES_MAX = 2048  # FSG_CODES_MAX


def build_device(parts, stride: int):
    """`build` for CUDA volumes through the library's one-pass kernel (`fsg_seed_codes_build`, csrc/fsg_codes.hip: an exact hash
    set of columns): ~1 ms per 256^3 subject (40 ms for the torch formulation below with its 24 sorts).  Synchronises once (the number
    of distinct columns decides whether the codes are usable)."""
    import ctypes as C

    from . import _lib
    from . import kernels as K

    if not parts or stride <= len(parts) or stride > 256 or len(parts) > 64:
        raise ValueError("seed codes: need 0 < len(parts) < stride <= 256 and at most 64 volumes")
    shape, dev = parts[0].shape, parts[0].device
    for p in parts:
        if p.dtype != torch.uint8 or p.shape != shape or not p.is_contiguous() or p.device != dev or not p.is_cuda:
            raise ValueError("seed codes: contiguous uint8 CUDA volumes of one shape on one device")
    lib = _lib.load()
    n = parts[0].numel()
    codes = torch.empty(shape, dtype=torch.int16, device=dev)
    tuples = torch.empty((CODES_MAX, stride), dtype=torch.uint8, device=dev)
    work = torch.empty(int(lib.fsg_seed_codes_work_bytes()), dtype=torch.uint8, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ptrs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
    _lib.check(lib.fsg_seed_codes_build(ptrs, len(parts), n, stride, codes.data_ptr(), tuples.data_ptr(), CODES_MAX, work.data_ptr(),
                                        work.numel(), count.data_ptr(), K._stream(parts[0])), "fsg_seed_codes_build")
    ntup = int(count.item())  # synchronises: the kernel has finished, `work` may go
    if ntup > CODES_MAX:
        return None
    return codes, tuples[:max(ntup, 1)].clone()


def build(parts, stride: int):
    """parts: uint8 device tensors of one shape, column j of a tuple row = parts[j]; stride > len(parts) (the last byte of a
    row stays 0: the column an absent meta label selects).  Returns (codes int16 of the volumes' shape, tuples uint8
    (T, stride)) or None when the subject has more than CODES_MAX distinct columns.  (The torch formulation: tests, CPU.)"""
    if not parts or stride <= len(parts) or stride > 256:
        raise ValueError("seed codes: need 0 < len(parts) < stride <= 256")
    shape = parts[0].shape
    flat = [p.reshape(-1) for p in parts]
    code = torch.zeros(flat[0].numel(), dtype=torch.int64, device=flat[0].device)
    for v in flat:  # refine the partition volume by volume: codes stay < CODES_MAX, keys < CODES_MAX * 256
        if v.dtype != torch.uint8 or v.numel() != code.numel():
            raise ValueError("seed codes: uint8 volumes of one shape")
        uniq, code = torch.unique(code * 256 + v.to(torch.int64), return_inverse=True)
        if uniq.numel() > CODES_MAX:
            return None
    ntup = int(code.max()) + 1 if code.numel() else 0
    tuples = torch.zeros((max(ntup, 1), stride), dtype=torch.uint8, device=code.device)
    for j, v in enumerate(flat):  # every voxel of a code holds the same value: any of them writes the row's byte
        tuples[:, j].scatter_(0, code, v)
    return code.to(torch.int16).reshape(shape).contiguous(), tuples.contiguous()


def labels_of(codes, tuples, sel):
    """The label volume a selection (byte indices into a row, one per meta label) stands for -- what the kernel looks up."""
    t = tuples.to(torch.int64)
    lab = sum(t[:, int(s)] for s in sel) & 255
    return lab[codes.reshape(-1).to(torch.int64)].reshape(codes.shape).to(torch.uint8)
