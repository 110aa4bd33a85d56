"""CPU: the C-ABI library loads and exports every symbol include/fsg_hip.h declares; argument
validation happens before any launch (no GPU needed for these calls)."""
import ctypes
import re
from pathlib import Path

import pytest

from fetalsyngen_amd import _lib

REPO = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (REPO / "include" / "fsg_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fsg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/fsg_hip.h but not exported"
    bound = set(_lib.SIGNATURES) | set(_lib.SPECIAL_RESTYPE)
    assert set(syms) == bound, f"ctypes table and header disagree: {set(syms) ^ bound}"


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.fsg_abi_version() == _lib.ABI_VERSION
    assert b"bad argument" in lib.fsg_error_string(-1)
    assert lib.fsg_error_string(0) == b"success"


def test_key_roundtrip_orders_floats():
    lib = _lib.load()
    import numpy as np

    vals = np.array([-np.inf, -3.5, -0.0, 0.0, 1e-30, 2.0, 255.0, np.inf], dtype=np.float32)
    bits = vals.view(np.int32)
    keys = np.where(bits >= 0, bits, bits ^ 0x7FFFFFFF)
    assert (np.diff(keys.astype(np.int64)) > 0).all()
    for v, k in zip(vals, keys):
        assert lib.fsg_key_to_float(int(k)) == v


def test_bad_arguments_are_rejected_before_launch():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.fsg_randn_f32(null, 16, 0, 0, null) == _lib.E_BADARG
    assert lib.fsg_gamma_f32(null, 16, 1.0, null, null) == _lib.E_BADARG
    assert lib.fsg_blur_axis_f32(null, null, 4, 4, 4, 0, null, 3, null) == _lib.E_BADARG
    one = ctypes.c_void_p(16)
    two = ctypes.c_void_p(32)
    assert lib.fsg_blur_axis_f32(one, two, 4, 4, 4, 3, one, 3, null) == _lib.E_BADARG  # axis
    assert lib.fsg_blur_axis_f32(one, two, 4, 4, 4, 0, one, 4, null) == _lib.E_BADARG  # even taps
    assert lib.fsg_blur_axis_f32(one, one, 4, 4, 4, 0, one, 3, null) == _lib.E_BADARG  # in place
    assert lib.fsg_zoom3d_f32(one, 0, 4, 4, 1, one, one, one, two, 4, 4, 4, null) == _lib.E_BADARG
    assert lib.fsg_interp3d_f32(one, 4, 4, 4, one, one, one, 8, 2, 0.0, two, null) == _lib.E_BADARG  # mode
    d = _lib.Deform()
    assert lib.fsg_coords_minmax_f32(ctypes.byref(d), one, null) == _lib.E_BADARG  # zero shape
    d.shape[:] = [2048, 2048, 2048]
    assert lib.fsg_coords_minmax_f32(ctypes.byref(d), one, null) == _lib.E_TOOBIG
    with pytest.raises(_lib.FsgError):
        _lib.check(-1, "x")


def test_product_refuses_cpu_tensors():
    import torch

    from fetalsyngen_amd import kernels as K

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.gamma(torch.zeros(2, 2, 2), 1.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.blur_axis(torch.zeros(4, 4, 4), 0, [0.25, 0.5, 0.25])


def test_sample_plan_mirror_has_the_c_layout():
    """The ctypes mirror of fsg_sample_plan against the compiled struct: total size and the offsets of a field behind the
    large tap array, of the last r01 field, and of the two newest ones (a mirror that drifts would hand every later
    pointer to the wrong slot)."""
    import ctypes as C

    from fetalsyngen_amd import _lib

    lib = _lib.load()
    P = _lib.SamplePlan
    assert lib.fsg_sample_plan_layout(0) == C.sizeof(P)
    assert lib.fsg_sample_plan_layout(1) == P.blur_taps.offset
    assert lib.fsg_sample_plan_layout(2) == P.out.offset
    assert lib.fsg_sample_plan_layout(3) == P.seg_in_u8.offset
    assert lib.fsg_sample_plan_layout(4) == P.ws_seq.offset
    assert lib.fsg_sample_plan_layout(5) == P.code_sel.offset
    assert lib.fsg_sample_plan_layout(99) == -1
