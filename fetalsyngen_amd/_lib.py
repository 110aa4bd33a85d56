"""ctypes binding of libfsg_hip.so (C ABI in include/fsg_hip.h).

There is NO fallback: if the library is missing or does not load, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("FSG_LIB", PKG / "libfsg_hip.so"))  # FSG_LIB: A/B another build of the same ABI

E_BADARG, E_TOOBIG, E_ALIGN = -1, -2, -3
ABI_VERSION = 3  # include/fsg_hip.h: FSG_ABI_VERSION


class FsgError(RuntimeError):
    def __init__(self, code: int, where: str, msg: str):
        super().__init__(f"{where}: {msg} (code {code})")
        self.code = code


class Tap(C.Structure):
    _fields_ = [("lo", C.c_int32), ("hi", C.c_int32), ("w_lo", C.c_float), ("w_hi", C.c_float)]


class Deform(C.Structure):
    _fields_ = [
        ("shape", C.c_int32 * 3),
        ("A", C.c_float * 9),
        ("centre", C.c_float * 3),
        ("c2", C.c_float * 3),
        ("flip", C.c_int32),
        ("field_dims", C.c_int32 * 3),
        ("field", C.c_void_p),
        ("tx", C.c_void_p),
        ("ty", C.c_void_p),
        ("tz", C.c_void_p),
        ("rows", C.c_void_p),
        ("row_stride", C.c_int32),
    ]


class Epilogue(C.Structure):
    _fields_ = [
        ("gamma", C.c_float),
        ("bias_dims", C.c_int32 * 3),
        ("bias", C.c_void_p),
        ("bx", C.c_void_p),
        ("by", C.c_void_p),
        ("bz", C.c_void_p),
    ]


class SamplePlan(C.Structure):
    _fields_ = [
        ("shape", C.c_int32 * 3),
        ("label_parts", C.c_void_p * 4),
        ("mus", C.c_void_p),
        ("sigmas", C.c_void_p),
        ("ntab", C.c_int32),
        ("gmm_noise", C.c_void_p),
        ("gmm_seed", C.c_uint64),
        ("gmm_stream", C.c_uint64),
        ("deform_active", C.c_int32),
        ("deform", Deform),
        ("seg_in", C.c_void_p),
        ("seg_out", C.c_void_p),
        ("epi", Epilogue),
        ("resample_active", C.c_int32),
        ("low_shape", C.c_int32 * 3),
        ("rs_tab", C.c_void_p * 3),
        ("back_tab", C.c_void_p * 3),
        ("blur_ntaps", C.c_int32 * 3),
        ("blur_taps", (C.c_float * 129) * 3),
        ("noise_mode", C.c_int32),
        ("noise", C.c_void_p),
        ("noise_seed", C.c_uint64),
        ("noise_stream", C.c_uint64),
        ("noise_std", C.c_float),
        ("scale01", C.c_int32),
        ("ws0", C.c_void_p),
        ("ws1", C.c_void_p),
        ("ws_low", C.c_void_p),
        ("ws_rows", C.c_void_p),
        ("row_stride", C.c_int32),
        ("mm8", C.c_void_p),
        ("mm8_preset", C.c_int32),
        ("out", C.c_void_p),
        ("ev_blur_begin", C.c_void_p),
        ("ev_blur_end", C.c_void_p),
        ("mm_slots", C.c_void_p),
        ("mm_nslots", C.c_int32),
        ("seg_in_u8", C.c_void_p),
        ("arena_host", C.c_void_p),
        ("arena_dev", C.c_void_p),
        ("arena_bytes", C.c_uint64),
        ("overlap", C.c_int32),
        ("ws_seq", C.c_uint64),
        ("seg_out_u8", C.c_void_p),
        ("trace_events", C.c_void_p),
        ("trace_ids", C.c_void_p),
        ("trace_cap", C.c_int32),
        ("trace_start", C.c_int32),
        ("trace_first_id", C.c_int32),
        ("label_codes", C.c_void_p),
        ("code_tuples", C.c_void_p),
        ("code_ntuples", C.c_int32),
        ("code_stride", C.c_int32),
        ("code_sel", C.c_int32 * 4),
        ("ride_draw", C.c_void_p),
        ("ride_draw_blocks", C.c_uint32),
        ("rode", C.c_void_p),
    ]


class KeyedConfig(C.Structure):
    _fields_ = [
        ("shape", C.c_int32 * 3), ("size", C.c_int32 * 3), ("resolution", C.c_double * 3),
        ("min_subclusters", C.c_int32), ("max_subclusters", C.c_int32), ("meta_labels", C.c_int32),
        ("nlabels", C.c_int32), ("n_seed_labels", C.c_int32), ("tie_classes", C.c_int32),
        ("seed_labels", C.c_uint8 * 256), ("generation_classes", C.c_uint8 * 256),
        ("deform_prob", C.c_double), ("flip_prb", C.c_double), ("max_rotation", C.c_double), ("max_shear", C.c_double),
        ("max_scaling", C.c_double), ("nonlinear", C.c_int32),
        ("nonlin_scale_min", C.c_double), ("nonlin_scale_max", C.c_double), ("nonlin_std_max", C.c_double),
        ("gamma_prob", C.c_double), ("gamma_std", C.c_double),
        ("bias_prob", C.c_double), ("bf_scale_min", C.c_double), ("bf_scale_max", C.c_double), ("bf_std_min", C.c_double),
        ("bf_std_max", C.c_double),
        ("resample_prob", C.c_double), ("min_resolution", C.c_double), ("max_resolution", C.c_double),
        ("noise_prob", C.c_double), ("noise_std_min", C.c_double), ("noise_std_max", C.c_double),
    ]


class KeyedDraws(C.Structure):
    _fields_ = [
        ("key", C.c_uint64), ("subclusters", C.c_int32 * 4), ("ntab", C.c_int32),
        ("deform_active", C.c_int32), ("flip", C.c_int32),
        ("rotations", C.c_double * 3), ("shears", C.c_double * 3), ("scalings", C.c_double * 3),
        ("A", C.c_float * 9), ("c2", C.c_double * 3),
        ("nonlinear", C.c_int32), ("nonlin_scale", C.c_double), ("nonlin_std", C.c_double), ("field_dims", C.c_int32 * 3),
        ("gamma_active", C.c_int32), ("gamma", C.c_double),
        ("bias_active", C.c_int32), ("bf_scale", C.c_double), ("bf_std", C.c_double), ("bias_dims", C.c_int32 * 3),
        ("resample_active", C.c_int32), ("spacing", C.c_double), ("u_std", C.c_double), ("stds", C.c_double * 3),
        ("low_shape", C.c_int32 * 3), ("blur_ntaps", C.c_int32 * 3),
        ("noise_active", C.c_int32), ("noise_std", C.c_double), ("noise_std32", C.c_float),
        ("off_mm8", C.c_int32), ("off_slots", C.c_int32), ("off_mus", C.c_int32), ("off_sigmas", C.c_int32),
        ("off_bias", C.c_int32), ("off_field", C.c_int32), ("block_bytes", C.c_int32),
        ("rode", C.c_int32),
    ]


E_NOTABLE = -4
KT_RESAMPLE, KT_BACK, KT_FIELD, KT_BIAS = 0, 1, 2, 3
KEYED_I = dict(KEY=0, OUT=1, SEG_OUT=2, SEG_OUT_U8=3, SEG_IN=4, SEG_IN_U8=5, BLOCK=6, WS0=7, WS1=8, WS_LOW=9, WS_ROWS=10,
               ROW_STRIDE=11, SCALE01=12, TRACE_EVENTS=13, TRACE_IDS=14, TRACE_CAP=15, BANK=16, EV_BLUR_BEGIN=80,
               EV_BLUR_END=81, CODES=82, CODE_TUPLES=83, CODE_NTUPLES=84, CODE_STRIDE=85, FLAGS=86, NEXT_KEY=87, NEXT_BLOCK=88, COUNT=89)

STAGE_NAMES = ("begin", "upload", "draw", "head", "floormin", "warp", "blur_x", "blur_y", "blur_z", "blur_yz", "k7", "k9a", "k9b",
               "gmm", "rows", "pointwise", "blur_rs_x", "blur_rs_yz")  # include/fsg_hip.h: FSG_ST_*


P, I, F, SZ, U64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64

# name -> argtypes; every function returns int except where noted.  Must list EVERY symbol the
# header declares (tests/test_abi.py cross-checks against include/fsg_hip.h).
SIGNATURES = {
    "fsg_abi_version": [],
    "fsg_set_tuning": [I],
    "fsg_warp_set_variant": [I],
    "fsg_randn_f32": [P, SZ, U64, U64, P],
    "fsg_gmm_sample_u8": [P, SZ, P, P, I, P, U64, U64, P, P],
    "fsg_gmm_sample_i64": [P, SZ, P, P, I, P, U64, U64, P, P],
    "fsg_gmm_sample_u8x4": [P, P, P, P, SZ, P, P, I, P, U64, U64, P, P],
    "fsg_gmm_sample_u8x4_mm": [P, P, P, P, SZ, P, P, I, P, U64, U64, P, P, I, I, P],
    "fsg_label_stats_u8": [P, P, SZ, I, P, P, P, P],
    "fsg_zoom3d_f32": [P, I, I, I, I, P, P, P, P, I, I, I, P],
    "fsg_resample_noise_f32": [P, I, I, I, P, P, P, P, I, I, I, I, P, U64, U64, F, P],
    "fsg_zoom3d_minmax_f32": [P, I, I, I, P, P, P, I, I, I, P, P],
    "fsg_zoom3d_normalise_f32": [P, I, I, I, P, P, P, P, I, I, I, P, I, P],
    "fsg_minmax_init": [P, I, I, P],
    "fsg_zoom3d_minmax_sharded_f32": [P, I, I, I, P, P, P, I, I, I, P, I, P],
    "fsg_zoom3d_normalise_sharded_f32": [P, I, I, I, P, P, P, P, I, I, I, P, I, I, P],
    "fsg_deform_rows_f32": [C.POINTER(Deform), C.POINTER(Epilogue), P, I, P],
    "fsg_coords_minmax_f32": [C.POINTER(Deform), P, P],
    "fsg_coords_floormin_f32": [C.POINTER(Deform), P, P],
    "fsg_coords_f32": [C.POINTER(Deform), P, P, P, P, P],
    "fsg_warp_f32": [C.POINTER(Deform), P, P, P, P, P, C.POINTER(Epilogue), P],
    "fsg_warp_f32_u8": [C.POINTER(Deform), P, P, P, P, P, C.POINTER(Epilogue), P],
    "fsg_warp_f32_u8_to_f32": [C.POINTER(Deform), P, P, P, P, P, C.POINTER(Epilogue), P],
    "fsg_interp3d_f32": [P, I, I, I, P, P, P, SZ, I, F, P, P],
    "fsg_gamma_f32": [P, SZ, F, P, P],
    "fsg_bias_mul_f32": [P, I, I, I, P, I, I, I, P, P, P, P, P],
    "fsg_blur_axis_f32": [P, P, I, I, I, I, P, I, P],
    "fsg_blur_axis_taps_host_f32": [P, P, I, I, I, I, C.POINTER(C.c_float), I, P],
    "fsg_blur_yz_taps_host_f32": [P, P, I, I, I, C.POINTER(C.c_float), I, C.POINTER(C.c_float), I, P],
    "fsg_add_noise_f32": [P, SZ, P, U64, U64, F, P, P],
    "fsg_reduce_minmax_f32": [P, SZ, P, P],
    "fsg_scale_f32": [P, SZ, P, I, P, P],
    "fsg_slice_acq_forward_f32": [P, P, P, P, I, I, I, P, P, P, I, I, I, I, I, I, F, I, P],
    "fsg_slice_acq_adjoint_f32": [P, P, I, I, I, P, P, P, P, P, P, I, I, I, I, I, I, F, I, P],
    "fsg_equalize_f32": [P, P, P, F, SZ, P],
    "fsg_slice_acq_set_tuning": [I, I, I],
    "fsg_mog3d_f32": [P, P, I, I, I, I, P, P, P],
    "fsg_perlin_fractal_f32": [P, P, P, P, I, I, I, I, P, P, P],
    "fsg_blend_f32": [P, P, P, SZ, I, P, F, P, I, P, P, F, P, P, P],
    "fsg_slice_noise_f32": [P, SZ, F, F, P, P, U64, U64, P],
    "fsg_slice_void_f32": [P, I, I, P, P, I, P, P, P],
    "fsg_slice_sums_f32": [P, I, SZ, P, P],
    "fsg_nonzero_count_f32": [P, SZ, I, F, P, P],
    "fsg_nonzero_count_u8": [P, SZ, I, F, P, P],
    "fsg_nonzero_select_f32": [P, SZ, I, F, P, P, I, P, P],
    "fsg_nonzero_select_u8": [P, SZ, I, F, P, P, I, P, P],
    "fsg_compact_f32": [P, P, SZ, I, F, P, P, P],
    "fsg_ewise_f32": [P, P, SZ, I, F, P, P],
    "fsg_dist_pass_f32": [P, P, I, I, I, I, I, I, I, P],
    "fsg_boundary_mask_f32": [P, P, P, P, P, I, SZ, P, P, P],
    "fsg_bernoulli_keep_f32": [P, SZ, F, U64, U64, P, P],
    "fsg_scatter_const_f32": [P, SZ, P, I, F, P],
    "fsg_copy_bytes": [P, P, SZ, P],
    "fsg_zoom_set_tuning": [I, I],
    "fsg_sample_head_f32": [P, P, P, P, SZ, P, P, I, P, U64, U64, P, C.POINTER(Deform), C.POINTER(Epilogue), P, I, P, P],
    "fsg_seed_codes_build": [P, I, SZ, I, P, P, I, P, SZ, P, P],
    "fsg_sample_head_codes_f32": [P, P, I, I, C.POINTER(C.c_int32), SZ, P, P, I, U64, U64, P, C.POINTER(Deform), C.POINTER(Epilogue), P, I, P, P],
    "fsg_coords_floormin_rest_f32": [C.POINTER(Deform), P, P],
    "fsg_sample_run": [C.POINTER(SamplePlan), P],
    "fsg_sample_plan_pack": [C.POINTER(SamplePlan), P, I, P, I, P],
    "fsg_sample_pack_run": [P, I, P, I, P, P],
    "fsg_sample_run_batch": [C.POINTER(SamplePlan), I, C.POINTER(C.c_void_p), I],
    "fsg_cast_f32_to_f16": [P, SZ, P, P],
    "fsg_pipeline_teardown": [],
    "fsg_blur_resample_supported": [I, I, I, I, I, I, I, I, I],
    "fsg_blur_resample_x_f32": [P, I, I, I, P, I, C.POINTER(C.c_float), I, P, P],
    "fsg_blur_resample_yz_noise_f32": [P, I, I, I, P, P, I, I, C.POINTER(C.c_float), I, C.POINTER(C.c_float), I, I, P, U64, U64,
                                       F, P, P],
    "fsg_keyed_create": [C.POINTER(KeyedConfig), C.POINTER(C.c_void_p)],
    "fsg_keyed_destroy": [P],
    "fsg_keyed_set_table": [P, I, I, I, P],
    "fsg_keyed_draw": [P, U64, C.POINTER(KeyedDraws)],
    "fsg_keyed_sample_run": [P, P, I, P, P],
    "fsg_keyed_fill_block": [P, C.POINTER(KeyedDraws), P, P],
    "fsg_event_destroy": [P],
    "fsg_event_record": [P, P],
    "fsg_event_elapsed_ms": [P, P, C.POINTER(C.c_float)],
}
SPECIAL_RESTYPE = {"fsg_error_string": (C.c_char_p, [I]), "fsg_key_to_float": (F, [C.c_int32]),
                   "fsg_event_create": (C.c_void_p, []), "fsg_sample_plan_layout": (C.c_int64, [I]),
                   "fsg_keyed_block_bytes": (C.c_int64, [P]), "fsg_seed_codes_work_bytes": (C.c_size_t, [])}

_lib = None


def load():
    """dlopen libfsg_hip.so and declare prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). fetalsyngen_amd has no CPU or PyTorch fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (res, args) in SPECIAL_RESTYPE.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    if lib.fsg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libfsg_hip.so ABI version {lib.fsg_abi_version()}, this binding expects {ABI_VERSION}")
    # the structs cross the boundary by pointer: a mirror that disagrees on their size would have the library read past its end
    if lib.fsg_sample_plan_layout(0) != C.sizeof(SamplePlan):
        raise RuntimeError(f"fsg_sample_plan is {lib.fsg_sample_plan_layout(0)} bytes in libfsg_hip.so, "
                           f"{C.sizeof(SamplePlan)} in the ctypes mirror")
    _lib = lib
    return lib


def check(code: int, where: str):
    if code != 0:
        msg = load().fsg_error_string(code)
        raise FsgError(code, where, msg.decode() if msg else "?")
