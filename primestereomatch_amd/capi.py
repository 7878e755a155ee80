"""ctypes loader for libprimesm_hip.so - the C ABI declared in include/primesm_hip.h.

This is the Python-side "hipUtil": it dlopen()s the HIP library at run time the way the
reference's oclUtil JIT-loads its .cl files (src/oclUtil.cpp:438-496).  There is NO CPU
fallback: if the library is missing or no GPU is present the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprimesm_hip.so")

# enums of include/primesm_hip.h
PSM_F32, PSM_U8 = 0, 1
PSM_IMG_U8, PSM_IMG_F32 = 0, 1
PSM_LEFT, PSM_RIGHT = 0, 1
PSM_STAGE_CVC, PSM_STAGE_CVF, PSM_STAGE_DISPSEL, PSM_STAGE_PP = 0, 1, 2, 3
(PSM_K_PREP, PSM_K_CVC, PSM_K_GUIDE, PSM_K_CVF_A, PSM_K_CVF_B, PSM_K_WTA, PSM_K_MERGE, PSM_K_BOX,
 PSM_K_LRC, PSM_K_CVF_F, PSM_K_FGF, PSM_K_WMF) = range(12)
(PSM_OPT_ASYNC, PSM_OPT_KERNEL_VARIANT, PSM_OPT_PROFILE, PSM_OPT_SEG_ROWS, PSM_OPT_WAVES, PSM_OPT_FLAGS, PSM_OPT_GRAPH,
 PSM_OPT_GATHER_STAGED, PSM_OPT_FRAMES_IN_FLIGHT) = range(9)
# enum psm_flag (PSM_OPT_FLAGS bits)
PSM_FLAG_MATERIALISE_COSTS, PSM_FLAG_FGF_STORE, PSM_FLAG_STORE_FILTERED = 128, 4096, 8192
PSM_FLAG_TWO_PHASE_ON, PSM_FLAG_TWO_PHASE_OFF = 1048576, 2097152
PSM_FLAG_WMF_DATAFLOW, PSM_FLAG_WMF_TWO_SWEEPS, PSM_FLAG_WMF_NO_CACHE = 4194304, 8388608, 16777216
PSM_FLAG_F32_TOL = 33554432
PSM_FLAG_FMA_SOLVE = 67108864

# every symbol include/primesm_hip.h declares: (name, restype, argtypes)
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_pi, _pd = C.POINTER(C.c_int), C.POINTER(C.c_double)
SYMBOLS = [
    ("psm_device_count", _i, []),
    ("psm_create", _i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    ("psm_create_shard", _i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i]),
    ("psm_create_shard_strided", _i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i]),
    ("psm_destroy", None, [_vp]),
    ("psm_last_error", C.c_char_p, [_vp]),
    ("psm_set_option", _i, [_vp, _i, _i]),
    ("psm_set_stream", _i, [_vp, _vp]),
    ("psm_synchronize", _i, [_vp]),
    ("psm_release_scratch", _i, [_vp]),
    ("psm_upload_pair", _i, [_vp, _vp, _vp, _i, _sz, _i]),
    ("psm_upload_pair_async", _i, [_vp, _vp, _vp, _i, _sz, _i]),
    ("psm_cost_construct", _i, [_vp]),
    ("psm_cost_filter", _i, [_vp]),
    ("psm_cost_filter_side", _i, [_vp, _i]),
    ("psm_cost_filter_fgf", _i, [_vp, _i]),
    ("psm_disp_select", _i, [_vp, _vp, _vp, _sz]),
    ("psm_disp_select_partial_side", _i, [_vp, _i, _vp]),
    ("psm_disp_select_partial", _i, [_vp, _vp]),
    ("psm_set_key_buffer", _i, [_vp, _vp]),
    ("psm_partial_keys", _i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    ("psm_disp_merge", _i, [_vp, _vp, _i, _vp, _vp, _sz]),
    ("psm_disp_merge_ctx", _i, [_vp, C.POINTER(_vp), _i, _vp, _vp, _sz]),
    ("psm_compute_batch", _i, [C.POINTER(_vp), _i]),
    ("psm_share_streams", _i, [C.POINTER(_vp), _i]),
    ("psm_download_maps", _i, [_vp, _vp, _vp, _sz]),
    ("psm_download_maps_async", _i, [_vp]),
    ("psm_download_maps_wait", _i, [_vp, _vp, _vp, _sz]),
    ("psm_lr_check", _i, [_vp, _vp, _vp, _sz]),
    ("psm_fill_invalid", _i, [_vp, _vp, _vp, _sz]),
    ("psm_wgt_median", _i, [_vp, _vp, _vp, _sz]),
    ("psm_wgt_median_stats", _i, [_vp, _vp, _vp]),
    ("psm_set_rows", _i, [_vp, _i, _i]),
    ("psm_set_map_buffer", _i, [_vp, _vp, _i]),
    ("psm_gather_rows_ctx", _i, [_vp, _vp, _i, _vp, _vp, _sz]),
    ("psm_gather_staged_legs", _i, [_vp]),
    ("psm_upload_maps", _i, [_vp, _vp, _vp, _vp, _vp, _sz]),
    ("psm_download_volume", _i, [_vp, _i, _i, _i, _vp]),
    ("psm_upload_volume", _i, [_vp, _i, _i, _i, _vp]),
    ("psm_filter_stage_a", _i, [_vp, _i]),
    ("psm_download_ab", _i, [_vp, _i, _i, _vp]),
    ("psm_download_guidance", _i, [_vp, _i, _vp]),
    ("psm_box8_volume", _i, [_vp, _i, _vp]),
    ("psm_stage_time_us", _i, [_vp, _i, _pd]),
    ("psm_kernel_time_ms", _i, [_vp, _i, _pd, _pi]),
    ("psm_reset_kernel_times", _i, [_vp]),
    ("psm_filter_launch_times", _i, [_vp, _pd, _pi, _i, _pi]),
    ("psm_get_info", _i, [_vp, _pi, _pi, _pi, _pi, _pi, _pi, _pi]),
]

_lib = None


class PsmError(RuntimeError):
    """A C-ABI call returned non-zero (the reference's `_cl` methods return 1 on failure)."""


def load(path: str | None = None):
    """dlopen the HIP library and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PRIMESM_HIP_LIB") or LIB_PATH   # same override as the C++ hipUtil
    if not os.path.exists(p):
        raise PsmError(
            f"{p} not found: build it with `make -C primestereomatch_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib


def device_count() -> int:
    return int(load().psm_device_count())


def last_error(handle=None) -> str:
    s = load().psm_last_error(handle)
    return s.decode("utf-8", "replace") if s else ""


def check(rc: int, handle=None, what: str = ""):
    if rc != 0:
        raise PsmError(f"{what or 'psm call'} failed: {last_error(handle)}")
