"""ctypes binding of the C ABI in include/hnsw_amd.h (libhnsw_amd.so).

The library is the product; it must exist (built in-tree by hnsw_amd.build).  There is
no fallback: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HNSW_AMD_LIB: load another build of the same library (e.g. the phase-timing diagnostic build)
LIB_PATH = os.environ.get("HNSW_AMD_LIB") or os.path.join(_HERE, "libhnsw_amd.so")

OK, E_ARG, E_FULL, E_IO, E_HIP, E_SCRATCH, E_GRAPH = 0, -1, -2, -3, -4, -5, -6
DIST_SSE, DIST_SCALAR = 0, 1
KEYS_INT32, KEYS_FLOAT = 0, 2
W_STD_HEAP = 4
ROWS_U8 = 8
ROWS_Q8 = 16
METRIC_IP = 32

_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)

STATS_DTYPE = np.dtype([("n_dist", "<u4"), ("n_expand", "<u4"), ("n_sweeps", "<u4"), ("sum_deg0", "<u4"),
                        ("sum_degu", "<u4"), ("max_c", "<u4"), ("n_visited0", "<u4"), ("n_visited_up", "<u4")])


class FlatGraphC(C.Structure):
    _fields_ = [("n", C.c_uint32), ("d", C.c_uint32), ("nlayers", C.c_uint32), ("deg0", C.c_uint32), ("degu", C.c_uint32),
                ("adj0", _u32p), ("cnt0", _u32p), ("up_off", _u32p), ("up_order", _u32p), ("up_down", _u32p),
                ("up_cnt", _u32p), ("up_adj_slot", _u32p), ("up_adj_order", _u32p), ("ep_slot", _u32p), ("ep_order", _u32p)]


# every symbol include/hnsw_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "hnsw_amd_last_error": (C.c_char_p, []),
    "hnsw_amd_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "hnsw_amd_destroy": (None, [C.c_void_p]),
    "hnsw_amd_init": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "hnsw_amd_insert": (C.c_int, [C.c_void_p, _f32p]),
    "hnsw_amd_insert_batch": (C.c_int, [C.c_void_p, _f32p, C.c_uint32, C.c_int]),
    "hnsw_amd_insert_batch_assisted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "hnsw_amd_insert_batch_linked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "hnsw_amd_set_build_float_keys": (C.c_int, [C.c_void_p, C.c_int]),
    "hnsw_amd_set_build_distance": (C.c_int, [C.c_void_p, C.c_uint32]),
    "hnsw_amd_aproximate_knn": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_uint32, _u32p, _f32p, _u32p]),
    "hnsw_amd_search_batch": (C.c_int, [C.c_void_p, _f32p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, _u32p, _f32p,
                                        _u32p, C.c_void_p]),
    "hnsw_amd_search_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u32p]),
    "hnsw_amd_search_batch_device_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u32p]),
    "hnsw_amd_search_batch_device_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hnsw_amd_search_batch_device_finish": (C.c_int, [C.c_void_p, C.c_void_p, _u32p]),
    "hnsw_amd_reload_env": (C.c_int, [C.c_void_p]),
    "hnsw_amd_save_graph": (C.c_int, [C.c_void_p, C.c_char_p]),
    "hnsw_amd_load_graph": (C.c_int, [C.c_void_p, C.c_char_p]),
    "hnsw_amd_import_flat": (C.c_int, [C.c_void_p, C.POINTER(FlatGraphC), _f32p]),
    "hnsw_amd_export_flat": (C.c_int, [C.c_void_p, C.POINTER(FlatGraphC), C.POINTER(_f32p)]),
    "hnsw_amd_sync_device": (C.c_int, [C.c_void_p, C.c_int]),
    "hnsw_amd_hdf_get_dimensions": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64)]),
    "hnsw_amd_hdf_read_float": (C.c_int, [C.c_char_p, C.c_char_p, _f32p]),
    "hnsw_amd_hdf_read_int": (C.c_int, [C.c_char_p, C.c_char_p, _i32p]),
    "hnsw_amd_bruteforce_knn_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, _f32p]),
    "hnsw_amd_merge_topk_device": (C.c_int, [C.c_void_p, C.c_void_p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hnsw_amd_gather_ceiling_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, _f32p, C.POINTER(C.c_uint64)]),
    "hnsw_amd_last_launch_info": (C.c_int, [C.c_void_p, _u32p, _u32p, _u32p]),
    "hnsw_amd_vector_data": (C.c_int, [C.c_void_p, C.POINTER(_f32p)]),
    "hnsw_amd_get_info": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p, _u32p, _i32p, _i32p, _u32p]),
}

_LIB = None


class HnswAmdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hnsw_amd error {code}: {msg}")
        self.code = code


def _share_hip_runtime_with_torch() -> None:
    """A PyTorch-ROCm wheel ships its own libamdhip64.so and loads it by path.  If this
    library were loaded first it would bind the system copy of the same soname, the process
    would then hold two HIP runtimes, and whichever initialises second finds no device.
    Mapping torch's copy first (when a torch install is present; torch itself is not imported)
    makes the dynamic linker resolve this library's libamdhip64 dependency to that one copy,
    whatever the import order.  Without torch the system runtime is used."""
    if "torch" in sys.modules:
        return                          # torch's runtime is already mapped
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):  # the message below is part of the contract: tests look for it
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m hnsw_amd.build` (there is no CPU fallback)")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)       # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc: int) -> None:
    if rc != OK:
        raise HnswAmdError(rc, lib().hnsw_amd_last_error().decode(errors="replace"))


def f32(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def u32(a: np.ndarray):
    return a.ctypes.data_as(_u32p)
