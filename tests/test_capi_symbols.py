"""The C-ABI library loads and exports every symbol include/hnsw_amd.h declares (no compute)."""
import ctypes
import os
import re

from hnsw_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hnsw_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hnsw_amd_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name


def test_argument_errors_are_status_codes_not_crashes():
    L = capi.lib()
    h = ctypes.c_void_p()
    assert L.hnsw_amd_create(1, 16, 200, ctypes.byref(h)) == capi.E_ARG       # M < 2
    assert b"M" in L.hnsw_amd_last_error()
    assert L.hnsw_amd_create(16, 16, 200, ctypes.byref(h)) == capi.OK
    assert L.hnsw_amd_init(h, 0, 10) == capi.E_ARG
    assert L.hnsw_amd_load_graph(h, b"/nonexistent/graph.bin") == capi.E_IO
    assert L.hnsw_amd_insert(h, None) == capi.E_ARG
    L.hnsw_amd_destroy(h)
