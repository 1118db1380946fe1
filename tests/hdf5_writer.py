"""Minimal HDF5 writer over libhdf5's C API (ctypes), for the tests of the HDF5 input path.
h5py is not installed for this interpreter; libhdf5 itself is in the image."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_CANDIDATES = [os.environ.get("HNSW_AMD_HDF5_LIB"), "libhdf5.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so"]


def _load():
    for n in _CANDIDATES:
        if not n:
            continue
        try:
            return C.CDLL(n)
        except OSError:
            continue
    return None


def available() -> bool:
    return _load() is not None


def write(path: str, datasets: dict[str, np.ndarray]) -> None:
    L = _load()
    hid = C.c_int64
    L.H5open()
    L.H5Fcreate.restype = hid; L.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]
    L.H5Screate_simple.restype = hid; L.H5Screate_simple.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.H5Dcreate2.restype = hid; L.H5Dcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid, hid]
    L.H5Dwrite.restype = C.c_int; L.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    for fn in (L.H5Dclose, L.H5Sclose, L.H5Fclose):
        fn.argtypes = [hid]
    f = L.H5Fcreate(path.encode(), 2, 0, 0)       # H5F_ACC_TRUNC
    assert f >= 0
    types = {np.dtype("float32"): hid.in_dll(L, "H5T_NATIVE_FLOAT_g").value,
             np.dtype("int32"): hid.in_dll(L, "H5T_NATIVE_INT_g").value,
             np.dtype("float64"): hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value}
    for name, arr in datasets.items():
        a = np.ascontiguousarray(arr)
        dims = (C.c_uint64 * a.ndim)(*a.shape)
        sp = L.H5Screate_simple(a.ndim, dims, None)
        t = types[a.dtype]
        d = L.H5Dcreate2(f, name.encode(), t, sp, 0, 0, 0)
        assert d >= 0
        assert L.H5Dwrite(d, t, 0, 0, 0, a.ctypes.data) >= 0
        L.H5Dclose(d); L.H5Sclose(sp)
    L.H5Fclose(f)
