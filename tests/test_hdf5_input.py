"""HDF5 input path (hdfReader::getDimensions / readData of the reference, src/hdfReader.cpp)
through the C ABI, on an ann-benchmarks-shaped file ("train", "test", "neighbors")."""
import ctypes as C

import numpy as np
import pytest

from hnsw_amd import capi
from tests import hdf5_writer

pytestmark = pytest.mark.skipif(not hdf5_writer.available(), reason="libhdf5 not present")


def _dims(path, name):
    d = (C.c_uint64 * 2)()
    capi.check(capi.lib().hnsw_amd_hdf_get_dimensions(path.encode(), name.encode(), d))
    return int(d[0]), int(d[1])


def test_read_train_test_neighbors(tmp_path):
    rng = np.random.default_rng(0)
    train = rng.random((50, 12), dtype=np.float32)
    test = rng.random((7, 12), dtype=np.float32)
    nb = rng.integers(0, 50, size=(7, 100)).astype(np.int32)
    p = str(tmp_path / "a.hdf5")
    hdf5_writer.write(p, {"train": train, "test": test, "neighbors": nb, "distances": rng.random((7, 100))})
    assert _dims(p, "train") == (50, 12) and _dims(p, "test") == (7, 12) and _dims(p, "neighbors") == (7, 100)
    out = np.empty((50, 12), np.float32)
    capi.check(capi.lib().hnsw_amd_hdf_read_float(p.encode(), b"train", capi.f32(out)))
    assert np.array_equal(out, train)
    outn = np.empty((7, 100), np.int32)
    capi.check(capi.lib().hnsw_amd_hdf_read_int(p.encode(), b"neighbors", outn.ctypes.data_as(C.POINTER(C.c_int32))))
    assert np.array_equal(outn, nb)
    # a float64 dataset read as float is converted by the library, as H5Dread(H5T_NATIVE_FLOAT) does upstream
    outd = np.empty((7, 100), np.float32)
    capi.check(capi.lib().hnsw_amd_hdf_read_float(p.encode(), b"distances", capi.f32(outd)))
    assert np.isfinite(outd).all()


def test_missing_file_and_dataset_are_io_errors(tmp_path):
    d = (C.c_uint64 * 2)()
    assert capi.lib().hnsw_amd_hdf_get_dimensions(b"/nonexistent/x.hdf5", b"train", d) == capi.E_IO
    p = str(tmp_path / "b.hdf5")
    hdf5_writer.write(p, {"train": np.zeros((2, 2), np.float32)})
    assert capi.lib().hnsw_amd_hdf_get_dimensions(p.encode(), b"nope", d) == capi.E_IO
    assert b"nope" in capi.lib().hnsw_amd_last_error()
