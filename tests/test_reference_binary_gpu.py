"""GPU vs the UNMODIFIED reference, end to end: this library builds and saves an index in the reference's file format,
the reference binary (oracle/_ref/ref_driver_*, compiled from the reference's own sources by oracle/Makefile; it
travels with the tree) reads it with its own loadGraph and runs its own aproximateKnn, and the complete W_ of every
query (all ef ids and float distances) is compared with what the HIP kernel returns.  Skipped when the binary is not there."""
import os
import subprocess

import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen
from oracle import oracle
from tests.helpers import assert_same_results

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,build,flags,n,efs", [
    ("sift", "sse", capi.DIST_SSE | capi.W_STD_HEAP, 100_000, (10, 64, 150)),
    ("sift", "sse", capi.DIST_SSE, 100_000, (50,)),
    ("sift", "sse", capi.DIST_SSE | capi.ROWS_U8, 100_000, (50,)),
    ("fmnist", "sse", capi.DIST_SSE, 20_000, (64,)),
    ("deep", "sse", capi.DIST_SSE | capi.W_STD_HEAP, 40_000, (20,)),
    ("gist", "sse", capi.DIST_SSE, 15_000, (40, 100)),
    ("glove", "scalar", capi.DIST_SCALAR | capi.W_STD_HEAP, 40_000, (10,)),
])
def test_kernel_matches_the_reference_binary(tmp_path, kind, build, flags, n, efs):
    drv = oracle.ref_driver_path(build)
    if drv is None:
        pytest.skip("oracle/_ref is not built (needs the reference sources)")
    try:
        subprocess.run([drv], capture_output=True, timeout=20)
    except OSError:
        pytest.skip("the reference binary does not run on this host")
    nq = 300
    base = datagen.make(kind, n, seed=21)
    q = datagen.make(kind, nq, seed=21, stream=1)
    h = HNSW(16, 16, 100)
    h.set_build_distance(build)              # (insert() and the search of one reference build use the same distance)
    h.init(base.shape[1], n)
    h.insert_batch(base, threads=16)
    gpath, qpath, rpath = (str(tmp_path / x) for x in ("g.bin", "q.bin", "r.bin"))
    h.saveGraph(gpath)
    q.tofile(qpath)
    # (the reference never returns from a query that visits 16384 nodes, src/hnsw.h:90-107: hence modest ef on unit-norm data and a time limit)
    subprocess.run([drv, "lquery", gpath, "16", "16", "100", qpath, str(nq), rpath] + [str(e) for e in efs], check=True,
                   capture_output=True, timeout=200)
    ref = oracle.parse_ref_results(rpath)
    for ef in efs:
        r_ids, r_d, r_cnt = ref[ef]
        ids, dist, cnt = h.search_batch(q, ef, flags=flags)
        assert np.array_equal(cnt, r_cnt)
        assert_same_results(ids, dist, r_ids, r_d, what=f"{kind} ef={ef} vs the reference binary")


@pytest.mark.parametrize("kind,build,flags,n,efc,efs", [
    ("sift", "sse", capi.DIST_SSE, 200_000, 100, (10, 50, 150)),
    ("glove", "scalar", capi.DIST_SCALAR, 30_000, 60, (10,)),
])
def test_index_built_and_saved_by_the_reference(tmp_path, kind, build, flags, n, efc, efs):
    """The other direction: the reference binary BUILDS the index with its own insert and writes it with its own
    saveGraph; this library reads that file with loadGraph and searches it on the GPU.  Every W_ must equal what the
    reference returns from the index it built, and the adjacency this library's sequential build makes of the same rows
    must be the loaded one, list for list — fifty times the size of the committed fixtures."""
    drv = oracle.ref_driver_path(build)
    if drv is None:
        pytest.skip("oracle/_ref is not built (needs the reference sources)")
    try:
        subprocess.run([drv], capture_output=True, timeout=20)
    except OSError:
        pytest.skip("the reference binary does not run on this host")
    nq = 300
    base = datagen.make(kind, n, seed=33)
    q = datagen.make(kind, nq, seed=33, stream=1)
    d = base.shape[1]
    bpath, dpath, gpath, qpath, rpath = (str(tmp_path / x) for x in ("base.bin", "dump.bin", "g.bin", "q.bin", "r.bin"))
    base.tofile(bpath)
    q.tofile(qpath)
    subprocess.run([drv, "build", bpath, str(n), str(d), "16", "16", str(efc), dpath, gpath], check=True, capture_output=True, timeout=300)
    subprocess.run([drv, "lquery", gpath, "16", "16", str(efc), qpath, str(nq), rpath] + [str(e) for e in efs], check=True,
                   capture_output=True, timeout=200)
    ref = oracle.parse_ref_results(rpath)
    h = HNSW(16, 16, efc)
    h.loadGraph(gpath)
    assert h.vector_size_ == d
    for ef in efs:
        r_ids, r_d, r_cnt = ref[ef]
        ids, dist, cnt = h.search_batch(q, ef, flags=flags)
        assert np.array_equal(cnt, r_cnt)
        assert_same_results(ids, dist, r_ids, r_d, what=f"{kind} ef={ef}, index built by the reference")
    # the sequential build of this library on the same rows, with the distance of that reference build: the same graph
    mine = HNSW(16, 16, efc)
    mine.set_build_distance(build)
    mine.init(d, n)
    mine.insert_batch(base, threads=1)
    a, b = mine.export_flat(), h.export_flat()
    for name in ("adj0", "up_off", "up_order", "up_down", "up_adj_slot", "up_adj_order", "ep_slot", "ep_order"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
