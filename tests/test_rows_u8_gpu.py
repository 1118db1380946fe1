"""GPU: the byte-row table (HNSW_AMD_ROWS_U8) is a lossless re-encoding for integer-valued data, so a search
that reads it must return exactly what the float-row search returns - and what the unmodified reference
returned for the golden fixtures - down to the per-query counters."""
import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen
from tests.helpers import Golden, assert_same_results

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["int_d32_n2000", "int_d128_n4000", "int_d16_n1000_m5", "int_d32_n1500_m32"])
def test_byte_rows_reproduce_the_reference_W(name):
    g = Golden(name)
    h = HNSW(*g.meta["ctor"])
    h.import_flat(g.graph, g.base)
    for ef in g.meta["efs"]:
        ids, dist, cnt = h.search_batch(g.queries, ef, flags=capi.DIST_SSE | capi.ROWS_U8)
        gi, gd, gc = g.results[ef]
        assert np.array_equal(cnt, gc)
        assert_same_results(ids, dist, gi, gd, what=f"{name} ef={ef} byte rows")


@pytest.mark.parametrize("kind,n,nq,efs,extra", [
    ("sift", 60000, 500, (10, 50, 100, 200), 0),
    ("sift", 20000, 300, (50,), capi.W_STD_HEAP),
    ("fmnist", 20000, 200, (64,), 0),                  # 784 dims: six full 128-dim blocks and a 16-dim tail
    ("sift", 20000, 300, (40,), capi.KEYS_FLOAT),
])
def test_byte_rows_equal_float_rows(kind, n, nq, efs, extra):
    base = datagen.make(kind, n, seed=6)
    q = datagen.make(kind, nq, seed=6, stream=1) + np.float32(0.25)      # queries need not be integers
    h = HNSW(16, 16, 100)
    h.init(base.shape[1], n)
    h.insert_batch(base, threads=16)
    for ef in efs:
        a = h.search_batch(q, ef, flags=capi.DIST_SSE | extra, stats=True)
        b = h.search_batch(q, ef, flags=capi.DIST_SSE | extra | capi.ROWS_U8, stats=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert np.array_equal(a[2], b[2])
        for f in a[3].dtype.names:
            assert np.array_equal(a[3][f], b[3][f]), f


def test_byte_rows_refused_for_other_data():
    base = datagen.make("glove", 3000, seed=6)                           # unit-norm floats
    q = datagen.make("glove", 10, seed=6, stream=1)
    h = HNSW(16, 16, 100)
    h.init(100, 3000)
    h.insert_batch(base, threads=8)
    with pytest.raises(capi.HnswAmdError) as e:
        h.search_batch(q, 20, flags=capi.DIST_SSE | capi.ROWS_U8)
    assert e.value.code == capi.E_ARG and "0..255" in str(e.value)
    ints = datagen.make("sift", 3000, seed=6)
    h2 = HNSW(16, 16, 100)
    h2.init(128, 3000)
    h2.insert_batch(ints, threads=8)
    with pytest.raises(capi.HnswAmdError) as e:
        h2.search_batch(ints[:5], 20, flags=capi.DIST_SCALAR | capi.ROWS_U8)
    assert e.value.code == capi.E_ARG


@pytest.mark.parametrize("kind,n,ef", [("glove", 40000, 60), ("deep", 40000, 60), ("sift", 40000, 50), ("gist", 6000, 64)])
def test_quantised_rows_search_then_exact_rescoring(kind, n, ef):
    """HNSW_AMD_ROWS_Q8 is an approximation of the reference's search (never used for parity): what is
    checked is that the reported distances are the exact float distances of the reported ids, in order,
    and that recall against exact brute force stays close to the float-row search's."""
    from oracle import oracle
    from tests.helpers import recall_at_k
    nq, k = 300, 10
    base = datagen.make(kind, n, seed=8)
    q = datagen.make(kind, nq, seed=8, stream=1)
    d = base.shape[1]
    used = d & ~15
    h = HNSW(16, 16, 100)
    h.init(d, n)
    h.insert_batch(base, threads=16)
    ids_f, dist_f, _ = h.search_batch(q, ef, flags=capi.DIST_SSE | capi.KEYS_FLOAT)
    ids_q, dist_q, cnt_q = h.search_batch(q, ef, flags=capi.DIST_SSE | capi.KEYS_FLOAT | capi.ROWS_Q8)
    assert (cnt_q == ef).all() and (ids_q < n).all()
    assert (np.diff(dist_q, axis=1) >= 0).all()
    exact = ((q[:, None, :used].astype(np.float64) - base[ids_q][:, :, :used].astype(np.float64)) ** 2).sum(-1)
    assert np.allclose(dist_q, exact, rtol=1e-5, atol=1e-6)
    gt = oracle.bruteforce(base[:, :used], q[:, :used], k, threads=8)
    r_f, r_q = recall_at_k(ids_f[:, :k], gt, k), recall_at_k(ids_q[:, :k], gt, k)
    assert r_q >= r_f - 0.03, (r_f, r_q)
