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
def test_quantised_rows_search_then_exact_rescoring(kind, n, ef, monkeypatch):
    """HNSW_AMD_ROWS_Q8 is an approximation of the reference's search (never used for parity): what is
    checked is that the reported distances are the exact float distances of the reported ids, in order,
    and that recall against exact brute force stays close to the float-row search's — through the integer dot
    products (rows of up to 271 dims: the query rounded to the table's grid) and through the float arithmetic
    (HNSW_AMD_NO_DOT4, and the 960-dim set whatever the switch says), under both key modes, with W in registers
    (one, two, four per lane) and as the heap in LDS."""
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
    gt = oracle.bruteforce(base[:, :used], q[:, :used], k, threads=8)
    for keys in (capi.KEYS_FLOAT, capi.KEYS_INT32):
        for e, extra in ((ef, 0), (150, 0), (40, capi.W_STD_HEAP), (300, 0)):
            ids_f, dist_f, _ = h.search_batch(q, e, flags=capi.DIST_SSE | capi.KEYS_FLOAT | extra)
            r_f = recall_at_k(ids_f[:, :k], gt, k)
            for no_dot4 in (False, True):
                if no_dot4:
                    monkeypatch.setenv("HNSW_AMD_NO_DOT4", "1")
                else:
                    monkeypatch.delenv("HNSW_AMD_NO_DOT4", raising=False)
                h.reload_env()
                ids_q, dist_q, cnt_q = h.search_batch(q, e, flags=capi.DIST_SSE | keys | capi.ROWS_Q8 | extra)
                assert (cnt_q == e).all() and (ids_q < n).all()
                assert (np.diff(dist_q, axis=1) >= 0).all()
                exact = ((q[:, None, :used].astype(np.float64) - base[ids_q][:, :, :used].astype(np.float64)) ** 2).sum(-1)
                assert np.allclose(dist_q, exact, rtol=1e-5, atol=1e-6)
                r_q = recall_at_k(ids_q[:, :k], gt, k)
                assert r_q >= r_f - 0.03, (kind, keys, e, extra, no_dot4, r_f, r_q)
    monkeypatch.delenv("HNSW_AMD_NO_DOT4", raising=False)
    h.reload_env()


@pytest.mark.parametrize("d,extra", [(128, 0), (128, capi.W_STD_HEAP), (200, 0), (48, capi.KEYS_FLOAT), (272, 0)])
def test_integer_dot_products_equal_the_float_arithmetic(d, extra, monkeypatch):
    """Integer-valued rows and queries with 255^2 * d below 2^24: no step of the reference's float chains rounds, so the
    byte rows may be read through packed integer dot products; HNSW_AMD_NO_DOT4 keeps the float arithmetic.  Same ids,
    same distance bits, same counters, and nothing is handed back for a re-run.  d = 272 is past the bound
    (272 * 255^2 > 2^24): the float arithmetic is used whatever the switch says."""
    import torch
    n, nq = 30000, 700
    base = datagen.integer_vectors(n, d, seed=31)
    q = datagen.integer_vectors(nq, d, seed=31, stream=1)
    h = HNSW(16, 16, 100)
    h.init(d, n)
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    q_t = torch.from_numpy(q).cuda()
    flags = capi.DIST_SSE | capi.ROWS_U8 | extra
    for ef in (10, 50, 64, 100, 200):
        res = {}
        for mode in ("dot4", "float"):
            if mode == "float":
                monkeypatch.setenv("HNSW_AMD_NO_DOT4", "1")
            else:
                monkeypatch.delenv("HNSW_AMD_NO_DOT4", raising=False)
            h.reload_env()
            ids = torch.empty((nq, ef), dtype=torch.int32, device="cuda")
            ds = torch.empty((nq, ef), dtype=torch.float32, device="cuda")
            cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
            st = torch.empty((nq, 8), dtype=torch.int32, device="cuda")
            retried = h.search_batch_device(q_t.data_ptr(), nq, ef, ef, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0, flags)
            torch.cuda.synchronize()
            res[mode] = (ids.cpu().numpy(), ds.cpu().numpy().view(np.uint32), cnt.cpu().numpy(), st.cpu().numpy(), retried)
        for x, y, what in zip(res["dot4"][:4], res["float"][:4], ("ids", "distances", "counts", "counters")):
            assert np.array_equal(x, y), f"d={d} ef={ef}: {what} differ"
        assert res["dot4"][4] == 0
    # and against the float rows
    a = h.search_batch(q, 50, flags=capi.DIST_SSE | extra)
    b = h.search_batch(q, 50, flags=flags)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_queries_that_are_not_integer_valued_take_the_float_arithmetic():
    """A query with a fractional (or out-of-range) element cannot use the integer dot products: the kernel hands it back
    and the call re-runs it with the float arithmetic on the same byte rows.  Results equal the float-row search's;
    exactly the affected queries are re-run."""
    import torch
    n, nq, ef = 30000, 600, 50
    base = datagen.make("sift", n, seed=33)
    q = datagen.make("sift", nq, seed=33, stream=1)
    odd = np.arange(nq) % 7 == 3
    q[odd, 5] += np.float32(0.5)
    q[np.arange(nq) % 50 == 1, 100] = np.float32(300.0)
    odd |= np.arange(nq) % 50 == 1
    h = HNSW(16, 16, 100)
    h.init(128, n)
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    a = h.search_batch(q, ef, flags=capi.DIST_SSE, stats=True)
    q_t = torch.from_numpy(q).cuda()
    ids = torch.empty((nq, ef), dtype=torch.int32, device="cuda")
    ds = torch.empty((nq, ef), dtype=torch.float32, device="cuda")
    cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    for _ in range(2):
        retried = h.search_batch_device(q_t.data_ptr(), nq, ef, ef, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, 0, capi.DIST_SSE | capi.ROWS_U8)
        torch.cuda.synchronize()
        assert retried == int(odd.sum())
        assert np.array_equal(ids.cpu().numpy().view(np.uint32), a[0]) and np.array_equal(ds.cpu().numpy().view(np.uint32), a[1].view(np.uint32))
        assert np.array_equal(cnt.cpu().numpy().view(np.uint32), a[2])
