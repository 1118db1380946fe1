"""Two LDS economies of the search kernel that must not change any result: candidate-heap entries of four bytes (key in
the top byte, node_order below: indexes of fewer than 2^24 short rows under int32 keys, e.g. unit-norm data) and the
tables laid over the query's LDS copy in the kernels that keep the whole query in registers (96 used dims)."""
import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen
from oracle import oracle
from tests.helpers import assert_same_results

pytestmark = pytest.mark.gpu


def _dev_search(h, q_t, ef, flags):
    import torch
    nq = q_t.shape[0]
    ids = torch.empty((nq, ef), dtype=torch.int32, device="cuda")
    ds = torch.empty((nq, ef), dtype=torch.float32, device="cuda")
    cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    st = torch.empty((nq, 8), dtype=torch.int32, device="cuda")
    retried = h.search_batch_device(q_t.data_ptr(), nq, ef, ef, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0, flags)
    torch.cuda.synchronize()
    return ids.cpu().numpy().view(np.uint32), ds.cpu().numpy(), cnt.cpu().numpy().view(np.uint32), st.cpu().numpy(), retried


@pytest.mark.parametrize("kind,dist", [("deep", "sse"), ("glove", "sse"), ("glove", "scalar"), ("gist", "sse")])
def test_packed_entries_change_nothing(kind, dist, monkeypatch):
    """Default (packed where the launcher may) against HNSW_AMD_NO_PACK_C, and both against the oracle, counters included;
    every form of the visited set; ef on both sides of the one- and two-register forms of W."""
    import torch
    n = 6000 if kind == "gist" else 40000
    base = datagen.make(kind, n, seed=41)
    q = datagen.make(kind, 300, seed=41, stream=1)
    h = HNSW(16, 16, 100)
    h.init(base.shape[1], n)
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    idx = oracle.OracleIndex(h.export_flat(), base)
    q_t = torch.from_numpy(q).cuda()
    dmode = 1 if dist == "scalar" else 0
    flags = (capi.DIST_SCALAR if dmode else capi.DIST_SSE) | capi.KEYS_INT32
    for form in (None, "HNSW_AMD_VISITED16", "HNSW_AMD_VISITEDB"):
        for e in ("HNSW_AMD_VISITED16", "HNSW_AMD_VISITEDB"):
            monkeypatch.delenv(e, raising=False)
        if form:
            monkeypatch.setenv(form, "1")
        for ef in (10, 40, 64, 100):
            oi, od, oc, ost = idx.search(q, ef, dist_mode=dmode, stats=True, threads=8)
            res = {}
            for mode in ("packed", "plain"):
                if mode == "plain":
                    monkeypatch.setenv("HNSW_AMD_NO_PACK_C", "1")
                else:
                    monkeypatch.delenv("HNSW_AMD_NO_PACK_C", raising=False)
                h.reload_env()
                res[mode] = _dev_search(h, q_t, ef, flags)
                ids, ds, cnt, st, retried = res[mode]
                assert np.array_equal(cnt, oc)
                assert_same_results(ids, ds, oi, od, what=f"{kind} {dist} ef={ef} {mode} {form}")
                stv = st.view(np.uint32).reshape(len(q), 8).copy().view(capi.STATS_DTYPE).reshape(len(q))
                for f in stv.dtype.names:
                    assert np.array_equal(stv[f], ost[f]), (kind, ef, mode, f)
            assert np.array_equal(res["packed"][0], res["plain"][0])
            # (the packed launch needs less LDS per query)
    monkeypatch.delenv("HNSW_AMD_NO_PACK_C", raising=False)


def test_a_query_too_long_for_small_keys_is_handed_back():
    """Unit-norm rows, but some queries are twenty times as long: their distances pass 255, the packed key would not
    hold them, the kernel hands them back before it starts and the call re-runs them with eight-byte entries."""
    import torch
    n, nq, ef = 30000, 400, 40
    base = datagen.make("deep", n, seed=43)
    q = datagen.make("deep", nq, seed=43, stream=1)
    far = np.arange(nq) % 9 == 2
    q[far] *= np.float32(20.0)
    h = HNSW(16, 16, 100)
    h.init(96, n)
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    idx = oracle.OracleIndex(h.export_flat(), base)
    oi, od, oc = idx.search(q, ef, threads=8)
    assert od[far].min() > 256.0
    for _ in range(2):
        ids, ds, cnt, _, retried = _dev_search(h, torch.from_numpy(q).cuda(), ef, capi.DIST_SSE | capi.KEYS_INT32)
        assert retried >= int(far.sum())
        assert np.array_equal(cnt, oc)
        assert_same_results(ids, ds, oi, od, what="long queries")


def test_long_rows_never_pack():
    """Integer-valued rows (lengths in the thousands): keys are far beyond a byte, the launcher must not try."""
    import torch
    base = datagen.make("sift", 20000, seed=44)
    q = datagen.make("sift", 200, seed=44, stream=1)
    h = HNSW(16, 16, 100)
    h.init(128, len(base))
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    idx = oracle.OracleIndex(h.export_flat(), base)
    oi, od, oc = idx.search(q, 50, threads=8)
    ids, ds, cnt, _, retried = _dev_search(h, torch.from_numpy(q).cuda(), 50, capi.DIST_SSE | capi.KEYS_INT32)
    assert retried == 0
    assert np.array_equal(cnt, oc)
    assert_same_results(ids, ds, oi, od, what="integer rows")
