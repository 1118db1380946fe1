"""The search call in two halves (hnsw_amd_search_batch_device_begin / _finish): several batches in flight on one
index, on different streams, equal what the same batches give as blocking calls — including batches whose queries
outgrow their scratch and are re-run by finish, and batches that take spill tables (each call in flight has its own)."""
import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen

pytestmark = pytest.mark.gpu


def _bufs(nq, k):
    import torch
    return (torch.full((nq, k), -7, dtype=torch.int32, device="cuda"), torch.full((nq, k), -7.0, dtype=torch.float32, device="cuda"),
            torch.full((nq,), -7, dtype=torch.int32, device="cuda"), torch.full((nq, 8), -7, dtype=torch.int32, device="cuda"))


def _host(b):
    return [x.cpu().numpy() for x in b]


@pytest.mark.parametrize("setup", ["overflow", "spill", "plain"])
def test_begin_begin_finish_finish_equals_two_blocking_calls(monkeypatch, setup):
    import torch
    if setup == "overflow":
        monkeypatch.setenv("HNSW_AMD_LDS_BUDGET", "1")          # bare first-tier estimates: queries overflow and are re-run
        kind, n, ef = "glove", 30000, 40
    elif setup == "spill":
        monkeypatch.setenv("HNSW_AMD_VISITEDB", "1")
        monkeypatch.setenv("HNSW_AMD_VIS_RB_MIN", "8")
        monkeypatch.setenv("HNSW_AMD_WANT_BULK", "900")         # LDS tables far too small: spill tables in device memory
        kind, n, ef = "deep", 40000, 30
    else:
        kind, n, ef = "sift", 30000, 50
    base = datagen.make(kind, n, seed=4)
    h = HNSW(16, 16, 100)
    h.init(base.shape[1], n)
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    flags = capi.DIST_SSE | capi.KEYS_INT32
    q = datagen.make(kind, 4000, seed=4, stream=1)
    qa, qb = torch.from_numpy(q[:2500]).cuda(), torch.from_numpy(q[2500:]).cuda()
    k = 10
    # blocking reference
    ra, rb = _bufs(len(qa), k), _bufs(len(qb), k)
    s0 = torch.cuda.current_stream().cuda_stream
    ret_a = h.search_batch_device(qa.data_ptr(), len(qa), ef, k, *[x.data_ptr() for x in ra], s0, flags)
    ret_b = h.search_batch_device(qb.data_ptr(), len(qb), ef, k, *[x.data_ptr() for x in rb], s0, flags)
    torch.cuda.synchronize()
    ra, rb = _host(ra), _host(rb)
    if setup == "overflow":
        # (a fresh index, so that the first tier is the bare estimate again and not what the blocking calls taught it)
        h2 = HNSW(16, 16, 100)
        h2.import_flat(h.export_flat(), base)
        h2.sync_device(0)
        h = h2
    for rounds in range(2):
        pa, pb = _bufs(len(qa), k), _bufs(len(qb), k)
        st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        ca = h.search_batch_device_begin(qa.data_ptr(), len(qa), ef, k, *[x.data_ptr() for x in pa], st_a.cuda_stream, flags)
        cb = h.search_batch_device_begin(qb.data_ptr(), len(qb), ef, k, *[x.data_ptr() for x in pb], st_b.cuda_stream, flags)
        got_b = h.search_batch_device_finish(cb)               # (finished in the other order)
        got_a = h.search_batch_device_finish(ca)
        torch.cuda.synchronize()
        for x, y, what in zip(_host(pa) + _host(pb), ra + rb, ("ids", "distances", "counts", "counters") * 2):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"{setup}: {what} differ in round {rounds}"
        if setup == "overflow" and rounds == 0:
            assert got_a + got_b > 0 and ret_a + ret_b > 0, "the bare first tier was expected to overflow on this data"


def test_at_most_four_calls_in_flight_and_every_begin_needs_its_finish():
    import torch
    base = datagen.make("sift", 5000, seed=3)
    h = HNSW(16, 16, 100)
    h.init(128, len(base))
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    q = torch.from_numpy(datagen.make("sift", 64, seed=3, stream=1)).cuda()
    bufs = [_bufs(64, 10) for _ in range(5)]
    calls = [h.search_batch_device_begin(q.data_ptr(), 64, 20, 10, *[x.data_ptr() for x in bufs[i]], 0) for i in range(4)]
    with pytest.raises(capi.HnswAmdError) as e:
        h.search_batch_device_begin(q.data_ptr(), 64, 20, 10, *[x.data_ptr() for x in bufs[4]], 0)
    assert e.value.code == capi.E_ARG and "in flight" in str(e.value)
    # the index may not change under a call in flight
    with pytest.raises(capi.HnswAmdError) as e:
        h.init(128, 100)
        h.insert_batch(base[:100])
        h.sync_device(0)
    assert "in flight" in str(e.value)
    for c in calls:
        h.search_batch_device_finish(c)
    with pytest.raises(capi.HnswAmdError):                      # a handle is good for one finish
        h.search_batch_device_finish(calls[0])
    torch.cuda.synchronize()
    first = _host(bufs[0])
    for b in bufs[1:4]:
        for x, y in zip(_host(b), first):
            assert np.array_equal(x, y)
