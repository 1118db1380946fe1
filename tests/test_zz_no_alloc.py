"""The search entry point that allocates nothing (hnsw_amd_search_batch_device_noalloc) against the regular
device entry point, through the C ABI: same ids, distances (bit for bit), counts and counters, on two calls in a
row, the second with a smaller batch (the static scratch array keeps the first call's contents); and the property in
its name: the free device memory is the same before and after a call."""
import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen

pytestmark = pytest.mark.gpu


def _run(h, fn, q_t, ef, out_k, flags, bufs=None):
    import torch
    nq = q_t.shape[0]
    if bufs is None:
        bufs = (torch.empty((nq, out_k), dtype=torch.int32, device="cuda"), torch.empty((nq, out_k), dtype=torch.float32, device="cuda"),
                torch.empty((nq,), dtype=torch.int32, device="cuda"), torch.empty((nq, 8), dtype=torch.int32, device="cuda"))
    ids, ds, cnt, st = bufs
    ids.fill_(-7); ds.fill_(-7.0); cnt.fill_(-7); st.fill_(-7)
    retried = fn(q_t.data_ptr(), nq, ef, out_k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), st.data_ptr(),
                 torch.cuda.current_stream().cuda_stream, flags)
    torch.cuda.synchronize()
    return ids.cpu().numpy(), ds.cpu().numpy().view(np.uint32), cnt.cpu().numpy(), st.cpu().numpy(), retried


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_noalloc_entry_point_equals_the_allocating_one(monkeypatch):
    import torch
    # unit-norm data under int32 keys with the tables held to the bare estimate: queries overflow their first
    # scratch tier, so the retry lists (the part that lives in the static array) are really used
    monkeypatch.setenv("HNSW_AMD_LDS_BUDGET", "1")
    base = datagen.make("glove", 30000, seed=4)
    h = HNSW(16, 16, 100)
    h.init(base.shape[1], len(base))
    h.insert_batch(base, threads=16)
    flags = capi.DIST_SSE | capi.KEYS_INT32
    q_big = torch.from_numpy(datagen.make("glove", 600, seed=4, stream=1)).cuda()
    q_small = q_big[100:170].contiguous()

    # before the index is on the device the entry point must refuse (it may not upload: that allocates)
    with pytest.raises(capi.HnswAmdError) as e:
        _run(h, h.search_batch_device_noalloc, q_small, 40, 10, flags)
    assert e.value.code == capi.E_ARG and "sync_device" in str(e.value)
    h.sync_device(0)

    # another index over the same graph goes first, so that whatever the runtime sets up lazily on a kernel's first
    # launch (this configuration overflows: the retry launches included) exists before free memory is compared
    warm = HNSW(16, 16, 100)
    warm.import_flat(h.export_flat(), base)
    warm.sync_device(0)
    assert _run(warm, warm.search_batch_device_noalloc, q_big, 40, 40, flags)[4] > 0
    _run(warm, warm.search_batch_device_noalloc, q_small, 24, 10, flags)

    seen_retry = 0
    for q_t, ef, out_k in ((q_big, 40, 40), (q_small, 24, 10)):        # second call: smaller batch, other ef
        nq = q_t.shape[0]
        bufs = (torch.empty((nq, out_k), dtype=torch.int32, device="cuda"), torch.empty((nq, out_k), dtype=torch.float32, device="cuda"),
                torch.empty((nq,), dtype=torch.int32, device="cuda"), torch.empty((nq, 8), dtype=torch.int32, device="cuda"))
        # nothing allocated, nothing freed: the device's free memory (the driver's figure, whoever allocates) stays put
        # across calls into preallocated buffers — the first of them overflows, so the retry launches are covered
        free0 = _free_bytes()
        a = _run(h, h.search_batch_device_noalloc, q_t, ef, out_k, flags, bufs)
        assert _free_bytes() == free0, "the no-allocation entry point changed the device's free memory"
        a2 = _run(h, h.search_batch_device_noalloc, q_t, ef, out_k, flags, bufs)
        assert _free_bytes() == free0
        b = _run(h, h.search_batch_device, q_t, ef, out_k, flags)
        for x, y, z, what in zip(a[:4], b[:4], a2[:4], ("ids", "distances", "counts", "counters")):
            assert np.array_equal(x, y) and np.array_equal(z, y), f"{what} differ (nq={nq}, ef={ef})"
        assert (a[2] > 0).all()
        seen_retry += a[4]
    assert seen_retry > 0, "the bare first tier was expected to overflow on this data"

    # beyond the static array's capacity: a clear error, nothing launched
    with pytest.raises(capi.HnswAmdError) as e:
        h.search_batch_device_noalloc(q_big.data_ptr(), 65536, 40, 10, 1, 1, 0, 0, 0, flags)
    assert e.value.code == capi.E_ARG and "65535" in str(e.value)


def test_noalloc_without_a_spill_pool_equals_a_spilling_run(monkeypatch):
    """Tables in LDS cut far below what the queries visit: the allocating entry point continues those queries in spill
    tables in device memory (a pool it allocates on first need); on a fresh index the entry point that allocates nothing
    has no pool and must get the same results through re-runs on larger tables."""
    import torch
    monkeypatch.setenv("HNSW_AMD_VISITEDB", "1")
    monkeypatch.setenv("HNSW_AMD_VIS_RB_MIN", "8")
    monkeypatch.setenv("HNSW_AMD_WANT_BULK", "900")
    base = datagen.make("deep", 40000, seed=6)
    q_t = torch.from_numpy(datagen.make("deep", 400, seed=6, stream=1)).cuda()
    flags = capi.DIST_SSE | capi.KEYS_INT32
    res = {}
    built = HNSW(16, 16, 100)
    built.init(96, len(base))
    built.insert_batch(base, threads=16)
    graph = built.export_flat()
    for which in ("noalloc", "alloc"):
        h = HNSW(16, 16, 100)                       # a fresh index each (same graph): no pool left behind by the other entry point
        h.import_flat(graph, base)
        h.sync_device(0)
        fn = h.search_batch_device_noalloc if which == "noalloc" else h.search_batch_device
        res[which] = [_run(h, fn, q_t, ef, 10, flags) for ef in (30, 30)]
        if which == "noalloc":
            free0 = _free_bytes()
            _run(h, fn, q_t, 30, 10, flags, None)
            # (the output buffers of _run are torch's: cached blocks, no driver allocation on the second use of a size)
            assert _free_bytes() == free0
    for a, b in zip(res["noalloc"], res["alloc"]):
        for x, y, what in zip(a[:4], b[:4], ("ids", "distances", "counts", "counters")):
            assert np.array_equal(x, y), what
    assert res["noalloc"][0][3][:, 6].max() > 1500      # n_visited0: far beyond the 900 the tables were sized for
