"""The search entry point that allocates nothing (hnsw_amd_search_batch_device_noalloc) against the regular
device entry point, through the C ABI: same ids, distances (bit for bit), counts and counters, on two calls in a
row, the second with a smaller batch (the static scratch array keeps the first call's contents)."""
import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen

pytestmark = pytest.mark.gpu


def _run(h, fn, q_t, ef, out_k, flags):
    import torch
    nq = q_t.shape[0]
    ids = torch.full((nq, out_k), -7, dtype=torch.int32, device="cuda")
    ds = torch.full((nq, out_k), -7.0, dtype=torch.float32, device="cuda")
    cnt = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
    st = torch.full((nq, 8), -7, dtype=torch.int32, device="cuda")
    retried = fn(q_t.data_ptr(), nq, ef, out_k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), st.data_ptr(),
                 torch.cuda.current_stream().cuda_stream, flags)
    torch.cuda.synchronize()
    return ids.cpu().numpy(), ds.cpu().numpy().view(np.uint32), cnt.cpu().numpy(), st.cpu().numpy(), retried


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

    seen_retry = 0
    for q_t, ef, out_k in ((q_big, 40, 40), (q_small, 24, 10)):        # second call: smaller batch, other ef
        a = _run(h, h.search_batch_device_noalloc, q_t, ef, out_k, flags)
        b = _run(h, h.search_batch_device, q_t, ef, out_k, flags)
        for x, y, what in zip(a[:4], b[:4], ("ids", "distances", "counts", "counters")):
            assert np.array_equal(x, y), f"{what} differ (nq={q_t.shape[0]}, ef={ef})"
        assert (a[2] > 0).all()
        seen_retry += a[4]
    assert seen_retry > 0, "the bare first tier was expected to overflow on this data"

    # beyond the static array's capacity: a clear error, nothing launched
    with pytest.raises(capi.HnswAmdError) as e:
        h.search_batch_device_noalloc(q_big.data_ptr(), 65536, 40, 10, 1, 1, 0, 0, 0, flags)
    assert e.value.code == capi.E_ARG and "65535" in str(e.value)
