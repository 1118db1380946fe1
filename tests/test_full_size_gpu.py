"""GPU, at the headline configuration's full size (1M x 128 base, 10 000 queries): properties that
do not need the CPU checker to process the whole batch, plus the checker on a sample."""
import numpy as np
import pytest

from hnsw_amd import HNSW, datagen, harness
from oracle import oracle
from tests.helpers import assert_same_results

pytestmark = pytest.mark.gpu

N, NQ, EF, K = 1_000_000, 10_000, 50, 10


@pytest.fixture(scope="module")
def big():
    base = datagen.make("sift", N, seed=1)
    q = datagen.make("sift", NQ, seed=1, stream=1)
    h = HNSW(16, 16, 200)
    h.init(128, N)
    h.insert_batch(base, threads=16)
    return base, q, h


def test_full_size_properties(big):
    base, q, h = big
    ids, dist, cnt, st = h.search_batch(q, EF, stats=True)
    assert (cnt == EF).all()
    assert (np.diff(dist, axis=1) >= 0).all()                       # ascending
    assert (ids < N).all()
    srt = np.sort(ids, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                        # no id twice in a row
    # integer-valued data: every reported distance is the exact squared L2 of (query, base[id])
    sel = np.arange(0, NQ, 97)
    exact = ((q[sel][:, None, :].astype(np.float64) - base[ids[sel]].astype(np.float64)) ** 2).sum(-1)
    assert np.array_equal(dist[sel].astype(np.float64), exact)
    # idempotent, and independent of how the batch is cut
    ids2, dist2, _ = h.search_batch(q, EF)
    assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)
    a_i, a_d, _ = h.search_batch(q[:3333], EF)
    b_i, b_d, _ = h.search_batch(q[3333:], EF)
    assert np.array_equal(np.vstack([a_i, b_i]), ids) and np.array_equal(np.vstack([a_d, b_d]), dist)
    # top-k is a prefix of the full list
    k_i, k_d, _ = h.search_batch(q, EF, out_k=K)
    assert np.array_equal(k_i, ids[:, :K]) and np.array_equal(k_d, dist[:, :K])
    # the counters describe a plausible search
    assert (st["n_expand"] >= 1).all() and (st["n_dist"] >= st["n_visited0"]).all()


def test_full_size_sample_against_cpu_restatement(big):
    base, q, h = big
    sel = np.arange(0, NQ, 41)[:200]
    ids, dist, cnt, st = h.search_batch(q[sel], EF, stats=True)
    oi, od, oc, ost = oracle.OracleIndex(h.export_flat(), base).search(q[sel], EF, stats=True, threads=8)
    assert np.array_equal(cnt, oc)
    assert_same_results(ids, dist, oi, od, what="1M sample")
    for f in st.dtype.names:
        assert np.array_equal(st[f], ost[f]), f


def test_full_size_recall(big):
    import torch
    base, q, h = big
    _, gt, info = harness.exact_topk_mfma(torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda(), K)
    ids, _, _ = h.search_batch(q, EF, out_k=K)
    assert harness.recall_at_k(ids, gt.cpu().numpy(), K) >= 0.95


def test_full_size_against_the_reference_binary(big, tmp_path):
    """The 1M index saved in the reference's format, read by the UNMODIFIED reference with its own loadGraph: the complete
    W_ (all ef entries) of 1000 queries against the kernel's, and the reference driver's own Precision figure
    (src/main.cpp:236-280, from the reference's W_) against this library's top-11 figure over 2000 queries: within 0.001."""
    import json
    import subprocess
    import torch
    base, q, h = big
    drv = oracle.ref_driver_path("sse")
    if drv is None:
        pytest.skip("oracle/_ref is not built (needs the reference sources)")
    try:
        subprocess.run([drv], capture_output=True, timeout=20)
    except OSError:
        pytest.skip("the reference binary does not run on this host")
    npar, nprec = 1000, 2000
    gpath, qpath, rpath, gtpath = (str(tmp_path / x) for x in ("g.bin", "q.bin", "r.bin", "gt.bin"))
    h.saveGraph(gpath)
    q[:nprec].tofile(qpath)
    _, gt, _ = harness.exact_topk_mfma(torch.from_numpy(base).cuda(), torch.from_numpy(q[:nprec]).cuda(), K)
    gt = gt.cpu().numpy().astype(np.uint32)
    gt.tofile(gtpath)
    subprocess.run([drv, "lquery", gpath, "16", "16", "200", qpath, str(npar), rpath, str(EF)], check=True, capture_output=True, timeout=400)
    r_ids, r_d, r_cnt = oracle.parse_ref_results(rpath)[EF]
    ids, dist, cnt = h.search_batch(q[:npar], EF)
    assert np.array_equal(cnt, r_cnt)
    assert_same_results(ids, dist, r_ids, r_d, what="1M, full W_ vs the reference binary")
    out = subprocess.run([drv, "lprecision", gpath, "16", "16", "200", qpath, str(nprec), gtpath, str(K), str(EF)], check=True,
                         capture_output=True, text=True, timeout=400).stderr
    ref_precision = json.loads(out.strip().splitlines()[-1])["ref_precision"]
    ids11, _, _ = h.search_batch(q[:nprec], EF, out_k=K + 1)
    ours = harness.reference_precision(ids11, gt, K)
    assert abs(ours - ref_precision) <= 0.001, (ours, ref_precision)


@pytest.fixture(scope="module")
def deep2m():
    n = 2_000_000
    base = datagen.make("deep", n, seed=1)
    q = datagen.make("deep", 4000, seed=1, stream=1)
    h = HNSW(16, 16, 200)
    h.init(96, n)
    h.set_build_float_keys(True)
    h.insert_batch(base, threads=16)
    return base, q, h


def test_deep_shaped_two_million_rows(deep2m):
    """A DEEP-96-shaped shard of 2M rows (768 MB of vectors, 256 MB of neighbour lists), searched with the reference's int32 keys at ef=30: properties over the
    whole batch, the CPU restatement with all counters on a sample."""
    base, q, h = deep2m
    ef = 30
    ids, dist, cnt, st = h.search_batch(q, ef, stats=True)
    assert (cnt == ef).all() and (ids < len(base)).all()
    assert (np.diff(dist, axis=1) >= 0).all()
    srt = np.sort(ids, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()
    ids2, dist2, _ = h.search_batch(q, ef)                       # sized from what the first batch needed: same results
    assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)
    a_i, a_d, _ = h.search_batch(q[:1500], ef)
    assert np.array_equal(a_i, ids[:1500]) and np.array_equal(a_d, dist[:1500])
    sel = np.arange(0, len(q), 27)[:150]
    oi, od, oc, ost = oracle.OracleIndex(h.export_flat(), base).search(q[sel], ef, stats=True, threads=8)
    assert np.array_equal(cnt[sel], oc)
    assert_same_results(ids[sel], dist[sel], oi, od, what="deep 2M sample")
    for f in st.dtype.names:
        assert np.array_equal(st[f][sel], ost[f]), f
    assert st["n_visited0"].max() > 3000           # the regime this test is for: thousands of visited nodes at ef=30
