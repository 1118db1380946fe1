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
