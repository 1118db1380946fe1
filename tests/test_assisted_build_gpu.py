"""GPU-assisted build: the construction searches of layer-0-only vectors run as batched GPU searches, the host threads
select and link.  The graph is not the reference's (vectors of one sub-batch do not see each other), so the checks
are those of any HNSW graph — list invariants, recall against exact ground truth next to the host-only concurrent
build — plus bit-exact search parity against the oracle ON the graph it built."""
import time

import numpy as np
import pytest

from hnsw_amd import HNSW, datagen, harness
from oracle import oracle
from tests.helpers import assert_same_results

pytestmark = pytest.mark.gpu


def _recall(h, q, gt, ef, k=10):
    ids, _, _ = h.search_batch(q, ef, out_k=k)
    return harness.recall_at_k(ids.astype(np.int64), gt, k)


@pytest.mark.parametrize("kind,n,d", [("sift", 150_000, 128), ("deep", 60_000, 96)])
def test_assisted_build_makes_a_graph_as_good_as_the_host_build(kind, n, d):
    import torch
    base = datagen.make(kind, n, seed=12)
    q = datagen.make(kind, 2000, seed=12, stream=1)
    _, gt, _ = harness.exact_topk_mfma(torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda(), 10)
    gt = gt.cpu().numpy()

    t0 = time.time()
    host = HNSW(16, 16, 200)
    host.init(d, n)
    host.insert_batch(base, threads=16)
    t_host = time.time() - t0
    t0 = time.time()
    gpu = HNSW(16, 16, 200)
    gpu.init(d, n)
    gpu.insert_batch(base, threads=16, gpu=True)
    t_gpu = time.time() - t0
    print(f"{kind} {n}: host build {t_host:.1f} s, GPU-assisted {t_gpu:.1f} s")

    g = gpu.export_flat()
    assert g.n == n
    adj = g.adj0.reshape(n, g.deg0)
    cnt = g.cnt0
    live = np.arange(g.deg0)[None, :] < cnt[:, None]
    assert (adj[live] < n).all() and (adj[~live] == 0xFFFFFFFF).all()
    assert cnt.min() >= 1                                            # every vector is linked
    assert not (adj == np.arange(n, dtype=np.uint32)[:, None]).any()  # no self loops
    srt = np.sort(np.where(live, adj.astype(np.int64), -1 - np.arange(g.deg0)[None, :]), axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                          # no duplicate neighbours

    for ef in (20, 50):
        r_host, r_gpu = _recall(host, q, gt, ef), _recall(gpu, q, gt, ef)
        print(f"  ef={ef}: recall@10 host build {r_host:.4f}, GPU-assisted {r_gpu:.4f}")
        assert r_gpu >= r_host - 0.01

    # the search on the graph it built is the oracle's, bit for bit
    idx = oracle.OracleIndex(g, base)
    ids, dist, c = gpu.search_batch(q[:300], 40)
    oi, od, oc = idx.search(q[:300], 40, threads=8)
    assert np.array_equal(c, oc)
    assert_same_results(ids, dist, oi, od, what="search on the GPU-assisted graph")


def test_assisted_build_in_several_calls_and_small_batches():
    """The index grows over several calls (with searches in between: the device copy is refreshed, not rebuilt from
    nothing); a call too small for the concurrent path falls back to plain inserts."""
    import torch
    n, d = 60_000, 128
    base = datagen.make("sift", n, seed=14)
    q = datagen.make("sift", 1000, seed=14, stream=1)
    h = HNSW(16, 16, 100)
    h.init(d, n)
    h.insert_batch(base[:500], threads=16, gpu=True)              # (below 2048 rows: sequential inserts)
    assert h.actual_node_count_ == 499
    h.insert_batch(base[500:30_000], threads=16, gpu=True)
    ids, _, _ = h.search_batch(q, 50, out_k=10)
    assert ids.max() < 30_000
    h.insert_batch(base[30_000:], threads=16, gpu=True)
    assert h.actual_node_count_ == n - 1
    _, gt, _ = harness.exact_topk_mfma(torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda(), 10)
    assert _recall(h, q, gt.cpu().numpy(), 50) > 0.97
    idx = oracle.OracleIndex(h.export_flat(), base)
    ids, dist, c = h.search_batch(q[:200], 30)
    oi, od, oc = idx.search(q[:200], 30, threads=8)
    assert np.array_equal(c, oc)
    assert_same_results(ids, dist, oi, od, what="after three assisted calls")


def test_assisted_build_after_init_with_other_data_of_the_same_shape():
    """A handle that is init()ed again with the same row count and dimension holds other rows: the device copy of
    the first table must not pass for the second one (few rows, so that no layer-raising vector forces a new upload
    during the second build)."""
    n, d = 4000, 128
    first = datagen.make("sift", n, seed=21)
    second = datagen.make("sift", n, seed=22)
    q = datagen.make("sift", 200, seed=22, stream=1)
    h = HNSW(16, 16, 100)
    h.init(d, n)
    h.insert_batch(first, threads=16, gpu=True)
    h.search_batch(q, 30)
    h.init(d, n)
    h.insert_batch(second, threads=16, gpu=True)
    idx = oracle.OracleIndex(h.export_flat(), second)
    ids, dist, c = h.search_batch(q, 30)
    oi, od, oc = idx.search(q, 30, threads=8)
    assert np.array_equal(c, oc)
    assert_same_results(ids, dist, oi, od, what="second table of the same shape")
    # and the graph was built on the second table's rows: its vectors find themselves
    ids, dist, _ = h.search_batch(second[:100], 30, out_k=1)
    assert (dist[:, 0] == 0).all()
