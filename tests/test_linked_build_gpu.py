"""Device-linked build: search, selection and linking of layer 0 all run on the GPU (hnsw_amd_insert_batch_linked); the
host only inserts into the upper layers.  The graph is neither the reference's nor the host-linked assisted build's
(another summation order between stored rows, sub-batches that do not see each other), so the checks are those of any
HNSW graph — list invariants, the selection rule re-checked on the host for a sample of lists, recall against exact
ground truth next to the host-only concurrent build — plus bit-exact search parity against the oracle ON the graph it
built."""
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


def _check_lists(g, n):
    adj = g.adj0.reshape(n, g.deg0)
    cnt = g.cnt0
    live = np.arange(g.deg0)[None, :] < cnt[:, None]
    assert (adj[live] < n).all() and (adj[~live] == 0xFFFFFFFF).all()
    assert cnt.min() >= 1                                            # every vector is linked
    assert not (adj == np.arange(n, dtype=np.uint32)[:, None]).any()  # no self loops
    srt = np.sort(np.where(live, adj.astype(np.int64), -1 - np.arange(g.deg0)[None, :]), axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                          # no duplicate neighbours


def _check_selection_rule(g, base, d_used, rows, slack=1e-4):
    """Every list that was produced by the selection rule obeys it: taken in ascending order of distance to the owner, no
    entry after the second is closer to an earlier entry than to the owner (up to rounding)."""
    adj = g.adj0.reshape(g.n, g.deg0)
    b = base[:, :d_used].astype(np.float64)
    bad = 0
    for i in rows:
        ids = adj[i, :g.cnt0[i]].astype(np.int64)
        own = ((b[ids] - b[i]) ** 2).sum(1)
        order = np.argsort(own, kind="stable")
        ids, own = ids[order], own[order]
        for t in range(2, len(ids)):
            pair = ((b[ids[:t]] - b[ids[t]]) ** 2).sum(1)
            if (pair < own[t] * (1 - slack)).any():
                bad += 1
    return bad


@pytest.mark.parametrize("kind,n,d,dist", [("sift", 150_000, 128, "sse"), ("deep", 60_000, 96, "sse"), ("glove", 40_000, 100, "sse"),
                                           ("glove", 40_000, 100, "scalar")])
def test_linked_build_makes_a_graph_as_good_as_the_host_build(kind, n, d, dist):
    import torch
    base = datagen.make(kind, n, seed=12)
    q = datagen.make(kind, 2000, seed=12, stream=1)
    d_used = d if dist == "scalar" else d & ~15
    bt = torch.from_numpy(np.ascontiguousarray(base[:, :d_used])).cuda()
    qt = torch.from_numpy(np.ascontiguousarray(q[:, :d_used])).cuda()
    _, gt, _ = harness.exact_topk_mfma(bt, qt, 10)
    gt = gt.cpu().numpy()
    def build(mode):
        h = HNSW(16, 16, 200)
        h.init(d, n)
        h.set_build_distance(dist)
        if kind != "sift":
            h.set_build_float_keys(True)
        t0 = time.time()
        h.insert_batch(base, threads=16, gpu=mode)
        return h, time.time() - t0

    host, t_host = build(False)
    dev, t_dev = build("link")
    print(f"{kind} {n} ({dist}): host build {t_host:.1f} s, device-linked {t_dev:.1f} s")
    from hnsw_amd import capi
    flags = (capi.DIST_SCALAR if dist == "scalar" else capi.DIST_SSE) | capi.KEYS_INT32
    host.flags = flags; dev.flags = flags

    g = dev.export_flat()
    assert g.n == n
    _check_lists(g, n)
    # the vectors of the last sub-batch were selected for and never linked to afterwards: their lists are pure outcomes
    # of the selection rule
    assert g.cnt0[n - 200:].max() <= 16
    assert _check_selection_rule(g, base, d_used, np.arange(n - 200, n)) == 0

    for ef in (20, 50):
        r_host, r_dev = _recall(host, q, gt, ef), _recall(dev, q, gt, ef)
        print(f"  ef={ef}: recall@10 host build {r_host:.4f}, device-linked {r_dev:.4f}")
        assert r_dev >= r_host - 0.01

    # the search on the graph it built is the oracle's, bit for bit
    idx = oracle.OracleIndex(g, base)
    ids, dist_, c = dev.search_batch(q[:300], 40)
    oi, od, oc = idx.search(q[:300], 40, threads=8, dist_mode=(oracle.DIST_SCALAR if dist == "scalar" else oracle.DIST_SSE))
    assert np.array_equal(c, oc)
    assert_same_results(ids, dist_, oi, od, what="search on the device-linked graph")


def test_linked_build_in_several_calls_and_small_batches():
    """The index grows over several calls with searches in between (layer 0 goes to the device and comes back each
    time); a call too small for the concurrent path falls back to plain inserts."""
    import torch
    n, d = 60_000, 128
    base = datagen.make("sift", n, seed=14)
    q = datagen.make("sift", 1000, seed=14, stream=1)
    h = HNSW(16, 16, 100)
    h.init(d, n)
    h.insert_batch(base[:500], threads=16, gpu="link")              # (below 2048 rows: sequential inserts)
    assert h.actual_node_count_ == 499
    h.insert_batch(base[500:30_000], threads=16, gpu="link")
    ids, _, _ = h.search_batch(q, 50, out_k=10)
    assert ids.max() < 30_000
    h.insert_batch(base[30_000:], threads=16, gpu="link")
    assert h.actual_node_count_ == n - 1
    g = h.export_flat()
    _check_lists(g, n)
    _, gt, _ = harness.exact_topk_mfma(torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda(), 10)
    assert _recall(h, q, gt.cpu().numpy(), 50) > 0.97
    idx = oracle.OracleIndex(g, base)
    ids, dist, c = h.search_batch(q[:200], 30)
    oi, od, oc = idx.search(q[:200], 30, threads=8)
    assert np.array_equal(c, oc)
    assert_same_results(ids, dist, oi, od, what="after three device-linked calls")


def test_linked_build_with_long_lists_falls_back():
    """Lists longer than the device linking handles (Mmax0 = 2 * 24 = 48 > 32) are built by the host-linked assisted
    path behind the same entry point."""
    n, d = 20_000, 32
    base = datagen.make("sift", n, seed=15)[:, :d].copy()
    h = HNSW(24, 24, 100)
    h.init(d, n)
    h.insert_batch(base, threads=16, gpu="link")
    g = h.export_flat()
    assert g.deg0 == 48
    _check_lists(g, n)
    ids, dist, _ = h.search_batch(base[:100], 30, out_k=1)
    assert (dist[:, 0] == 0).all()


def test_linked_build_refuses_a_gpu_that_is_not_there():
    from hnsw_amd import capi
    base = datagen.make("sift", 3000, seed=16)
    h = HNSW(16, 16, 100)
    h.init(128, 3000)
    rc = capi.lib().hnsw_amd_insert_batch_linked(h._h, capi.f32(base), 3000, 4, 99)
    assert rc != 0 and b"no such GPU" in capi.lib().hnsw_amd_last_error()
    # the index is untouched and still usable
    h.insert_batch(base, threads=4, gpu="link")
    ids, dist, _ = h.search_batch(base[:50], 30, out_k=1)
    assert (dist[:, 0] == 0).all()
