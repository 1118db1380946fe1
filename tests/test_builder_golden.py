"""Host index construction (HNSW::insert restated in hnsw_amd/csrc/hnsw_host.cpp) against the
graphs the unmodified reference built from the same vectors: identical adjacency, order
included, on every layer."""
import json
import os

import numpy as np
import pytest

from hnsw_amd import HNSW, datagen
from hnsw_amd import capi
from tests.helpers import GOLD, Golden, golden_names

# (the fixture named *_scalar was built by a -DNO_MANUAL_VECTORIZATION reference: insert() then computes distances with
# that build's plain loop, set_build_distance("scalar"))
@pytest.mark.parametrize("name", golden_names())
def test_sequential_build_reproduces_reference_graph(name):
    g = Golden(name)
    M, Mmax, efC = g.meta["ctor"]
    h = HNSW(M, Mmax, efC)
    h.set_build_distance(g.meta["build"])
    h.init(g.meta["d"], g.meta["n"])
    for row in g.base[:7]:          # the one-at-a-time entry point ...
        h.insert(row)
    h.insert_batch(g.base[7:], threads=1)   # ... and the loop form give the same sequence
    fg = h.export_flat()
    assert fg.layer_sizes() == g.graph.layer_sizes()
    assert fg.same_adjacency(g.graph)
    assert h.actual_node_count_ == g.meta["n"] - 1


@pytest.mark.parametrize("name", ["int_d128_n20000", "unit_d96_n10000", "unit_d100_n10000_scalar"])
def test_larger_build_matches_reference_adjacency_hash(name):
    with open(os.path.join(GOLD, "adjacency_hashes.json")) as f:
        rec = json.load(f)[name]
    base = datagen.make(rec["kind"], rec["n"], seed=rec["seed"], stream=0, d=rec["d"])
    assert datagen.checksum(base) == rec["base_sha"]
    h = HNSW(*rec["ctor"])
    h.set_build_distance(rec["build"])
    h.init(rec["d"], rec["n"])
    h.insert_batch(base, threads=1)
    fg = h.export_flat()
    assert fg.layer_sizes() == rec["layer_sizes"]
    assert fg.adjacency_hash() == rec["adjacency_hash"]


def test_insert_beyond_capacity_is_an_error():
    h = HNSW(16, 16, 200)
    h.init(8, 3)
    rows = np.arange(32, dtype=np.float32).reshape(4, 8)
    for r in rows[:3]:
        h.insert(r)
    with pytest.raises(capi.HnswAmdError) as e:      # reference: std::runtime_error, src/hnsw.cpp:176-179
        h.insert(rows[3])
    assert e.value.code == capi.E_FULL


def test_concurrent_build_is_a_valid_graph_of_the_same_shape():
    base = datagen.make("sift", 6000, seed=3, d=32)
    a = HNSW(16, 16, 100); a.init(32, 6000); a.insert_batch(base, threads=1)
    b = HNSW(16, 16, 100); b.init(32, 6000); b.insert_batch(base, threads=4)
    fa, fb = a.export_flat(), b.export_flat()
    # levels come from the same generator in id order: same layers, same slots, same entry points
    assert fa.layer_sizes() == fb.layer_sizes()
    assert np.array_equal(fa.up_order, fb.up_order) and np.array_equal(fa.up_down, fb.up_down)
    assert np.array_equal(fa.ep_order, fb.ep_order)
    fb.validate()
    # comparable connectivity
    assert abs(float(fa.cnt0.mean()) - float(fb.cnt0.mean())) < 2.0
    assert int(fb.cnt0.min()) >= 1


def test_concurrent_build_keeps_every_list_well_formed():
    """Many threads, long lists, few nodes: inserts overlap as much as they can.  Every exported list must be a
    prefix of distinct in-range ids followed by padding (export_flat validates the padding), every node
    reachable from layer 0's lists must exist, and the graph must be searchable (recall by the CPU checker)."""
    import numpy as np
    from hnsw_amd import HNSW, datagen
    from oracle import oracle
    from tests.helpers import recall_at_k
    n = 6000
    base = datagen.make("deep", n, seed=11)
    q = datagen.make("deep", 50, seed=11, stream=1)
    gt = oracle.bruteforce(base, q, 10, threads=4)
    for rep in range(25):
        h = HNSW(32, 32, 60)
        h.init(96, n)
        h.insert_batch(base, threads=16)
        fg = h.export_flat()            # validates ranges and padding
        for adj, cnt in ((fg.adj0, fg.cnt0), (fg.up_adj_order, fg.up_cnt)):
            if adj.size == 0:
                continue
            srt = np.sort(adj.astype(np.int64), axis=1)
            dup = (srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 0xFFFFFFFF)
            assert not dup.any(), f"rep {rep}: a neighbour list holds an id twice"
        if rep % 8 == 0:
            ids, _, _ = oracle.OracleIndex(fg, base).search(q, 64, out_k=10, key_mode=oracle.KEYS_FLOAT, threads=4)
            assert recall_at_k(ids, gt, 10) > 0.9


def test_float_build_keys_change_nothing_on_integer_data_and_work_on_unit_norm_data():
    """The construction option keys the insert-time searches by the float distance.  On integer-valued data the
    int32 truncation is the identity, so the sequential build still reproduces the reference's graph; on unit-norm
    data (truncated keys all 0..3) it gives another, regular HNSW graph - checked by recall."""
    import numpy as np
    from hnsw_amd import HNSW, datagen
    from oracle import oracle
    from tests.helpers import Golden, recall_at_k
    g = Golden("int_d32_n2000")
    h = HNSW(*g.meta["ctor"])
    h.init(g.meta["d"], g.meta["n"])
    h.set_build_float_keys(True)
    for row in g.base:
        h.insert(row)
    assert h.export_flat().adjacency_hash() == g.meta["adjacency_hash"]
    n = 20000
    base = datagen.make("deep", n, seed=13)
    q = datagen.make("deep", 100, seed=13, stream=1)
    gt = oracle.bruteforce(base, q, 10, threads=4)
    h2 = HNSW(16, 16, 100)
    h2.init(96, n)
    h2.set_build_float_keys(True)
    h2.insert_batch(base, threads=8)
    ids, _, _ = oracle.OracleIndex(h2.export_flat(), base).search(q, 64, out_k=10, key_mode=oracle.KEYS_FLOAT, threads=4)
    assert recall_at_k(ids, gt, 10) > 0.95


def test_concurrent_build_leaves_no_more_weakly_linked_nodes_than_the_sequential_one():
    """A concurrent insert can descend through a node another thread has only just added (no neighbours there yet)
    and link itself thinly.  Counted here: nodes of low degree after a threaded build against the sequential
    (reference-exact) build of the same rows."""
    from hnsw_amd import datagen
    base = datagen.make("sift", 20000, seed=3)

    def weak(threads):
        h = HNSW(16, 16, 200)
        h.init(128, len(base))
        h.insert_batch(base, threads=threads)
        g = h.export_flat()
        up = [g.up_cnt[g.up_off[l]:g.up_off[l + 1]] for l in range(1, g.nlayers)]
        return int((g.cnt0 < 8).sum()), int((g.cnt0 < 2).sum()), sum(int((u <= 1).sum()) for u in up if len(u) > 16), float(g.cnt0.mean())

    s8, s2, sup, smean = weak(1)
    c8, c2, cup, cmean = weak(8)
    assert c8 <= 1.25 * s8 + 20, (c8, s8)            # layer-0 nodes with fewer than M/2 neighbours
    assert c2 <= s2 + 3, (c2, s2)                    # nearly isolated layer-0 nodes
    assert cup <= sup + 3, (cup, sup)                # upper-layer nodes with at most one neighbour
    assert abs(cmean - smean) < 0.2


@pytest.mark.parametrize("kind,n", [("deep", 30_000), ("sift", 30_000)])
def test_concurrent_build_keeps_the_list_invariants(kind, n):
    """Whatever the interleaving of the inserting threads: ids in range behind a live prefix, padding after it, nobody
    its own neighbour (an insert that reaches its own vector through links other inserts made first must pass it by),
    no neighbour twice."""
    base = datagen.make(kind, n, seed=17)
    h = HNSW(16, 16, 60)
    h.init(base.shape[1], n)
    h.insert_batch(base, threads=8)
    g = h.export_flat()
    adj = g.adj0.reshape(n, g.deg0)
    live = np.arange(g.deg0)[None, :] < g.cnt0[:, None]
    assert (adj[live] < n).all() and (adj[~live] == 0xFFFFFFFF).all()
    assert not (adj == np.arange(n, dtype=np.uint32)[:, None]).any()
    srt = np.sort(np.where(live, adj.astype(np.int64), -1 - np.arange(g.deg0)[None, :]), axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()
    U = int(g.up_off[g.nlayers])
    if U:
        up = g.up_adj_order.reshape(U, g.degu)
        ulive = np.arange(g.degu)[None, :] < g.up_cnt[:, None]
        assert not (up == g.up_order[:, None])[ulive].any()


def test_rows_generated_in_slices_side_by_side_are_the_same_rows():
    """datagen.make(threads > 1) generates slices of the rows concurrently; every row depends on its own index only."""
    for kind in ("sift", "deep"):
        a = datagen.make(kind, 300_000, seed=3, row0=12_345)
        b = datagen.make(kind, 300_000, seed=3, row0=12_345, threads=4)
        assert np.array_equal(a, b)


def test_concurrent_build_grows_an_index_over_several_calls():
    """Only a small index gets a sequential prefix; later calls go straight to the concurrent path."""
    base = datagen.make("sift", 9000, seed=5)
    h = HNSW(16, 16, 100)
    h.init(128, 9000)
    h.insert_batch(base[:3000], threads=4)
    h.insert_batch(base[3000:6000], threads=4)
    h.insert_batch(base[6000:], threads=4)
    g = h.export_flat()
    assert g.n == 9000 and g.cnt0.min() >= 1
    adj = g.adj0.reshape(9000, g.deg0)
    live = np.arange(g.deg0)[None, :] < g.cnt0[:, None]
    assert (adj[live] < 9000).all() and not (adj == np.arange(9000, dtype=np.uint32)[:, None]).any()
