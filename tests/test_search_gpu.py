"""GPU parity: the HIP search kernel, called through the C ABI, on the graphs exported from
the reference build, against (a) the W_ the unmodified reference produced (golden fixtures)
and (b) the CPU restatement (oracle) on the same inputs, counters included.

Bar: every float distance bit for bit; ids identical except for permutations inside groups
of exactly equal distance (the reference's final std::sort leaves their order open)."""
import numpy as np
import pytest

from hnsw_amd import HNSW, capi
from oracle import oracle
from tests.helpers import Golden, assert_same_results, golden_names

pytestmark = pytest.mark.gpu


def _flags(g):
    return (capi.DIST_SCALAR if g.dist_mode == 1 else capi.DIST_SSE) | capi.KEYS_INT32


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("w", ["set", "std_heap"])
def test_kernel_matches_reference_W(name, w):
    g = Golden(name)
    h = HNSW(*g.meta["ctor"])
    h.import_flat(g.graph, g.base)
    h.flags = _flags(g) | (capi.W_STD_HEAP if w == "std_heap" else 0)
    for ef, (ids_ref, dist_ref, cnt_ref) in g.results.items():
        ids, dist, cnt = h.search_batch(g.queries, ef)
        assert np.array_equal(cnt, cnt_ref), f"{name} ef={ef}: |W_| differs"
        # W_ kept as the reference keeps it (std_heap): even the members of the last tie group agree
        assert_same_results(ids, dist, ids_ref, dist_ref, what=f"{name} ef={ef}", last_group_open=(w != "std_heap"))
        if w == "std_heap":
            assert np.array_equal(np.sort(ids, axis=1), np.sort(ids_ref, axis=1))


@pytest.mark.parametrize("name", ["int_d32_n2000", "unit_d100_n2000_sse", "unit_d100_n2000_scalar", "int_d16_n1000_m5"])
@pytest.mark.parametrize("keys", ["int32", "float"])
@pytest.mark.parametrize("w", ["set", "std_heap"])
def test_kernel_matches_oracle_with_counters(name, keys, w):
    g = Golden(name)
    h = HNSW(*g.meta["ctor"])
    h.import_flat(g.graph, g.base)
    kflag, kmode = (capi.KEYS_FLOAT, oracle.KEYS_FLOAT) if keys == "float" else (capi.KEYS_INT32, oracle.KEYS_INT32)
    dflag = (capi.DIST_SCALAR if g.dist_mode == 1 else capi.DIST_SSE) | (capi.W_STD_HEAP if w == "std_heap" else 0)
    idx = oracle.OracleIndex(g.graph, g.base)
    for ef in (1, 7, 33, 64, 65, 128, 129, 300):
        ids, dist, cnt, st = h.search_batch(g.queries, ef, flags=dflag | kflag, stats=True)
        oi, od, oc, ost = idx.search(g.queries, ef, dist_mode=g.dist_mode, key_mode=kmode, stats=True)
        assert np.array_equal(cnt, oc)
        # (W_ kept as the reference keeps it: even the members of the last tie group agree)
        assert_same_results(ids, dist, oi, od, what=f"{name} ef={ef} keys={keys}", last_group_open=(w != "std_heap"))
        for field in st.dtype.names:
            assert np.array_equal(st[field], ost[field]), f"{name} ef={ef}: counter {field} differs"


@pytest.mark.parametrize("name", ["int_d128_n4000", "unit_d100_n2000_sse"])
def test_row_shape_kernels_equal_the_kernel_for_any_row_length(name, monkeypatch):
    """Rows of 128 and 96 used dims (d = 100 under the SSE loop) run kernels built for that shape; the kernel for any
    row length (HNSW_AMD_GENERIC_ROWS) must give the same W_ and the same counters, and both the reference's."""
    g = Golden(name)
    h = HNSW(*g.meta["ctor"])
    h.import_flat(g.graph, g.base)
    for ef in (10, 50, 100, 200):
        monkeypatch.delenv("HNSW_AMD_GENERIC_ROWS", raising=False)
        h.sync_device()
        h.reload_env()           # (the switches are read when the index goes to the GPU, not per call)
        ids, dist, cnt, st = h.search_batch(g.queries, ef, flags=_flags(g), stats=True)
        monkeypatch.setenv("HNSW_AMD_GENERIC_ROWS", "1")
        h.reload_env()
        ids2, dist2, cnt2, st2 = h.search_batch(g.queries, ef, flags=_flags(g), stats=True)
        assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2) and np.array_equal(cnt, cnt2)
        for field in st.dtype.names:
            assert np.array_equal(st[field], st2[field]), f"{name} ef={ef}: counter {field} differs"
        if ef in g.results:
            ids_ref, dist_ref, cnt_ref = g.results[ef]
            assert np.array_equal(cnt, cnt_ref)
            assert_same_results(ids, dist, ids_ref, dist_ref, what=f"{name} ef={ef}")


def test_top_k_is_a_prefix_of_W():
    g = Golden("int_d128_n4000")
    h = HNSW(*g.meta["ctor"])
    h.import_flat(g.graph, g.base)
    ids_full, dist_full, _ = h.search_batch(g.queries, 64)
    ids, dist, cnt = h.search_batch(g.queries, 64, out_k=10)
    assert np.array_equal(dist, dist_full[:, :10]) and np.array_equal(ids, ids_full[:, :10])
    assert (cnt == 64).all()


def test_aproximateKnn_fills_W_like_the_reference():
    g = Golden("int_d32_n2000")
    h = HNSW(*g.meta["ctor"])
    h.import_flat(g.graph, g.base)
    ids_ref, dist_ref, cnt_ref = g.results[10]
    h.flags |= capi.W_STD_HEAP
    for i in range(5):
        h.aproximateKnn(g.queries[i], 10, 10)
        assert len(h.W_) == cnt_ref[i]
        assert [w.distance for w in h.W_] == dist_ref[i].tolist()
        assert sorted(w.node_order for w in h.W_) == sorted(ids_ref[i].tolist())


def test_single_layer_and_tiny_graphs():
    # fewer nodes than ef, one layer only: W_ holds every reachable node
    rng = np.random.default_rng(5)
    base = rng.integers(0, 50, size=(9, 16)).astype(np.float32)
    h = HNSW(16, 16, 200)
    h.init(16, 9)
    h.insert_batch(base)
    fg = h.export_flat()
    q = rng.integers(0, 50, size=(4, 16)).astype(np.float32)
    ids, dist, cnt = h.search_batch(q, 32)
    oi, od, oc = oracle.OracleIndex(fg, base).search(q, 32)
    assert np.array_equal(cnt, oc) and (cnt <= 9).all()
    assert_same_results(ids, dist, oi, od, what="tiny")
    assert (ids[:, 9:] == 0xFFFFFFFF).all() and np.isinf(dist[:, 9:]).all()


def test_scratch_overflow_is_retried_transparently(monkeypatch):
    # Unit-norm vectors under the reference's int32 keys visit several times the nodes the first-tier estimate
    # allows for (all keys are 0..3, the stop test never fires).  With the tables held to the bare estimate
    # (HNSW_AMD_LDS_BUDGET: no spare LDS handed to them) queries overflow, are re-run on larger tiers, and the
    # results are the oracle's all the same.
    import torch
    from hnsw_amd import datagen
    monkeypatch.setenv("HNSW_AMD_LDS_BUDGET", "1")
    base = datagen.make("glove", 30000, seed=4)
    q = datagen.make("glove", 200, seed=4, stream=1)
    h = HNSW(16, 16, 100)
    h.init(base.shape[1], len(base))
    h.insert_batch(base, threads=16)
    h.sync_device(0)
    ef, nq = 40, len(q)
    q_t = torch.from_numpy(q).cuda()
    ids_t = torch.empty((nq, ef), dtype=torch.int32, device="cuda")
    dist_t = torch.empty((nq, ef), dtype=torch.float32, device="cuda")
    cnt_t = torch.empty((nq,), dtype=torch.int32, device="cuda")
    retried = h.search_batch_device(q_t.data_ptr(), nq, ef, ef, ids_t.data_ptr(), dist_t.data_ptr(), cnt_t.data_ptr(), 0,
                                    torch.cuda.current_stream().cuda_stream, capi.DIST_SSE | capi.KEYS_INT32)
    torch.cuda.synchronize()
    assert retried > 0, "the bare first tier was expected to overflow on this data"
    oi, od, oc = oracle.OracleIndex(h.export_flat(), base).search(q, ef)
    assert np.array_equal(cnt_t.cpu().numpy().view(np.uint32), oc)
    assert_same_results(ids_t.cpu().numpy().view(np.uint32), dist_t.cpu().numpy(), oi, od, what="overflow + re-run")


def test_two_wavefront_form_with_overflow_and_rerun(monkeypatch):
    """Wide rows in a small batch run two wavefronts per query (the second computes half of every distance batch).
    With the tables held to the bare estimate some queries outgrow them: the first wavefront must let the second one
    go before it reports the overflow, and the re-run (through the list of overflowed queries) gives the oracle's
    results and counters."""
    from hnsw_amd import datagen
    monkeypatch.setenv("HNSW_AMD_LDS_BUDGET", "1")
    monkeypatch.setenv("HNSW_AMD_WANT_VISIT", "96")
    base = datagen.make("gist", 6000, seed=8)
    q = datagen.make("gist", 150, seed=8, stream=1)
    h = HNSW(16, 16, 60)
    h.init(base.shape[1], len(base))
    h.insert_batch(base, threads=16)
    idx = oracle.OracleIndex(h.export_flat(), base)
    for ef in (30, 30, 80):
        ids, dist, cnt, st = h.search_batch(q, ef, stats=True)
        oi, od, oc, ost = idx.search(q, ef, stats=True)
        assert ost["n_visited0"].max() > 200          # the premise: beyond what the first tier was sized for
        assert np.array_equal(cnt, oc)
        assert_same_results(ids, dist, oi, od, what=f"two wavefronts, ef={ef}")
        for f in st.dtype.names:
            assert np.array_equal(st[f], ost[f]), (ef, f)


def test_search_argument_errors():
    h = HNSW(16, 16, 200)
    h.init(8, 10)
    q = np.zeros((2, 8), np.float32)
    with pytest.raises(capi.HnswAmdError) as e:          # nothing inserted yet
        h.search_batch(q, 10)
    assert e.value.code == capi.E_ARG
    h.insert_batch(np.arange(80, dtype=np.float32).reshape(10, 8))
    with pytest.raises(capi.HnswAmdError) as e:          # out_k > ef
        h.search_batch(q, 4, out_k=8)
    assert e.value.code == capi.E_ARG
    for ef in (100000, 2_000_000_000):                   # an ef no on-chip scratch tier can hold
        with pytest.raises(capi.HnswAmdError) as e:
            h.search_batch(q, ef, out_k=1)
        assert e.value.code == capi.E_SCRATCH
    ids, dist, cnt = h.search_batch(q, 10)               # and the index still works
    assert (cnt == 10).all() and sorted(ids[0].tolist()) == list(range(10))


def test_index_can_grow_between_searches():
    base = np.random.default_rng(1).integers(0, 100, size=(400, 16)).astype(np.float32)
    q = base[:5] + 0.0
    h = HNSW(16, 16, 200)
    h.init(16, 400)
    h.insert_batch(base[:200])
    ids1, _, _ = h.search_batch(q, 20, out_k=1)
    assert (ids1[:, 0] == np.arange(5)).all()
    h.insert_batch(base[200:])
    idx = oracle.OracleIndex(h.export_flat(), base)
    ids, dist, cnt = h.search_batch(base[300:310], 20)
    oi, od, oc = idx.search(base[300:310], 20)
    assert_same_results(ids, dist, oi, od, what="after growth")
    assert (ids[:, 0] == np.arange(300, 310)).all() or (dist[:, 0] == 0).all()


def test_all_forms_of_the_visited_set_give_the_same_search(monkeypatch):
    """For larger ef and full batches the launcher switches the visited set to 16-bit slots in id-selected
    regions (half the LDS), either probed linearly inside large regions or kept in small regions that are read
    whole, with an overflow table; HNSW_AMD_VISITED32 / HNSW_AMD_VISITED16 / HNSW_AMD_VISITEDB force one form,
    HNSW_AMD_VIS_RB_MIN the region count a larger index would have.  All are exact sets, so everything must be
    equal, counters included."""
    from hnsw_amd import datagen
    n = 70000                                   # more than 65535 ids: two regions
    base = datagen.make("sift", n, seed=9)
    q = datagen.make("sift", 300, seed=9, stream=1)
    h = HNSW(16, 16, 100)
    h.init(128, n)
    h.insert_batch(base, threads=16)
    envs = ("HNSW_AMD_VISITED32", "HNSW_AMD_VISITED16", "HNSW_AMD_VISITEDB")

    def run(form, ef, rb_min=None):
        for e in envs:
            monkeypatch.delenv(e, raising=False)
        monkeypatch.setenv(form, "1")
        if rb_min is None:
            monkeypatch.delenv("HNSW_AMD_VIS_RB_MIN", raising=False)
        else:
            monkeypatch.setenv("HNSW_AMD_VIS_RB_MIN", str(rb_min))
        h.sync_device()
        h.reload_env()           # (the switches are read when the index goes to the GPU, not per call)
        return h.search_batch(q, ef, stats=True)

    for ef in (8, 60, 100, 200, 256, 300):
        b = run("HNSW_AMD_VISITED32", ef)
        for form, rb_min in (("HNSW_AMD_VISITED16", None), ("HNSW_AMD_VISITEDB", None), ("HNSW_AMD_VISITEDB", 8), ("HNSW_AMD_VISITEDB", 11)):
            a = run(form, ef, rb_min)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (form, rb_min, ef)
            assert np.array_equal(a[2], b[2])
            for f in a[3].dtype.names:
                assert np.array_equal(a[3][f], b[3][f]), (form, rb_min, ef, f)


def test_bucket_form_on_unit_norm_data_against_the_oracle(monkeypatch):
    """The data the bucket form is for: unit-norm rows under int32 keys visit thousands of nodes at a small ef.
    256 regions as for an index of 12.5 M rows; sizes learnt from the first batch are used by the second."""
    from hnsw_amd import datagen
    monkeypatch.setenv("HNSW_AMD_VISITEDB", "1")
    monkeypatch.setenv("HNSW_AMD_VIS_RB_MIN", "8")
    base = datagen.make("deep", 40000, seed=6)
    q = datagen.make("deep", 400, seed=6, stream=1)
    h = HNSW(16, 16, 100)
    h.init(96, len(base))
    h.insert_batch(base, threads=16)
    idx = oracle.OracleIndex(h.export_flat(), base)
    for ef in (30, 30, 10, 80):
        ids, dist, cnt, st = h.search_batch(q, ef, stats=True)
        oi, od, oc, ost = idx.search(q, ef, stats=True)
        assert np.array_equal(cnt, oc)
        assert_same_results(ids, dist, oi, od, what=f"deep ef={ef}")
        for f in st.dtype.names:
            assert np.array_equal(st[f], ost[f]), (ef, f)


def test_learnt_sizes_are_kept_per_ef():
    """The launcher sizes a query's scratch from what earlier batches of the same kind (flags, ef) needed.  A caller
    that alternates between ef values gets each value's learnt configuration back, not a fresh estimate, and the
    results stay the oracle's throughout."""
    from hnsw_amd import datagen
    base = datagen.make("deep", 40000, seed=6)
    q = datagen.make("deep", 4500, seed=6, stream=1)       # (more than 16 per CU: scratch is sized closely)
    h = HNSW(16, 16, 100)
    h.init(96, len(base))
    h.insert_batch(base, threads=16)
    idx = oracle.OracleIndex(h.export_flat(), base)
    settled = {}
    for ef in (30, 80):
        for _ in range(3):
            h.search_batch(q, ef)
        settled[ef] = h.last_launch_info()
    assert settled[30]["lds_bytes_per_query"] != settled[80]["lds_bytes_per_query"]
    for ef in (30, 80, 30, 80):
        ids, dist, cnt = h.search_batch(q, ef)
        assert h.last_launch_info() == settled[ef], ef
        oi, od, oc = idx.search(q, ef, threads=8)
        assert np.array_equal(cnt, oc)
        assert_same_results(ids, dist, oi, od, what=f"alternating ef={ef}")


@pytest.mark.parametrize("tables", [None, 5])
def test_bucket_form_spills_to_device_memory(monkeypatch, tables):
    """Tables in LDS sized for far fewer nodes than the queries visit: the overflow table closes and the search
    continues in a spill table in device memory; with a pool of only five tables most queries find it empty and are
    re-run on larger tables instead.  Results and counters are the oracle's either way."""
    from hnsw_amd import datagen
    monkeypatch.setenv("HNSW_AMD_VISITEDB", "1")
    monkeypatch.setenv("HNSW_AMD_VIS_RB_MIN", "8")
    monkeypatch.setenv("HNSW_AMD_WANT_BULK", "900")
    if tables is not None:
        monkeypatch.setenv("HNSW_AMD_SPILL_TABLES", str(tables))
    base = datagen.make("deep", 40000, seed=6)
    q = datagen.make("deep", 400, seed=6, stream=1)
    h = HNSW(16, 16, 100)
    h.init(96, len(base))
    h.insert_batch(base, threads=16)
    idx = oracle.OracleIndex(h.export_flat(), base)
    for ef in (30, 30, 80):
        ids, dist, cnt, st = h.search_batch(q, ef, stats=True)
        oi, od, oc, ost = idx.search(q, ef, stats=True)
        assert ost["n_visited0"].max() > 1500          # the premise: far beyond the 900 the tables were sized for
        assert np.array_equal(cnt, oc)
        assert_same_results(ids, dist, oi, od, what=f"deep ef={ef} spill")
        for f in st.dtype.names:
            assert np.array_equal(st[f], ost[f]), (ef, f)


def test_visited_set_invariants_under_the_checking_build():
    """The 16-bit forms of the visited set rely on lock-step execution instead of atomics.  A checking build of the
    library (make -C hnsw_amd/csrc checks; never loaded by the package itself) re-reads, after every insertion, that a
    tag counted as the lane's own sits where the lane wrote it, that tags and region counts stay in range and that a
    found tag is in its region, and turns any violation into an error.  The form-equality and spill tests run against it
    in a child process (the library is chosen when the binding is first loaded)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "hnsw_amd", "libhnsw_amd_checks.so")
    if not os.path.exists(lib):
        pytest.fail("hnsw_amd/libhnsw_amd_checks.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = dict(os.environ, HNSW_AMD_LIB=lib)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "--timeout", "300", "--timeout-method", "thread", __file__, "-k",
                          "all_forms_of_the_visited_set or spills_to_device_memory or bucket_form_on_unit_norm"],
                         capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "passed" in out.stdout
