"""The measurement helpers of hnsw_amd/harness.py on hand-made inputs (no GPU)."""
import numpy as np

from hnsw_amd import capi, harness


def test_reference_ef_sweep_is_the_drivers():
    # src/main.cpp:232-234 of the reference: 20..100 step 10, then 120..200 step 20
    assert harness.REFERENCE_EF_SWEEP == list(range(20, 101, 10)) + list(range(120, 201, 20))


def test_recall_definitions():
    gt = np.array([[0, 1, 2], [10, 11, 12]])
    ids = np.array([[2, 0, 9, 1], [12, 99, 98, 10]])
    # recall@3: the first 3 results against the first 3 true neighbours
    assert harness.recall_at_k(ids, gt, 3) == (2 + 1) / 6
    # the reference's "Precision" looks at one result more (its loop stops after k+1 pushes, src/main.cpp:236-280)
    assert harness.reference_precision(ids, gt, 3) == (3 + 2) / 6


def test_algorithmic_bytes_formula():
    st = np.zeros(2, capi.STATS_DTYPE)
    st["n_dist"] = [100, 7]
    st["sum_deg0"] = [50, 3]
    st["sum_degu"] = [6, 1]
    b = harness.algorithmic_bytes(st, d=100, d_used=96, k=10)
    assert b.tolist() == [100 * 4 * 96 + 4 * 56 + 400 + 80, 7 * 4 * 96 + 4 * 4 + 400 + 80]
    b8 = harness.algorithmic_bytes(st, d=100, d_used=96, k=10, row_elem_bytes=1)
    assert b8.tolist() == [100 * 96 + 4 * 56 + 400 + 80, 7 * 96 + 4 * 4 + 400 + 80]
