"""Shared helpers of the test-suite: golden-fixture loading and the parity rule."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

from hnsw_amd import datagen
from hnsw_amd.flatgraph import FlatGraph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


class Golden:
    """One fixture: graph exported from the reference build + the reference's W_ per (query, ef)."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.graph = FlatGraph.from_npz(z)
        m = self.meta
        self.base = datagen.make(m["kind"], m["n"], seed=m["seed"], stream=0, d=m["d"])
        self.queries = datagen.make(m["kind"], m["nq"], seed=m["seed"], stream=1, d=m["d"])
        assert datagen.checksum(self.base) == m["base_sha"], "synthetic base set does not regenerate bit-identically"
        assert datagen.checksum(self.queries) == m["query_sha"]
        self.results = {ef: (z[f"ids_ef{ef}"], z[f"dist_ef{ef}"], z[f"cnt_ef{ef}"]) for ef in m["efs"]}
        self.dist_mode = 1 if m["build"] == "scalar" else 0


def assert_same_results(ids_a, dist_a, ids_b, dist_b, *, rtol=0.0, what="", last_group_open=True):
    """The parity rule: row by row, distances equal (bit for bit when rtol == 0, else to
    rtol relative), ids equal except for permutations inside a group of equal (rtol:
    near-equal) distances.  Rows are ascending; padding = id 0xFFFFFFFF / +inf.
    last_group_open: the LAST tie group of a full row may hold different members of the same
    distance, so its ids are not compared.  That is the case when the rows were truncated (top-k
    of a longer list) and when W is kept as a sorted list (which of several entries tied with the
    worst kept distance is dropped is then not the reference's choice).  Pass False when complete
    lists of the std-heap mode are compared: every group must then agree."""
    ids_a, ids_b = np.asarray(ids_a), np.asarray(ids_b)
    dist_a, dist_b = np.asarray(dist_a), np.asarray(dist_b)
    assert ids_a.shape == ids_b.shape and dist_a.shape == dist_b.shape, what
    if rtol == 0.0:
        same = (dist_a.view(np.uint32) == dist_b.view(np.uint32))
        bad = np.argwhere(~same)
        assert bad.size == 0, f"{what}: {len(bad)} distances differ, first at {bad[0]}: {dist_a[tuple(bad[0])]!r} vs {dist_b[tuple(bad[0])]!r}"
    else:
        fin = np.isfinite(dist_a) | np.isfinite(dist_b)
        assert np.allclose(dist_a[fin], dist_b[fin], rtol=rtol, atol=0.0), what
        assert np.array_equal(np.isfinite(dist_a), np.isfinite(dist_b)), what
    rows = np.nonzero((ids_a != ids_b).any(axis=1))[0]
    for r in rows:
        da = dist_a[r]
        k = len(da)
        i = 0
        while i < k:
            j = i + 1
            while j < k and (da[j] == da[i] or (rtol > 0 and abs(da[j] - da[i]) <= rtol * abs(da[i]))):
                j += 1
            ga, gb = sorted(ids_a[r, i:j].tolist()), sorted(ids_b[r, i:j].tolist())
            last_group_truncated = last_group_open and (j == k) and np.isfinite(da[k - 1])
            if ga != gb and not last_group_truncated:
                raise AssertionError(f"{what}: row {r} ids differ outside a distance tie at [{i},{j}): {ids_a[r, i:j]} vs {ids_b[r, i:j]}")
            i = j


def recall_at_k(ids, gt, k=10):
    """Standard recall@k: |top-k(ids) ∩ top-k(gt)| / k averaged over queries."""
    hits = 0
    for a, b in zip(np.asarray(ids)[:, :k], np.asarray(gt)[:, :k]):
        hits += len(set(a.tolist()) & set(b.tolist()))
    return hits / (len(ids) * k)
