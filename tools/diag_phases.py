#!/usr/bin/env python3
"""Where a query's cycles go (diagnostic build only).

    make -C hnsw_amd/csrc diag
    HNSW_AMD_LIB=hnsw_amd/libhnsw_amd_diag.so python tools/diag_phases.py --n 1000000 --ef 50

The diagnostic library overwrites the per-query counters with s_memtime cycle sums per phase
of the layer-0 loop; the shipped library contains no stamps."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hnsw_amd import HNSW, datagen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ef", type=int, default=50)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--kind", default="sift")
ap.add_argument("--dist", default="sse", choices=["sse", "scalar"])
ap.add_argument("--keys", default="int32", choices=["int32", "float"])
ap.add_argument("--cache", default="", help="npz of the flat graph (as tools/probe.py --cache writes it): load instead of building")
ap.add_argument("--build-keys", default="int32", choices=["int32", "float"])
ap.add_argument("--want-visit", type=int, default=0,
                help="the diagnostic kernel reports no sizes, so the launcher cannot learn them: give the visited count to size for")
a = ap.parse_args()
base = datagen.make(a.kind, a.n, seed=1)
q = datagen.make(a.kind, a.nq, seed=1, stream=1)
h = HNSW(16, 16, 200)
if a.cache and os.path.exists(a.cache):
    from hnsw_amd.flatgraph import FlatGraph
    h.import_flat(FlatGraph.from_npz(np.load(a.cache)), base)
else:
    h.init(base.shape[1], a.n)
    if a.build_keys == "float":
        h.set_build_float_keys(True)
    h.set_build_distance(a.dist)
    t = time.time(); h.insert_batch(base, threads=a.threads); print(f"build {time.time()-t:.1f}s", flush=True)
from hnsw_amd import capi  # noqa: E402
h.flags = (capi.DIST_SCALAR if a.dist == "scalar" else capi.DIST_SSE) | (capi.KEYS_FLOAT if a.keys == "float" else capi.KEYS_INT32)
h.sync_device(0)
if a.want_visit:
    os.environ["HNSW_AMD_WANT_VISIT"] = str(a.want_visit)
    h.reload_env()
for rep in range(2):
    t = time.time()
    ids, ds, cnt, st = h.search_batch(q, a.ef, out_k=10, stats=True)
    dt = time.time() - t
names = [("n_dist", "entry point + upper layers"), ("sum_degu", "candidate pop"), ("n_expand", "adjacency wait + visited"),
         ("n_sweeps", "row gather + distance"), ("sum_deg0", "admission"), ("n_visited0", "sort + output"), ("n_visited_up", "TOTAL")]
tot = st["n_visited_up"].astype(np.float64).mean()
print(f"host wall {dt*1e3:.2f} ms for {a.nq} queries; expansions/query {st['max_c'].mean():.1f}")
for f, label in names:
    v = st[f].astype(np.float64).mean()
    print(f"{label:22s} {v:12.0f} cycles/query  {100*v/tot:5.1f}%   per expansion {v/st['max_c'].mean():8.0f}")
