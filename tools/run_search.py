#!/usr/bin/env python3
"""Minimal driver for profiling: build (or load) an index and run the batched search a few
times, nothing else on the GPU (no torch).  Usage: run_search.py [--n N] [--ef EF] [--reps R]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hnsw_amd import HNSW, datagen  # noqa: E402
from hnsw_amd.flatgraph import FlatGraph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ef", default="50", help="one ef or a comma separated list")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--cache", default="")
ap.add_argument("--kind", default="sift")
ap.add_argument("--keys", default="int32")
ap.add_argument("--rows", default="f32", choices=["f32", "u8"])
ap.add_argument("--build-keys", default="int32", choices=["int32", "float"])
ap.add_argument("--dist", default="sse", choices=["sse", "scalar"])
a = ap.parse_args()
bcache = (a.cache + ".base.npy") if a.cache else ""
if bcache and os.path.exists(bcache):
    base = np.load(bcache)                       # (generating 10^7 rows takes longer than reading them back)
else:
    base = datagen.make(a.kind, a.n, seed=1)
    if bcache and base.nbytes > (1 << 30):
        np.save(bcache, base)
q = datagen.make(a.kind, a.nq, seed=1, stream=1)
h = HNSW(16, 16, 200)
if a.cache and os.path.exists(a.cache):
    h.import_flat(FlatGraph.from_npz(np.load(a.cache)), base)
else:
    h.init(base.shape[1], a.n)
    if a.build_keys == "float":
        h.set_build_float_keys(True)
    h.set_build_distance(a.dist)
    t = time.time(); h.insert_batch(base, threads=a.threads); print(f"build {time.time()-t:.1f}s", flush=True)
    if a.cache:
        np.savez(a.cache, **h.export_flat().to_npz_dict())
h.sync_device(0)
from hnsw_amd import capi  # noqa: E402
for ef in [int(x) for x in a.ef.split(",")]:
  print(f"ef={ef}", flush=True)
  for rep in range(a.reps):
    t = time.time()
    h.flags = (capi.DIST_SCALAR if a.dist == "scalar" else capi.DIST_SSE) | (capi.KEYS_FLOAT if a.keys == "float" else capi.KEYS_INT32) | (capi.ROWS_U8 if a.rows == "u8" else 0)
    ids, ds, cnt, st = h.search_batch(q, ef, out_k=10, stats=True)
    if rep == a.reps - 1:
        from hnsw_amd import harness
        d_ = base.shape[1]
        print("algorithmic_bytes_per_launch", int(harness.algorithmic_bytes(st, d_, d_ if a.dist == "scalar" else d_ & ~15, 10, 1 if a.rows == "u8" else 4).sum()),
              "launch", h.last_launch_info(), flush=True)
    print(f"rep {rep}: host wall {1e3*(time.time()-t):.2f} ms", flush=True)
  print("n_dist/q", st["n_dist"].mean(), "expansions/q", st["n_expand"].mean())
  for f in ("max_c", "n_visited0", "n_expand", "n_dist"):
    v = st[f]
    print(f, "mean", v.mean(), "p50", np.percentile(v, 50), "p99", np.percentile(v, 99), "max", v.max())
