#!/usr/bin/env python3
"""Randomised parity campaign: random index shapes, build modes, search flags, ef and batch sizes; the GPU search
against the CPU restatement (results and counters) on the graph that was built.  Every case is reproducible from its
printed seed.

    python tools/fuzz_parity.py --seconds 300 [--seed 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from hnsw_amd import HNSW, capi, datagen  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.helpers import assert_same_results  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-n", type=int, default=60000)
ap.add_argument("--max-nq", type=int, default=3000)
args = ap.parse_args()

rng = np.random.default_rng(args.seed)
t_end = time.time() + args.seconds
cases = fails = 0
kinds = {}
while time.time() < t_end:
    case_seed = int(rng.integers(1, 1 << 30))
    r = np.random.default_rng(case_seed)
    unit = bool(r.integers(0, 2))
    d = int(r.choice([16, 24, 32, 48, 64, 96, 100, 112, 128, 144, 200, 256, 272, 384, 784, 960]))
    n_cap = args.max_n if d <= 256 else args.max_n // 6
    n = int(np.exp(r.uniform(np.log(300), np.log(n_cap))))
    M = int(r.choice([5, 8, 16, 16, 16, 24]))
    efc = int(r.choice([40, 100, 200]))
    dist = "scalar" if r.random() < 0.3 else "sse"
    build = str(r.choice(["seq", "host", "gpu", "link"])) if n >= 2100 else "seq"
    if build == "seq" and n > 8000:
        build = "host"
    nq = int(np.exp(r.uniform(0, np.log(args.max_nq))))
    ef = int(np.exp(r.uniform(0, np.log(300))))
    keys_float = bool(r.integers(0, 2))
    std_heap = r.random() < 0.2
    rows = "f32"
    if not unit and dist == "sse" and r.random() < 0.4:
        rows = "u8"
    out_k = ef if r.random() < 0.5 else int(r.integers(1, ef + 1))
    ip = unit and dist == "sse" and rows == "f32" and r.random() < 0.2      # this library's inner-product metric, on unit vectors
    desc = dict(seed=case_seed, unit=unit, d=d, n=n, M=M, efc=efc, dist=dist, build=build, nq=nq, ef=ef, keys_float=keys_float,
                std_heap=std_heap, rows=rows, out_k=out_k, ip=ip)
    base = datagen.make("deep" if unit else "sift", n, seed=case_seed % 1000 + 1, d=d)
    q = datagen.make("deep" if unit else "sift", nq, seed=case_seed % 1000 + 1, stream=1, d=d)
    if r.random() < 0.15:
        q = q + np.float32(0.25)                  # queries that are not integer-valued (the byte-row arithmetic must notice)
    try:
        h = HNSW(M, M, efc)
        h.init(d, n)
        h.set_build_distance(dist)
        if unit:
            h.set_build_float_keys(True)
        if build == "seq":
            h.insert_batch(base, threads=1)
        else:
            h.insert_batch(base, threads=8, gpu={"host": False, "gpu": True, "link": "link"}[build])
        flags = (capi.DIST_SCALAR if dist == "scalar" else capi.DIST_SSE) | (capi.KEYS_FLOAT if keys_float else capi.KEYS_INT32)
        flags |= capi.ROWS_U8 if rows == "u8" else 0
        flags |= capi.W_STD_HEAP if std_heap else 0
        flags |= capi.METRIC_IP if ip else 0
        ids, ds, cnt, st = h.search_batch(q, ef, out_k=out_k, flags=flags, stats=True)
        if r.random() < 0.5:
            # once more: the second batch is sized from what the first one needed (other forms of the visited set)
            ids2, ds2, cnt2 = h.search_batch(q, ef, out_k=out_k, flags=flags)
            assert np.array_equal(ids, ids2) and np.array_equal(ds.view(np.uint32), ds2.view(np.uint32)) and np.array_equal(cnt, cnt2), "second call differs"
        g = h.export_flat()
        if build != "seq":
            # a concurrent or GPU build: the lists are well formed whatever the interleaving was
            adj = g.adj0.reshape(n, g.deg0)
            live = np.arange(g.deg0)[None, :] < g.cnt0[:, None]
            assert (adj[live] < n).all() and (adj[~live] == 0xFFFFFFFF).all(), "layer 0: entries out of range or padding broken"
            assert g.cnt0.min() >= 1, "a vector without links"
            assert not (adj == np.arange(n, dtype=np.uint32)[:, None]).any(), "self loop"
            srt = np.sort(np.where(live, adj.astype(np.int64), -1 - np.arange(g.deg0)[None, :]), axis=1)
            assert (srt[:, 1:] != srt[:, :-1]).all(), "duplicate neighbour"
        idx = oracle.OracleIndex(g, base)
        oi, od, oc, ost = idx.search(q, ef, out_k=out_k, dist_mode=oracle.DIST_IP if ip else (oracle.DIST_SCALAR if dist == "scalar" else oracle.DIST_SSE),
                                     key_mode=oracle.KEYS_FLOAT if keys_float else oracle.KEYS_INT32, threads=8, stats=True)
        assert np.array_equal(cnt, oc), "counts differ"
        assert_same_results(ids, ds, oi, od, what="fuzz", last_group_open=not (std_heap and out_k == ef))
        for f in ("n_dist", "n_expand"):
            if f in st.dtype.names and f in ost.dtype.names:
                assert np.array_equal(st[f], ost[f]), f"counter {f} differs"
        del h
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", json.dumps(desc), "->", repr(e)[:300], flush=True)
    cases += 1
    kinds[build] = kinds.get(build, 0) + 1
    if cases % 20 == 0:
        print(f"{cases} cases, {fails} failures, {int(t_end - time.time())} s left", flush=True)
print(json.dumps({"cases": cases, "failures": fails, "by_build": kinds, "seed": args.seed, "seconds": args.seconds}))
sys.exit(1 if fails else 0)
