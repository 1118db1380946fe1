#!/usr/bin/env python3
"""Does the order of the queries in a batch matter?  Times the batched search with the queries as given (random), grouped by
their nearest of C sampled base rows, and grouped + dealt out so that consecutive groups land on the same XCD
(workgroup b runs on XCD b mod 8).

    python tools/order_probe.py --kind deep --n 2000000 --ef 30
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hnsw_amd import HNSW, capi, datagen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="deep")
ap.add_argument("--n", type=int, default=2_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ef", type=int, default=30)
ap.add_argument("--centroids", type=int, default=1024)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--build-keys", default="float", choices=["int32", "float"])
args = ap.parse_args()

base = datagen.make(args.kind, args.n, seed=1)
q = datagen.make(args.kind, args.nq, seed=1, stream=1)
d = base.shape[1]
h = HNSW(16, 16, 200)
h.init(d, args.n)
if args.build_keys == "float":
    h.set_build_float_keys(True)
t0 = time.time()
h.insert_batch(base, threads=args.threads, gpu=True)
print(f"built in {time.time() - t0:.1f}s", flush=True)
h.sync_device(0)
dev = torch.device("cuda:0")
nq, k = args.nq, 10
ids_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
flags = capi.DIST_SSE | capi.KEYS_INT32

cent = torch.from_numpy(base[np.random.default_rng(3).choice(args.n, args.centroids, replace=False)]).to(dev)
q_all = torch.from_numpy(q).to(dev)
key = torch.cdist(q_all, cent).argmin(1).cpu().numpy()
grouped = np.argsort(key, kind="stable")
# deal the grouped order out over the 8 XCDs: workgroup b = 8 * i + x takes element x * (nq / 8) + i
per = (nq + 7) // 8
b = np.arange(nq)
dealt_pos = (b % 8) * per + b // 8
dealt = grouped[np.minimum(dealt_pos, nq - 1)] if nq % 8 else grouped[dealt_pos]
orders = {"as given": np.arange(nq), "grouped": grouped, "grouped, one range per XCD": dealt}
ref = None
for name, order in orders.items():
    qt = torch.from_numpy(np.ascontiguousarray(q[order])).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        h.search_batch_device(qt.data_ptr(), nq, args.ef, k, ids_t.data_ptr(), dist_t.data_ptr(), 0, 0, stream, flags)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        h.search_batch_device(qt.data_ptr(), nq, args.ef, k, ids_t.data_ptr(), dist_t.data_ptr(), 0, 0, stream, flags)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    got = np.empty((nq, k), np.int32)
    got[order] = ids_t.cpu().numpy()
    if ref is None:
        ref = got
    print(f"RESULT kind={args.kind} n={args.n} ef={args.ef} order={name}: {ms:.4f} ms per batch, same results: {bool((got == ref).all())}", flush=True)
