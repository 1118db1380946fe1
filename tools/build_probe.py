#!/usr/bin/env python3
"""Build time and graph quality of the host-only concurrent build against the GPU-assisted build.

    python tools/build_probe.py --kind sift --n 1000000 --threads 16 --ef 50
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hnsw_amd import HNSW, datagen, harness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="sift")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--ef", default="30,50,100")
ap.add_argument("--modes", default="host,gpu")
ap.add_argument("--build-keys", default="int32", choices=["int32", "float"])
args = ap.parse_args()

base = datagen.make(args.kind, args.n, seed=1)
q = datagen.make(args.kind, args.nq, seed=1, stream=1)
d = base.shape[1]
_, gt, _ = harness.exact_topk_mfma(torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda(), 10)
gt = gt.cpu().numpy()
for mode in args.modes.split(","):
    h = HNSW(16, 16, 200)
    h.init(d, args.n)
    if args.build_keys == "float":
        h.set_build_float_keys(True)
    t0 = time.time()
    h.insert_batch(base, threads=args.threads, gpu=(mode == "gpu"))
    dt = time.time() - t0
    line = f"RESULT kind={args.kind} n={args.n} build={mode} threads={args.threads}: {dt:.1f} s"
    for ef in [int(x) for x in args.ef.split(",")]:
        ids, _, _ = h.search_batch(q, ef, out_k=10)
        line += f", recall@10(ef={ef}) {harness.recall_at_k(ids.astype(np.int64), gt, 10):.4f}"
    print(line, flush=True)
    del h
