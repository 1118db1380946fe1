#!/usr/bin/env python3
"""Rate of the exact brute-force k-NN (f32 matrix-core tile kernels) on random data.

    python tools/bf_probe.py --n 1000000 --nq 10000 --d 128 --reps 3
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hnsw_amd import harness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--integers", action="store_true", help="integer-valued rows in [0, 218] like SIFT")
args = ap.parse_args()

g = torch.Generator(device="cuda").manual_seed(1)
if args.integers:
    base = torch.randint(0, 219, (args.n, args.d), device="cuda", generator=g).float()
    q = torch.randint(0, 219, (args.nq, args.d), device="cuda", generator=g).float()
else:
    base = torch.rand((args.n, args.d), device="cuda", generator=g)
    q = torch.rand((args.nq, args.d), device="cuda", generator=g)
for r in range(args.reps):
    dist, ids, info = harness.exact_topk_mfma(base, q, args.k)
    print(f"rep {r}: {info}", flush=True)
# spot check of a few queries against a plain scan
sel = torch.arange(0, args.nq, max(1, args.nq // 16), device="cuda")
dd = torch.cdist(q[sel].double(), base.double()).pow(2)
ref = torch.topk(dd, args.k, dim=1, largest=False)
got = torch.gather(dd, 1, ids[sel])
bad = int((got.sort(dim=1).values - ref.values).abs().gt(1e-6 * ref.values.abs().clamp(min=1.0)).sum())
print("spot check against a float64 scan:", "ok" if bad == 0 else f"{bad} entries differ")
sys.exit(1 if bad else 0)
