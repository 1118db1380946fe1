#!/usr/bin/env python3
"""Rate of the brute-force distance kernel against the row length (random data):  python tools/bf_rate.py [d ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hnsw_amd import harness  # noqa: E402

dims = [int(x) for x in sys.argv[1:]] or [96, 100, 104, 112, 128, 132, 160, 164]
n, nq = 1_000_000, 10_000
for d in dims:
    g = torch.Generator(device="cuda").manual_seed(d)
    b = torch.rand((n, d), device="cuda", generator=g)
    q = torch.rand((nq, d), device="cuda", generator=g)
    for _ in range(2):
        _, _, info = harness.exact_topk_mfma(b, q, 11)
    print(d, info, "ms per dim", round(info["mfma_kernel_ms"] / d, 4), flush=True)
    del b, q
