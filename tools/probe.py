#!/usr/bin/env python3
"""Kernel A/B probe: one index (built GPU-assisted, or loaded from --cache), a list of ef values, event-timed batches;
optionally each ef is also run with an environment switch set (--ab HNSW_AMD_NO_DOT4 ...) and the two results are
compared bit for bit.  Prints one RESULT line per (ef, variant)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hnsw_amd import HNSW, capi, datagen, harness  # noqa: E402
from hnsw_amd.flatgraph import FlatGraph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="sift")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ef", default="50")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--rows", default="f32", choices=["f32", "u8", "q8"])
ap.add_argument("--dist", default="sse", choices=["sse", "scalar"])
ap.add_argument("--keys", default="int32", choices=["int32", "float"])
ap.add_argument("--build-keys", default="int32", choices=["int32", "float"])
ap.add_argument("--cache", default="")
ap.add_argument("--ab", default="", help="comma separated environment switches: each ef is run once more with each of them set")
ap.add_argument("--tag", default="")
a = ap.parse_args()

base = datagen.make(a.kind, a.n, seed=1)
q = datagen.make(a.kind, a.nq, seed=1, stream=1)
d = base.shape[1]
h = HNSW(16, 16, 200)
t = time.time()
if a.cache and os.path.exists(a.cache):
    h.import_flat(FlatGraph.from_npz(np.load(a.cache)), base)
else:
    h.init(d, a.n)
    if a.build_keys == "float":
        h.set_build_float_keys(True)
    h.set_build_distance(a.dist)
    h.insert_batch(base, threads=16, gpu=True)
    if a.cache:
        np.savez(a.cache, **h.export_flat().to_npz_dict())
print(f"index ready in {time.time() - t:.1f}s", flush=True)
h.sync_device(0)
flags = (capi.DIST_SCALAR if a.dist == "scalar" else capi.DIST_SSE) | (capi.KEYS_FLOAT if a.keys == "float" else capi.KEYS_INT32)
flags |= {"f32": 0, "u8": capi.ROWS_U8, "q8": capi.ROWS_Q8}[a.rows]
q_t = torch.from_numpy(q).cuda()
nq, k = a.nq, 10
ids = torch.empty((nq, k), dtype=torch.int32, device="cuda")
ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
st_t = torch.empty((nq, 8), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
variants = [""] + [v for v in a.ab.split(",") if v]
for ef in [int(x) for x in a.ef.split(",")]:
    ref = None
    for var in variants:
        for v in variants[1:]:
            os.environ.pop(v, None)
        if var:
            os.environ[var] = "1"
        h.reload_env()
        for _ in range(3):
            retried = h.search_batch_device(q_t.data_ptr(), nq, ef, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), st_t.data_ptr(), stream, flags)
        torch.cuda.synchronize()
        st = st_t.cpu().numpy().view(np.uint32).reshape(nq, 8).copy().view(capi.STATS_DTYPE).reshape(nq)
        byts = int(harness.algorithmic_bytes(st, d, d if a.dist == "scalar" else d & ~15, k, 1 if a.rows != "f32" else 4).sum())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            h.search_batch_device(q_t.data_ptr(), nq, ef, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, stream, flags)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        res = (ids.cpu().numpy().copy(), ds.cpu().numpy().view(np.uint32).copy(), cnt.cpu().numpy().copy())
        same = "" if ref is None else f" same_as_first={all(np.array_equal(x, y) for x, y in zip(res, ref))}"
        ref = ref or res
        info = h.last_launch_info()
        print(f"RESULT {a.tag} kind={a.kind} n={a.n} rows={a.rows} dist={a.dist} keys={a.keys} ef={ef} {var or 'default'}: {ms:.4f} ms, "
              f"{nq / ms / 1e3:.2f} M queries/s, {byts / ms / 1e6:.0f} GB/s algorithmic, retried {retried}, per CU {info['resident_queries_per_cu']}, "
              f"LDS {info['lds_bytes_per_query']}, {info['visited_form']}{same}", flush=True)
