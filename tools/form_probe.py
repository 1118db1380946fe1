#!/usr/bin/env python3
"""Kernel time of the batched search under each form of the visited set (environment switches of the
launcher), device buffers, HIP events through torch.  Usage: form_probe.py --kind deep --n 2000000 --ef 30
[--rb-min 8] [--forms 32,16,B,auto]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hnsw_amd import HNSW, capi, datagen, harness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="deep")
ap.add_argument("--n", type=int, default=2_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ef", default="30")
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--build-keys", default="float")
ap.add_argument("--rb-min", default="")
ap.add_argument("--forms", default="auto,32,16,B")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dist", default="sse", choices=["sse", "scalar"])
ap.add_argument("--keys", default="int32", choices=["int32", "float"])
a = ap.parse_args()
d = datagen.KINDS[a.kind]
base = datagen.make(a.kind, a.n, seed=1)
q = datagen.make(a.kind, a.nq, seed=1, stream=1)
h = HNSW(16, 16, 200)
h.init(d, a.n)
if a.build_keys == "float":
    h.set_build_float_keys(True)
t = time.time()
for r0 in range(0, a.n, 500_000):
    h.insert_batch(base[r0:r0 + 500_000], threads=a.threads)
    print(f"inserted {min(a.n, r0 + 500_000)} in {time.time() - t:.0f}s", flush=True)
h.sync_device(0)
q_t = torch.from_numpy(q).cuda()
nq = a.nq
ids = torch.empty((nq, 10), dtype=torch.int32, device="cuda")
ds = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
st_t = torch.empty((nq, 8), dtype=torch.int32, device="cuda")
flags = (capi.DIST_SCALAR if a.dist == "scalar" else capi.DIST_SSE) | (capi.KEYS_FLOAT if a.keys == "float" else capi.KEYS_INT32)
stream = torch.cuda.current_stream().cuda_stream
if a.rb_min:
    os.environ["HNSW_AMD_VIS_RB_MIN"] = a.rb_min
ENV = {"32": "HNSW_AMD_VISITED32", "16": "HNSW_AMD_VISITED16", "B": "HNSW_AMD_VISITEDB"}
for ef in [int(x) for x in a.ef.split(",")]:
    ref = None
    for form in a.forms.split(","):
        for v in list(ENV.values()) + ["HNSW_AMD_BULK_FRAC", "HNSW_AMD_NO_SPILL"]:
            os.environ.pop(v, None)
        name, _, opt = form.partition(":")           # e.g. auto:0.25 = bulk fraction 0.25, auto:nospill
        if name in ENV:
            os.environ[ENV[name]] = "1"
        if opt == "nospill":
            os.environ["HNSW_AMD_NO_SPILL"] = "1"
        elif opt:
            os.environ["HNSW_AMD_BULK_FRAC"] = opt
        try:
            os.environ["HNSW_AMD_VERBOSE"] = "1"
            h.reload_env()
            for _ in range(3):      # discovery, sized from observation, bulk sizing
                retried = h.search_batch_device(q_t.data_ptr(), nq, ef, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), st_t.data_ptr(), stream, flags)
            os.environ.pop("HNSW_AMD_VERBOSE")
            h.reload_env()
            retried = h.search_batch_device(q_t.data_ptr(), nq, ef, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), st_t.data_ptr(), stream, flags)
            torch.cuda.synchronize()
            st = st_t.cpu().numpy().view(np.uint32).reshape(nq, 8).copy().view(capi.STATS_DTYPE).reshape(nq)
            byts = int(harness.algorithmic_bytes(st, d, d if a.dist == "scalar" else d & ~15, 10, 4).sum())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                h.search_batch_device(q_t.data_ptr(), nq, ef, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, stream, flags)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            res = (ids.cpu().numpy().copy(), ds.cpu().numpy().copy())
            same = "" if ref is None else f" same_as_first={np.array_equal(res[0], ref[0]) and np.array_equal(res[1], ref[1])}"
            ref = ref or res
            print(f"RESULT kind={a.kind} n={a.n} ef={ef} form={form}: {ms:.4f} ms, {byts / ms / 1e6:.1f} GB/s algorithmic, retried {retried}, "
                  f"visited max {st['n_visited0'].max()} mean {st['n_visited0'].mean():.0f}, C max {st['max_c'].max()}{same}", flush=True)
        except capi.HnswAmdError as e:
            print(f"RESULT kind={a.kind} ef={ef} form={form}: {e}", flush=True)
