#!/usr/bin/env python3
"""The sharded layout of the DEEP-96 configuration on ONE GPU: S independently built sub-graphs over contiguous slices of
the base set, every query searched in every shard, per-shard top-k lists merged on the device (the kernels the
multi-GPU path runs, without the all-gather: the lists are already in one buffer).  288 GB of HBM hold all shards.

    python tools/deep_sharded_one_gpu.py --shards 8 --rows-per-shard 6250000 --ef 10,20,30

Prints one JSON line per ef: build time per shard, recall@10 of the merged result against exact ground truth over the whole
base, and the time of one pass (S searches + merge) — one GPU doing the work of S.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hnsw_amd import HNSW, capi, datagen, harness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="deep")
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--rows-per-shard", type=int, default=6_250_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ef", default="30", help="comma-separated: one result line per value")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--build-keys", default="float", choices=["int32", "float"])
ap.add_argument("--build", default="device", choices=["gpu", "device"], help="gpu: the host links (assisted build); device: layer 0 linked on the GPU")
args = ap.parse_args()

S, ns, nq, k = args.shards, args.rows_per_shard, args.nq, args.k
dev = torch.device("cuda:0")
q = datagen.make(args.kind, nq, seed=1, stream=1)
d = q.shape[1]
q_t = torch.from_numpy(q).to(dev)
offsets = np.arange(S, dtype=np.uint32) * np.uint32(ns)
shards, build_s = [], []
gt_d = torch.full((nq, k), float("inf"), device=dev)
gt_i = torch.zeros((nq, k), dtype=torch.int64, device=dev)
for s in range(S):
    t0 = time.time()
    rows = datagen.make(args.kind, ns, seed=1, stream=0, row0=s * ns, threads=args.threads)
    h = HNSW(16, 16, 200)
    h.init(d, ns)
    if args.build_keys == "float":
        h.set_build_float_keys(True)
    t1 = time.time()
    h.insert_batch(rows, threads=args.threads, gpu=("link" if args.build == "device" else True))
    build_s.append(round(time.time() - t1, 1))
    h.sync_device(0)
    # exact ground truth of this slice, folded into the running global top-k
    rows_t = torch.from_numpy(rows).to(dev)
    dd, ii, _ = harness.exact_topk_mfma(rows_t, q_t, k, id_offset=0)
    del rows_t
    cat_d = torch.cat([gt_d, dd], 1)
    cat_i = torch.cat([gt_i, ii + s * ns], 1)
    gt_d, sel = torch.topk(cat_d, k, dim=1, largest=False)
    gt_i = torch.gather(cat_i, 1, sel)
    shards.append(h)
    del rows
    print(f"shard {s}: data + build + ground truth in {time.time() - t0:.0f}s (build {build_s[-1]}s)", file=sys.stderr, flush=True)

gathered = torch.empty((S, 2, nq, k), dtype=torch.int32, device=dev)
out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
flags = capi.DIST_SSE | capi.KEYS_INT32


def one_pass(ef):
    for s, h in enumerate(shards):
        h.search_batch_device(q_t.data_ptr(), nq, ef, k, gathered[s, 1].data_ptr(), gathered[s, 0].data_ptr(), 0, 0, stream, flags)
    capi.check(capi.lib().hnsw_amd_merge_topk_device(gathered.data_ptr(), gathered.data_ptr() + 4 * nq * k, capi.u32(offsets), S, nq, k, k,
                                                     2 * nq * k, out_i.data_ptr(), out_d.data_ptr(), stream))


side_streams = [torch.cuda.Stream() for _ in range(2)]


def one_pass_overlapped(ef):
    """The same pass with two shard searches in flight (the call in two halves, a stream each): the next shard's queries
    take the slots the last queries of the previous one leave idle."""
    torch.cuda.synchronize()
    calls = []
    for s, h in enumerate(shards):
        if len(calls) == 2:
            hh, c = calls.pop(0)
            hh.search_batch_device_finish(c)
        calls.append((h, h.search_batch_device_begin(q_t.data_ptr(), nq, ef, k, gathered[s, 1].data_ptr(), gathered[s, 0].data_ptr(), 0, 0,
                                                     side_streams[s % 2].cuda_stream, flags)))
    for hh, c in calls:
        hh.search_batch_device_finish(c)
    torch.cuda.synchronize()
    capi.check(capi.lib().hnsw_amd_merge_topk_device(gathered.data_ptr(), gathered.data_ptr() + 4 * nq * k, capi.u32(offsets), S, nq, k, k,
                                                     2 * nq * k, out_i.data_ptr(), out_d.data_ptr(), stream))


gt = gt_i.cpu().numpy()
for ef in [int(x) for x in args.ef.split(",")]:
    for _ in range(3):
        one_pass(ef)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass(ef)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    got = out_i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    recall = harness.recall_at_k(got, gt, k)
    for _ in range(2):
        one_pass_overlapped(ef)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass_overlapped(ef)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / args.steps * 1e3
    same = bool(np.array_equal(got, out_i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF))
    print(json.dumps({"workload": f"{args.kind}-{d} synthetic, {S} shards x {ns} rows = {S * ns} rows, nq={nq}, ef={ef}, k={k}, keys=int32, "
                                  f"graphs built {'with layer 0 linked on the GPU' if args.build == 'device' else 'the GPU-assisted way'} ({args.build_keys} build keys), all shards on one GPU",
                      "ef": ef, "build_s_per_shard": build_s, "recall_at_10_merged": round(float(recall), 4),
                      "ms_per_pass_one_gpu_all_shards": round(ms, 3), "queries_per_s_one_gpu_all_shards": round(nq / ms * 1e3, 1),
                      "two_shard_searches_in_flight": {"ms_per_pass": round(ms2, 3), "queries_per_s": round(nq / ms2 * 1e3, 1), "same_result": same},
                      "note": "one pass = S shard searches + device merge on one GPU; S GPUs run the searches side by side and add an "
                              "all-gather of 80 KB x S per 1000 queries"}), flush=True)
