#!/usr/bin/env python3
"""Headline benchmark: batched HNSW k-NN search, QPS at recall@10 >= 0.95 on a SIFT-128-shaped
1M-vector set (synthetic), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the search over the whole query batch (10 000 queries resident in HBM).
N > 1: the base set is split into N contiguous shards, each rank builds and searches its own
sub-graph, the per-shard top-k lists are all-gathered over RCCL and merged on the device.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (spec)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--base-size", dest="n", type=int, default=1_000_000, help="base vectors (whole job)")
    ap.add_argument("--queries", dest="nq", type=int, default=10_000)
    ap.add_argument("--kind", default="sift")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=0, help="0 = smallest ef of the reference sweep with recall@10 >= target")
    ap.add_argument("--target-recall", type=float, default=0.95)
    ap.add_argument("--build-threads", type=int, default=0, help="0 = this rank's share of the usable host cores")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "port", "reference", "none"],
                    help="auto: the unmodified reference binary (oracle/_ref) when present, else the CPU restatement")
    ap.add_argument("--cpu-sample", type=int, default=4000)
    ap.add_argument("--keys", default="int32", choices=["int32", "float"])
    ap.add_argument("--dist", default="sse", choices=["sse", "scalar"],
                    help="sse: the shipped reference's distance (first 16*floor(d/16) dims); scalar: its -DNO_MANUAL_VECTORIZATION build (all dims)")
    ap.add_argument("--rows", default="f32", choices=["f32", "u8", "q8"],
                    help="u8: search the one-byte-per-dimension copy of the base (integer-valued data only; results unchanged); "
                         "q8: search a quantised byte copy of any base, the ef results re-scored with the float rows (approximate). "
                         "The algorithmic bytes are then counted at 1 byte per dimension (4 for the re-scored rows)")
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replicated"],
                    help="N > 1: sharded = the base split into N sub-graphs, every query on every GPU, all-gather + merge "
                         "(fixed total work); replicated = the whole index on every GPU, each with its own batch of "
                         "--queries queries and no collective in the data path (work grows with N)")
    ap.add_argument("--no-alongside", action="store_true",
                    help="N > 1, sharded mode: skip the replicated-index measurement reported next to the sharded one")
    ap.add_argument("--build", default="host", choices=["host", "gpu", "device"],
                    help="host: concurrent build on the host's cores; gpu: the construction searches on the GPU (a few times faster; "
                         "same construction rule, slightly different graph); device: selection and linking of layer 0 on the GPU as well")
    ap.add_argument("--build-keys", default="int32", choices=["int32", "float"],
                    help="float: the construction searches use float candidate keys (a regular HNSW graph, not the reference's; "
                         "several times faster to build on unit-norm data)")
    ap.add_argument("--sweep", default="", help="comma-separated ef values: report recall and QPS for each instead of one operating point")
    ap.add_argument("--preset", default="", choices=["", "sift1m", "deep100m"],
                    help="a named configuration: sift1m = the defaults; deep100m = DEEP-96-shaped 100M base split over the ranks "
                         "(--gpus 8: 12.5M rows per GPU), build with layer 0 searched and linked on the GPU, float construction keys, reference int32 search keys, "
                         "no replicated leg.  --base-size after the preset scales it down for rehearsals")
    args = ap.parse_args()
    if args.preset == "deep100m":
        defaults = ap.parse_args([])
        for name, value in (("kind", "deep"), ("n", 100_000_000), ("build", "device"), ("build_keys", "float"), ("no_alongside", True),
                            ("cpu_baseline", "none")):
            if getattr(args, name) == getattr(defaults, name):      # (an explicit flag wins over the preset)
                setattr(args, name, value)

    import torch
    import torch.distributed as dist

    from hnsw_amd import HNSW, capi, datagen, harness


    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the search has no CPU path)")
    # one GPU per rank; HNSW_AMD_BENCH_BACKEND=gloo lets several ranks share a GPU (rehearsal on a 1-GPU box)
    backend = os.environ.get("HNSW_AMD_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # small tensors that only travel through collectives: on the GPU for RCCL, on the host for gloo (gloo's own
    # handling of device tensors is not used)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import datetime
        # (index builds of different length run between collectives: a generous limit for the slowest rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=45))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=45))

    if args.build_threads <= 0:
        # every rank builds at the same time on the one host: share its cores
        args.build_threads = max(1, harness.usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    d = datagen.KINDS[args.kind]
    k = args.k
    # ---- shard of the base set, queries ----
    parts = world if args.mode == "sharded" else 1        # sub-graphs the base is split into
    prank = rank if args.mode == "sharded" else 0
    lo = prank * args.n // parts
    hi = (prank + 1) * args.n // parts
    t0 = time.time()
    base = datagen.make(args.kind, hi - lo, seed=1, stream=0, row0=lo, threads=args.build_threads)
    # replicated: every rank has its own batch of queries (rows rank*nq .. of the query stream)
    queries = datagen.make(args.kind, args.nq, seed=1, stream=1, row0=0 if parts == world else rank * args.nq)
    log(f"rank {rank}: data {base.shape} in {time.time() - t0:.1f}s")

    # ---- build (host, concurrent inserts) or load from the cache directory ----
    h = HNSW(16, 16, 200, device=local_rank)
    cache = os.environ.get("HNSW_AMD_BENCH_CACHE")
    cpath = os.path.join(cache, f"{args.kind}_n{args.n}_s{parts}_r{prank}{'_fk' if args.build_keys == 'float' else ''}{'_gb' if args.build == 'gpu' else ('_dl' if args.build == 'device' else '')}{'_sc' if args.dist == 'scalar' else ''}.npz") if cache else None
    t0 = time.time()
    if cpath and os.path.exists(cpath):
        from hnsw_amd.flatgraph import FlatGraph
        h.import_flat(FlatGraph.from_npz(np.load(cpath)), base)
        log(f"rank {rank}: graph loaded from {cpath} in {time.time() - t0:.1f}s")
    else:
        h.init(d, hi - lo)
        if args.build_keys == "float":
            h.set_build_float_keys(True)
        h.set_build_distance(args.dist)           # the same reference build for insert() and for the search
        # in pieces, so a long build keeps reporting progress (larger ones when layer 0 lives on the device: every call
        # hands it over and takes it back)
        step_rows = 2_500_000 if args.build == "device" else 500_000
        for r0 in range(0, hi - lo, step_rows):
            h.insert_batch(base[r0:r0 + step_rows], threads=args.build_threads, gpu=("link" if args.build == "device" else args.build == "gpu"))
            if hi - lo > step_rows:
                log(f"rank {rank}: inserted {min(hi - lo, r0 + step_rows)} / {hi - lo} in {time.time() - t0:.0f}s")
        log(f"rank {rank}: built {hi - lo} vectors with {args.build_threads} threads in {time.time() - t0:.1f}s")
        if cpath:
            os.makedirs(cache, exist_ok=True)
            np.savez(cpath, **h.export_flat().to_npz_dict())
    build_s = time.time() - t0
    h.sync_device(local_rank)
    # From here on (the host build above can take long and reports its own progress) a rank that stops making
    # progress prints every thread's stack and exits instead of hanging the job.
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("HNSW_AMD_BENCH_WATCHDOG", "1500")), exit=True)

    # ---- device buffers ----
    q_t = torch.from_numpy(queries).to(dev)
    nq = args.nq
    EFMAX = 200
    ids_t = torch.empty((nq, EFMAX), dtype=torch.int32, device=dev)
    dist_t = torch.empty((nq, EFMAX), dtype=torch.float32, device=dev)
    cnt_t = torch.empty((nq,), dtype=torch.int32, device=dev)
    stats_t = torch.empty((nq, 8), dtype=torch.int32, device=dev)
    flags = (capi.DIST_SCALAR if args.dist == "scalar" else capi.DIST_SSE) | (capi.KEYS_FLOAT if args.keys == "float" else capi.KEYS_INT32)
    flags |= capi.ROWS_U8 if args.rows == "u8" else (capi.ROWS_Q8 if args.rows == "q8" else 0)
    row_elem_bytes = 4 if args.rows == "f32" else 1
    h.flags = flags
    offsets = np.array([r * args.n // parts for r in range(parts)], dtype=np.uint32)
    out_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    out_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    L = capi.lib()

    from hnsw_amd.sharded import ShardedHNSW
    sh = ShardedHNSW.__new__(ShardedHNSW)      # wrap the index built above
    sh.n_total, sh.d, sh.rank, sh.world, sh.lo, sh.hi = args.n, d, prank, parts, lo, hi
    sh.offsets, sh.group, sh.index = offsets, None, h

    def step(ef, out_k, want_stats=False):
        """One pass over the query batch; leaves the final top-out_k (global ids) in out_ids/out_dist."""
        stream = torch.cuda.current_stream().cuda_stream
        if parts == 1 or want_stats:
            h.search_batch_device(q_t.data_ptr(), nq, ef, out_k, out_ids.data_ptr(), out_dist.data_ptr(), cnt_t.data_ptr(),
                                  stats_t.data_ptr() if want_stats else 0, stream, flags)
            if parts == 1:
                return
        i_t, d_t = sh.search(q_t, ef, out_k)
        out_ids.copy_(i_t); out_dist.copy_(d_t)

    # ---- exact ground truth (brute force on the GPU over every shard, merged) ----
    t0 = time.time()
    base_t = torch.from_numpy(base).to(dev)
    gd, gi, gt_info = harness.exact_topk_mfma(base_t, q_t, k + 1, id_offset=int(offsets[prank]))
    del base_t
    # cross-check of the ground truth on a sample: CPU scan in double precision (N=1 only: needs the whole base)
    if parts == 1 and rank == 0:
        from oracle import oracle as _oracle
        samp = np.linspace(0, args.nq - 1, 16).astype(int)
        cpu_gt = _oracle.bruteforce(base, queries[samp], k, threads=harness.usable_cores()).astype(np.int64)
        gsub = gi[torch.from_numpy(samp).to(dev)].cpu().numpy()[:, :k]
        dd = lambda ids: ((queries[samp][:, None, :].astype(np.float64) - base[ids].astype(np.float64)) ** 2).sum(-1)
        gt_info["cpu_double_check"] = bool(np.array_equal(dd(gsub), dd(cpu_gt)))    # same distances: equal up to exact ties
    if parts > 1:
        gd_c, gi_c = gd.to(cdev), gi.to(cdev)
        all_d = [torch.empty_like(gd_c) for _ in range(world)]
        all_i = [torch.empty_like(gi_c) for _ in range(world)]
        dist.all_gather(all_d, gd_c)
        dist.all_gather(all_i, gi_c)
        cd, ci = torch.cat(all_d, 1).to(dev), torch.cat(all_i, 1).to(dev)
        _, sel = torch.topk(cd, k + 1, dim=1, largest=False)
        gi = torch.gather(ci, 1, sel)
    gt = gi.cpu().numpy()
    log(f"ground truth in {time.time() - t0:.1f}s")

    if args.sweep:
        rows = []
        for ef in [int(x) for x in args.sweep.split(",")]:
            step(ef, k, want_stats=(parts == 1))
            torch.cuda.synchronize()
            ids = out_ids.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            rec = {"ef": ef, "recall_at_10": round(harness.recall_at_k(ids, gt, k), 4)}
            if parts == 1:
                st = stats_t.cpu().numpy().view(np.uint32).reshape(nq, 8).copy().view(capi.STATS_DTYPE).reshape(nq)
                rec["dist_per_query"] = round(float(st["n_dist"].mean()), 1)
                bpl = int(harness.algorithmic_bytes(st, d, d if args.dist == "scalar" else d & ~15, k, row_elem_bytes).sum())
                bpl += 3 * ef * (d & ~15) * nq if args.rows == "q8" else 0      # the re-scored rows are float rows
            t0 = time.perf_counter()
            for _ in range(3):
                step(ef, k)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            rec["qps"] = round(nq / dt, 1)
            if parts == 1:
                rec["algorithmic_gbs"] = round(bpl / dt / 1e9, 1)
            rows.append(rec)
            log(f"sweep {rec}")
        if rank == 0:
            print(json.dumps({"workload": f"{args.kind}-{d} synthetic, n={args.n}, nq={nq}, keys={args.keys}, dist={args.dist}",
                              "n_gpus": world, "build_s": round(build_s, 1), "ground_truth": gt_info, "sweep": rows}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- ef: the reference's sweep, smallest ef reaching the target recall@10 ----
    sweep = []
    ef_pick = args.ef
    # shards see fewer points each, so a sharded run may reach the target below the reference's first value (20)
    # (... and between its first values in steps of two: a shard's search time grows with ef, the merged recall with it)
    sweep_values = harness.REFERENCE_EF_SWEEP if parts == 1 else sorted(set(list(range(10, 40, 2)) + list(harness.REFERENCE_EF_SWEEP)))
    for ef in ([args.ef] if args.ef else sweep_values):
        step(ef, k)
        torch.cuda.synchronize()
        ids = out_ids.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        r = harness.recall_at_k(ids, gt, k)
        sweep.append((ef, round(r, 4)))
        if not args.ef and r >= args.target_recall:
            ef_pick = ef
            break
    if not ef_pick:
        ef_pick = harness.REFERENCE_EF_SWEEP[-1]
    ef = ef_pick
    log(f"ef sweep {sweep} -> ef={ef}")

    # ---- per-query counters at the operating point -> algorithmic bytes per launch ----
    step(ef, k, want_stats=True)
    torch.cuda.synchronize()
    st = stats_t.cpu().numpy().view(np.uint32).reshape(nq, 8).copy().view(capi.STATS_DTYPE).reshape(nq)
    bytes_per_launch = int(harness.algorithmic_bytes(st, d, d if args.dist == "scalar" else d & ~15, k, row_elem_bytes).sum())
    bytes_per_launch += 3 * ef * (d & ~15) * nq if args.rows == "q8" else 0      # the re-scored rows are float rows
    ids_final = out_ids.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    # top-(k+1) for the reference's own "Precision" figure
    prec_ref = None
    if parts == 1:
        kk_ids = torch.empty((nq, k + 1), dtype=torch.int32, device=dev)
        kk_d = torch.empty((nq, k + 1), dtype=torch.float32, device=dev)
        h.search_batch_device(q_t.data_ptr(), nq, max(ef, k + 1), k + 1, kk_ids.data_ptr(), kk_d.data_ptr(), 0, 0,
                              torch.cuda.current_stream().cuda_stream, flags)
        torch.cuda.synchronize()
        prec_ref = harness.reference_precision(kk_ids.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt, k)
    recall = harness.recall_at_k(ids_final, gt, k)
    if world > 1 and parts == 1:
        # replicated: every rank picked its operating point on its own queries; they must agree, and the
        # reported recall is the mean over ranks
        rt = torch.tensor([recall, float(ef), -float(ef)], dtype=torch.float64, device=cdev)
        mx = rt.clone()
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        if mx[1].item() != -mx[2].item():
            raise SystemExit("replicated ranks chose different ef values: pass --ef")
        recall = float(rt[0].item()) / world

    # ---- timed region ----
    log(f"rank {rank}: operating point ef={ef} recall@10={recall:.4f}; timing {args.steps} steps")
    for _ in range(args.warmup):
        step(ef, k)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step(ef, k)
        ev[i][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms_dev = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    qps = nq * (world if parts == 1 else 1) * args.steps / elapsed
    log(f"rank {rank}: {ms_per_step:.3f} ms/step")

    # search-kernel time alone (events around the search call only; N=1: the whole step)
    if parts == 1:
        kern_ms = step_ms_dev
    else:
        kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in kev:
            a.record()
            h.search_batch_device(q_t.data_ptr(), nq, ef, k, ids_t.data_ptr(), dist_t.data_ptr(), cnt_t.data_ptr(), 0,
                                  torch.cuda.current_stream().cuda_stream, flags)
            b.record()
        torch.cuda.synchronize()
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in kev]))
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    # second denominator, measured now: what a plain device-to-device copy of 1 GiB moves (read + write)
    copy_gbs = None
    if rank == 0:
        ca = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        cb = torch.empty_like(ca)
        cb.copy_(ca)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            cb.copy_(ca)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * ca.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del ca, cb
    # How the timed launches were configured, and what this GPU delivers for the same row reads with all search logic
    # removed (random rows of this index's own table, same loads and arithmetic, same residency): the ceiling the
    # achieved rate can be compared with when the peak of the memory is not reachable by a gather.
    launch_info, ceiling_gbs = None, None
    if rank == 0:
        h.search_batch_device(q_t.data_ptr(), nq, ef, k, ids_t.data_ptr(), dist_t.data_ptr(), cnt_t.data_ptr(), 0,
                              torch.cuda.current_stream().cuda_stream, flags)
        torch.cuda.synchronize()
        launch_info = h.last_launch_info()
        if args.rows == "f32":
            rows_per_wave = max(32, int(round(float(st["n_dist"].mean()) / 32.0)) * 32)
            c_ms, c_bytes = h.gather_ceiling(nq, rows_per_wave, launch_info["lds_bytes_per_query"], torch.cuda.current_stream().cuda_stream,
                                             scalar=(args.dist == "scalar"))
            ceiling_gbs = c_bytes / (c_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes of tools/pmc_passes.sh over this same workload
    # (FETCH_SIZE doubled as the gfx950 counter requires, plus WRITE_SIZE); a profile of
    # another workload does not apply
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_search_kernel_latest.json")
    if world == 1 and os.path.exists(tpath) and args.n == 1_000_000 and nq == 10_000 and args.kind == "sift" and ef == 50 and args.rows == "f32":
        with open(tpath) as f:
            pm = json.load(f)
        traffic = pm.get("hbm_read_bytes_x2", 0) + pm.get("hbm_write_bytes", 0) or None
        traffic_source = {"file": "profiles/pmc_search_kernel_latest.json",
                          "collected": "an earlier run of tools/pmc_passes.sh on this workload (rocprofv3 --pmc passes), not this run",
                          "counters": "2 x FETCH_SIZE + WRITE_SIZE of one launch; they count the L2's fabric-side requests, "
                                      "Infinity-Cache hits included, so this is not DRAM traffic alone"}
    table_bytes = (hi - lo) * ((d + 31) // 32 * 32) * 4 + (hi - lo) * 32 * 4        # stored rows + layer-0 lists on this GPU

    # ---- N=1: the same search on a batch four times as large (not part of `value`): the 10 000-query batch of the
    #      configuration is ~2.4 rounds of resident queries, so ramp-up and the last partial round weigh on it ----
    large = None
    if world == 1 and not args.sweep:
        nql = 4 * nq
        ql_t = torch.from_numpy(datagen.make(args.kind, nql, seed=1, stream=1)).to(dev)
        li = torch.empty((nql, k), dtype=torch.int32, device=dev)
        ld = torch.empty((nql, k), dtype=torch.float32, device=dev)
        lc = torch.empty((nql,), dtype=torch.int32, device=dev)
        ls = torch.empty((nql, 8), dtype=torch.int32, device=dev)
        st_ = torch.cuda.current_stream().cuda_stream
        h.search_batch_device(ql_t.data_ptr(), nql, ef, k, li.data_ptr(), ld.data_ptr(), lc.data_ptr(), ls.data_ptr(), st_, flags)
        torch.cuda.synchronize()
        stl = ls.cpu().numpy().view(np.uint32).reshape(nql, 8).copy().view(capi.STATS_DTYPE).reshape(nql)
        bl = int(harness.algorithmic_bytes(stl, d, d if args.dist == "scalar" else d & ~15, k, row_elem_bytes).sum())
        l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        l0.record()
        for _ in range(5):
            h.search_batch_device(ql_t.data_ptr(), nql, ef, k, li.data_ptr(), ld.data_ptr(), lc.data_ptr(), 0, st_, flags)
        l1.record()
        torch.cuda.synchronize()
        lms = l0.elapsed_time(l1) / 5
        large = {"queries": nql, "value": round(nql / (lms * 1e-3), 1), "unit": "queries/s", "ms_per_step": round(lms, 4),
                 "achieved_gbs": round(bl / (lms * 1e-3) / 1e9, 1), "frac": round(bl / (lms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del ql_t, li, ld, lc, ls

    # ---- N=1: batches in flight (not part of `value`): the search call in two halves lets the next batch start while
    #      the last queries of the previous one are still running, which is what the lone 10 000-query batch lacks ----
    pipelined = None
    if parts > 1 and backend == "nccl" or (parts > 1 and os.environ.get("HNSW_AMD_BENCH_PIPELINE_GLOO")):
        # every rank's search of batch i+1 runs while batch i is all-gathered and merged (two slots, two streams)
        def run_sharded(nb):
            pending = None
            last = None
            for i in range(nb):
                hnd = sh.search_begin(q_t, ef, k, slot=i % 2)
                if pending is not None:
                    last = sh.search_finish(pending)
                pending = hnd
            last = sh.search_finish(pending)
            torch.cuda.synchronize()
            return last

        run_sharded(4)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = max(args.steps, 4)
        li_p, ld_p, _ = run_sharded(nb)
        dist.barrier()
        torch.cuda.synchronize()
        tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        dtp = float(tp.item())
        pipelined = {"2_in_flight": {"value": round(nq * nb / dtp, 1), "unit": "queries/s", "batches": nb, "queries_per_batch": nq,
                                     "ms_per_batch": round(dtp / nb * 1e3, 4),
                                     "results_equal_the_blocking_call": bool(torch.equal(li_p, out_ids))},
                     "what": "every rank searches batch i+1 while batch i is all-gathered and merged (two slots on two streams); "
                             "slowest rank's host clock over all batches, merged outputs complete"}
    if world == 1 and not args.sweep:
        pipelined = {}
        for depth in (2, 3):
            streams = [torch.cuda.Stream() for _ in range(depth)]
            bufs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                     torch.empty((nq,), dtype=torch.int32, device=dev)) for _ in range(depth)]
            torch.cuda.synchronize()

            def run(nb):
                calls = []
                for i in range(nb):
                    if len(calls) == depth:
                        h.search_batch_device_finish(calls.pop(0))
                    b = bufs[i % depth]
                    calls.append(h.search_batch_device_begin(q_t.data_ptr(), nq, ef, k, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), 0,
                                                             streams[i % depth].cuda_stream, flags))
                for c in calls:
                    h.search_batch_device_finish(c)
                torch.cuda.synchronize()

            run(2 * depth)
            nb = max(args.steps, 2 * depth)
            t0 = time.perf_counter()
            run(nb)
            dtp = time.perf_counter() - t0
            same = all(bool(torch.equal(b[0], out_ids)) for b in bufs[:min(depth, nb)]) if parts == 1 else None
            pipelined[f"{depth}_in_flight"] = {"value": round(nq * nb / dtp, 1), "unit": "queries/s", "batches": nb, "queries_per_batch": nq,
                                               "ms_per_batch": round(dtp / nb * 1e3, 4),
                                               "achieved_gbs": round(bytes_per_launch * nb / dtp / 1e9, 1),
                                               "frac": round(bytes_per_launch * nb / dtp / 1e9 / HBM_PEAK_GBS, 4),
                                               "results_equal_the_blocking_call": same}
        pipelined["what"] = ("the same batch of queries as a stream of calls in two halves (hnsw_amd_search_batch_device_begin / _finish), "
                             "that many batches in flight on as many streams; host-clock time over all batches, outputs complete")

    # ---- CPU baseline on the host cores (rank 0, N=1 only): bounded sample, the reference's protocol ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline != "none":
        ns = min(args.cpu_sample, nq)
        from oracle import oracle
        from tests.helpers import assert_same_results
        mode = args.cpu_baseline
        build_kind = "scalar" if args.dist == "scalar" else "sse"
        if mode == "auto":
            mode = "reference" if oracle.ref_driver_path(build_kind) else "port"
        if mode == "reference" and args.keys == "float":
            mode = "port"      # the reference has no float-key mode
        ref_note = None
        if mode == "reference":
            drv = oracle.ref_driver_path(build_kind)
            try:
                with tempfile.TemporaryDirectory() as tmp:
                    _probe = subprocess.run([drv], capture_output=True, timeout=20)     # does the binary run on this host at all?
            except Exception as e:      # noqa: BLE001
                ref_note = f"reference binary not runnable here ({type(e).__name__})"
                mode = "port"
        if mode == "reference":
            try:
                with tempfile.TemporaryDirectory() as tmp:
                    gpath, qpath, rpath = (os.path.join(tmp, x) for x in ("g.bin", "q.bin", "r.bin"))
                    t0 = time.time()
                    h.saveGraph(gpath)            # the reference's own file format
                    queries[:ns].tofile(qpath)
                    # (the reference's visited table has no overflow check, src/hnsw.h:90-107: it never returns
                    # from a query that visits 16384 nodes — hence the time limits)
                    out = subprocess.run([drv, "ltime", gpath, "16", "16", "200", qpath, str(ns), str(ef), "30"],
                                         capture_output=True, text=True, check=True, timeout=150).stderr
                    rec = json.loads(out.strip().splitlines()[-1])
                    # full-size parity against the reference itself: its complete W_ (all ef entries) for every timed query
                    npar = ns
                    subprocess.run([drv, "lquery", gpath, "16", "16", "200", qpath, str(npar), rpath, str(ef)], check=True,
                                   capture_output=True, timeout=300)
                    r_ids, r_d, r_cnt = oracle.parse_ref_results(rpath)[ef]
                    # the reference driver's own "Precision" (src/main.cpp:236-280) from the reference's own W_, all queries
                    apath, gtpath = os.path.join(tmp, "qa.bin"), os.path.join(tmp, "gt.bin")
                    queries.tofile(apath)
                    np.ascontiguousarray(gt[:, :k], dtype=np.uint32).tofile(gtpath)
                    pout = subprocess.run([drv, "lprecision", gpath, "16", "16", "200", apath, str(nq), gtpath, str(k), str(max(ef, k + 1))],
                                          capture_output=True, text=True, check=True, timeout=300).stderr
                    ref_prec = json.loads(pout.strip().splitlines()[-1])["ref_precision"]
                    log(f"reference binary: save + load + timing + {npar}-query dump + precision over {nq} queries in {time.time() - t0:.1f}s")
                h.search_batch_device(q_t.data_ptr(), npar, ef, ef, ids_t.data_ptr(), dist_t.data_ptr(), cnt_t.data_ptr(), 0,
                                      torch.cuda.current_stream().cuda_stream, flags)
                torch.cuda.synchronize()
                g_ids = ids_t.view(-1)[:npar * ef].view(npar, ef).cpu().numpy().view(np.uint32)
                g_d = dist_t.view(-1)[:npar * ef].view(npar, ef).cpu().numpy()
                g_cnt = cnt_t[:npar].cpu().numpy().view(np.uint32)
                try:
                    assert np.array_equal(g_cnt, r_cnt), "|W_| differs"
                    assert_same_results(g_ids, g_d, r_ids, r_d, what="GPU vs reference binary")
                    same = True
                except AssertionError as e:
                    log(f"PARITY FAILURE against the reference binary: {e}")
                    same = False
                prec_ok = prec_ref is not None and abs(prec_ref - ref_prec) <= 0.001
                if not prec_ok:
                    log(f"PRECISION MISMATCH: reference binary {ref_prec} vs this library {prec_ref}")
                cpu = {"value": round(rec["qps"], 1), "unit": "queries/s", "cores": 1, "kind": "reference",
                       "sample": f"first {ns} queries, ef={ef}, min of {rec['passes']} passes, 1 thread; unmodified reference binary, "
                                 f"index read by its own loadGraph from the file this library saved",
                       "gpu_W_equals_reference_W_on_first": npar, "W_entries_compared": int(npar * ef), "parity": same,
                       "reference_precision": round(ref_prec, 5), "reference_precision_queries": nq,
                       "precision_delta": round(abs(prec_ref - ref_prec), 6) if prec_ref is not None else None,
                       "precision_within_0.001": bool(prec_ok),
                       "host_cpu_model": harness.cpu_model(), "host_cores": os.cpu_count()}
                # the reference object cannot be shared between threads (a query writes to it, src/hnsw.cpp:380), so the
                # many-core figure is the CPU restatement's: one private scratch per thread over the shared flat graph
                nt = harness.usable_cores()
                idx = oracle.OracleIndex(h.export_flat(), base)
                t0 = time.perf_counter()
                idx.search(queries, ef, out_k=k, threads=nt, dist_mode=1 if args.dist == "scalar" else 0, key_mode=0)
                cpu["all_cores"] = {"value": round(nq / (time.perf_counter() - t0), 1), "unit": "queries/s", "cores": nt, "kind": "port",
                                    "sample": f"all {nq} queries once, {nt} threads (CPU restatement, bit-equal to the reference)"}
            except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
                ref_note = (f"reference binary did not finish ({type(e).__name__}); its visited table never returns once a "
                            f"query visits 16384 nodes (src/hnsw.h:90-107)")
                log(ref_note)
                mode = "port"
        if mode == "port":
            idx = oracle.OracleIndex(h.export_flat(), base)
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                o_ids, o_d, _ = idx.search(queries[:ns], ef, out_k=k, threads=1, dist_mode=1 if args.dist == "scalar" else 0,
                                            key_mode=1 if args.keys == "float" else 0)
                best = min(best, time.perf_counter() - t0)
            nt = harness.usable_cores()
            t0 = time.perf_counter()
            idx.search(queries, ef, out_k=k, threads=nt, dist_mode=1 if args.dist == "scalar" else 0, key_mode=1 if args.keys == "float" else 0)
            mt = time.perf_counter() - t0
            same = bool(np.array_equal(o_d, out_dist.cpu().numpy()[:ns]))
            cpu = {"value": round(ns / best, 1), "unit": "queries/s", "cores": 1, "kind": "port",
                   "sample": f"first {ns} queries, ef={ef}, min of 3 passes, 1 thread (CPU restatement, bit-equal to the reference)",
                   "mt": {"value": round(nq / mt, 1), "cores": nt},
                   "gpu_distances_bit_equal_on_sample": same,
                   "host_cpu_model": harness.cpu_model(), "host_cores": os.cpu_count()}
            if ref_note:
                cpu["note"] = ref_note

    # ---- N > 1, sharded: the replicated layout measured next to it (whole index on every GPU, a batch of
    #      queries per rank, no collective in the data path; work grows with N) ----
    alongside = None
    if world > 1 and parts > 1 and not args.no_alongside and args.n <= 2_000_000:
        from hnsw_amd.flatgraph import FlatGraph
        t0 = time.time()
        fpath = os.path.join(tempfile.gettempdir(), f"hnsw_amd_bench_full_{os.environ.get('MASTER_PORT', '0')}_{args.kind}_{args.n}.npz")
        full = datagen.make(args.kind, args.n, seed=1, stream=0)
        hf = HNSW(16, 16, 200, device=local_rank)
        if rank == 0 and os.path.exists(fpath):
            os.remove(fpath)
        dist.barrier()                                  # (so that no rank still reads a file of an earlier run)
        if rank == 0:                                   # one build for the node, handed over through a local file
            hf.init(d, args.n)
            nt_full = harness.usable_cores()            # the other ranks wait: the whole share of the host
            for r0 in range(0, args.n, 500_000):
                hf.insert_batch(full[r0:r0 + 500_000], threads=nt_full)
                log(f"alongside: inserted {min(args.n, r0 + 500_000)} / {args.n} in {time.time() - t0:.0f}s")
            np.savez(fpath + ".tmp.npz", **hf.export_flat().to_npz_dict())
            os.replace(fpath + ".tmp.npz", fpath)
        else:
            # wait on the host for the file, not inside a collective on the GPU, however long the build takes
            limit = time.time() + 2400
            while not os.path.exists(fpath):
                if time.time() > limit:
                    raise SystemExit("alongside: rank 0 did not deliver the index file")
                time.sleep(0.5)
            hf.import_flat(FlatGraph.from_npz(np.load(fpath)), full)
        dist.barrier()
        if rank == 0:
            os.remove(fpath)
        hf.flags = flags
        hf.sync_device(local_rank)
        q_r = datagen.make(args.kind, nq, seed=1, stream=1, row0=rank * nq)
        qr_t = torch.from_numpy(q_r).to(dev)
        full_t = torch.from_numpy(full).to(dev)
        _, gi_r, _ = harness.exact_topk_mfma(full_t, qr_t, k)
        del full_t
        gt_r = gi_r.cpu().numpy()
        stream = torch.cuda.current_stream().cuda_stream

        def rstep(ef_):
            hf.search_batch_device(qr_t.data_ptr(), nq, ef_, k, out_ids.data_ptr(), out_dist.data_ptr(), cnt_t.data_ptr(), 0, stream, flags)

        ef_r, rec_r = harness.REFERENCE_EF_SWEEP[-1], 0.0
        for e in ([args.ef] if args.ef else harness.REFERENCE_EF_SWEEP):
            rstep(e)
            torch.cuda.synchronize()
            rec_r = harness.recall_at_k(out_ids.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt_r, k)
            ef_r = e
            if args.ef or rec_r >= args.target_recall:
                break
        et = torch.tensor([float(ef_r)], dtype=torch.float64, device=cdev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)       # every rank at the largest ef any rank needed
        ef_r = int(et.item())
        rstep(ef_r)
        torch.cuda.synchronize()
        rec_r = harness.recall_at_k(out_ids.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt_r, k)
        for _ in range(args.warmup):
            rstep(ef_r)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rstep(ef_r)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        tt = torch.tensor([el, rec_r], dtype=torch.float64, device=cdev)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        el = float(mx[0].item())
        alongside = {"layout": f"whole index on each of {world} GPUs, {nq} queries per GPU, no collective",
                     "value": round(world * nq * args.steps / el, 1), "unit": "queries/s", "scaling": "weak",
                     "ms_per_step": round(el / args.steps * 1e3, 4), "ef": ef_r, "recall_at_10": round(float(tt[1].item()) / world, 4)}
        log(f"alongside (replicated): {alongside} in {time.time() - t0:.0f}s")

    if rank == 0:
        out = {
            "metric": ("queries/s at recall@10>=0.95, HNSW search, SIFT-128-shaped 1M base / 10k queries"
                       if (args.kind, args.n, nq) == ("sift", 1_000_000, 10_000) else
                       f"queries/s at the recall reported, HNSW search, {args.kind}-{d}-shaped {args.n} base / {nq} queries"),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if parts > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.kind}-{d} synthetic, n={args.n}, nq={nq}, M=16, efConstruction=200, ef={ef}, k={k}, "
                                   f"keys={args.keys}, rows={args.rows}" + (f", {world} shards + RCCL all-gather + device merge" if parts > 1 else "")
                                   + (f", index replicated on {world} GPUs, {nq} queries each" if world > 1 and parts == 1 else ""),
                       "ef": ef, "recall_at_10": round(recall, 4), "reference_precision_top11": prec_ref,
                       "timed_step_output": f"the first k={k} entries of every W_ (the reference leaves all ef entries sorted in W_; "
                                            f"the parity check of the cpu_baseline leg compares all ef)",
                       "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6, 2),
                       "ef_sweep": sweep, "build_s": round(build_s, 1), "build_threads": args.build_threads, "build_keys": args.build_keys, "build": args.build,
                       "ground_truth": dict(gt_info, method="exact brute force, f32-input MFMA kernel of this library")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "measured_copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
                         "frac_of_measured_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
                         "traffic_source": traffic_source,
                         "gather_ceiling_gbs": round(ceiling_gbs, 1) if ceiling_gbs else None,
                         "frac_of_gather_ceiling": round(achieved / ceiling_gbs, 4) if ceiling_gbs else None,
                         "gather_ceiling_is": "random rows of this index's table read with the search kernel's loads and distance "
                                              "arithmetic, same residency, no search logic; measured in this run",
                         "working_set_bytes": table_bytes,
                         "infinity_cache_bytes": 256 << 20,
                         "infinity_cache_share_of_working_set": round(min(1.0, (256 << 20) / table_bytes), 3),
                         "note": "peak is the HBM specification; reads served by the 256 MiB Infinity Cache count as achieved bytes too, "
                                 "so with a working set of a few hundred MB part of the achieved rate is on-die, not DRAM",
                         "launch": launch_info,
                         "kernel": "hnsw_search_kernel", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_query": round(bytes_per_launch / nq, 1)},
            "cpu_baseline": cpu,
        }
        if alongside is not None:
            out["replicated_alongside"] = alongside
        if large is not None:
            out["large_batch"] = large
        if pipelined is not None:
            out["pipelined"] = pipelined
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    # a run whose results are not the reference's is a failed run, whatever it measured: the line above is printed
    # for the record, the exit status says so
    if rank == 0 and cpu is not None and (cpu.get("parity") is False or cpu.get("precision_within_0.001") is False
                                          or cpu.get("gpu_distances_bit_equal_on_sample") is False):
        log("FAILED: results differ from the reference's (see cpu_baseline in the line above)")
        sys.exit(3)


if __name__ == "__main__":
    main()
