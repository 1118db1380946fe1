"""The N > 1 path on CPU: two processes over gloo.  Each rank builds the sub-graph of its
contiguous shard with the host builder; the exchange (all-gather of per-shard top-k, merge
to global ids) runs exactly as on GPUs but with host tensors, the local search and the
merge being the CPU checkers (oracle restatement, numpy merge).  Expected result: the
per-shard oracle searches merged on one process."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N, D, NQ, EF, K = 3000, 32, 40, 48, 10


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hnsw_amd import datagen
    from hnsw_amd.sharded import ShardedHNSW, merge_topk_numpy
    from oracle import oracle

    sh = ShardedHNSW(N, D, rank, world, ctor=(16, 16, 100))
    rows = datagen.make("sift", sh.hi - sh.lo, seed=5, row0=sh.lo, d=D)
    sh.build(rows, threads=1)
    idx = oracle.OracleIndex(sh.index.export_flat(), rows)
    q = datagen.make("sift", NQ, seed=5, stream=1, d=D)

    def local_search(q_t, ef, k):
        ids, ds, _ = idx.search(q_t.numpy(), ef, out_k=k)
        return torch.from_numpy(ids.view(np.int32)), torch.from_numpy(ds)

    def merge(g_d, g_i, offsets, k):
        og, od = merge_topk_numpy(g_d.numpy(), g_i.numpy().view(np.uint32), offsets, k)
        return torch.from_numpy(og.view(np.int32)), torch.from_numpy(od)

    ids, ds = sh.search(torch.from_numpy(q), EF, K, local_search=local_search, merge=merge)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids.numpy().view(np.uint32), dist=ds.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_search_over_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    # every rank ends with the same merged result
    assert np.array_equal(r0["ids"], r1["ids"]) and np.array_equal(r0["dist"], r1["dist"])

    # expected: per-shard search on one process, merged
    sys.path.insert(0, ROOT)
    from hnsw_amd import HNSW, datagen
    from hnsw_amd.sharded import merge_topk_numpy, shard_offsets, shard_range
    from oracle import oracle
    q = datagen.make("sift", NQ, seed=5, stream=1, d=D)
    base = datagen.make("sift", N, seed=5, d=D)
    all_i, all_d = [], []
    for r in range(world):
        lo, hi = shard_range(N, r, world)
        assert np.array_equal(base[lo:hi], datagen.make("sift", hi - lo, seed=5, row0=lo, d=D))   # shards regenerate apart
        h = HNSW(16, 16, 100)
        h.init(D, hi - lo)
        h.insert_batch(base[lo:hi])
        i, d_, _ = oracle.OracleIndex(h.export_flat(), base[lo:hi]).search(q, EF, out_k=K)
        all_i.append(i); all_d.append(d_)
    eg, ed = merge_topk_numpy(np.stack(all_d), np.stack(all_i), shard_offsets(N, world), K)
    assert np.array_equal(r0["ids"], eg) and np.array_equal(r0["dist"], ed)
    # and the merged answer is a good one: recall against exact brute force over the whole set
    gt = oracle.bruteforce(base, q, K)
    from tests.helpers import recall_at_k
    assert recall_at_k(eg, gt, K) > 0.9


def test_shard_ranges_partition_the_base_set():
    from hnsw_amd.sharded import shard_range
    for n, w in [(10, 3), (1_000_000, 8), (7, 8)]:
        cuts = [shard_range(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
