"""GPU: the sharded layout on one device — every shard's sub-graph searched by the kernel, the
per-shard lists merged by the device kernel — against per-shard CPU restatement + host merge."""
import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen
from hnsw_amd.sharded import merge_topk_numpy, shard_offsets, shard_range
from oracle import oracle
from tests.helpers import assert_same_results, recall_at_k

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S", [2, 5])
def test_sharded_search_and_device_merge(S):
    import torch
    n, d, nq, ef, k = 30000, 128, 200, 32, 10
    base = datagen.make("sift", n, seed=6)
    q = datagen.make("sift", nq, seed=6, stream=1)
    offs = shard_offsets(n, S)
    g_d = torch.empty((S, nq, k), dtype=torch.float32, device="cuda")
    g_i = torch.empty((S, nq, k), dtype=torch.int32, device="cuda")
    q_t = torch.from_numpy(q).cuda()
    exp_i, exp_d = [], []
    keep = []
    for s in range(S):
        lo, hi = shard_range(n, s, S)
        h = HNSW(16, 16, 200)
        h.init(d, hi - lo)
        h.insert_batch(base[lo:hi], threads=8)
        keep.append(h)
        h.sync_device(0)
        h.search_batch_device(q_t.data_ptr(), nq, ef, k, g_i[s].data_ptr(), g_d[s].data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
        oi, od, _ = oracle.OracleIndex(h.export_flat(), base[lo:hi]).search(q, ef, out_k=k)
        exp_i.append(oi); exp_d.append(od)
    o_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    o_d = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    capi.check(capi.lib().hnsw_amd_merge_topk_device(g_d.data_ptr(), g_i.data_ptr(), capi.u32(offs), S, nq, k, k, 0, o_i.data_ptr(),
                                                     o_d.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    eg, ed = merge_topk_numpy(np.stack(exp_d), np.stack(exp_i), offs, k)
    assert_same_results(o_i.cpu().numpy().view(np.uint32), o_d.cpu().numpy(), eg, ed, what=f"{S} shards")
    gt = oracle.bruteforce(base, q, k, threads=8)
    assert recall_at_k(eg, gt, k) > 0.9


def test_deep_shaped_eight_shards_through_ShardedHNSW_pieces():
    """The DEEP-96 configuration's layout on one device: unit-norm 96-dim rows in 8 sub-graphs (built with the float
    construction keys that configuration uses), searched with the reference's int32 keys, the 8 lists merged by the
    device kernel from one gathered buffer laid out as the all-gather leaves it ([shard][dist|ids][nq][k]) — against
    the per-shard CPU restatement + host merge."""
    import torch
    from hnsw_amd.sharded import ShardedHNSW
    S, n, d, nq, ef, k = 8, 320_000, 96, 300, 30, 10
    base = datagen.make("deep", n, seed=8)
    q = datagen.make("deep", nq, seed=8, stream=1)
    q_t = torch.from_numpy(q).cuda()
    gathered = torch.empty((S, 2, nq, k), dtype=torch.int32, device="cuda")
    exp_i, exp_d, keep = [], [], []
    for s in range(S):
        sh = ShardedHNSW(n, d, s, S, device=0)
        sh.index.init(d, sh.hi - sh.lo)
        sh.index.set_build_float_keys(True)
        sh.index.insert_batch(base[sh.lo:sh.hi], threads=16)
        keep.append(sh)
        # the rank-local half of ShardedHNSW.search: the shard's own lists (world of one = no exchange)
        one = ShardedHNSW.__new__(ShardedHNSW)
        one.__dict__.update(sh.__dict__); one.world = 1
        li, ld = one.search(q_t, ef, k)
        gathered[s, 0].copy_(ld.view(torch.int32)); gathered[s, 1].copy_(li)
        oi, od, _ = oracle.OracleIndex(sh.index.export_flat(), base[sh.lo:sh.hi]).search(q, ef, out_k=k, threads=8)
        exp_i.append(oi); exp_d.append(od)
    offs = keep[0].offsets
    assert offs.tolist() == [s * n // S for s in range(S)]
    o_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    o_d = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    capi.check(capi.lib().hnsw_amd_merge_topk_device(gathered.data_ptr(), gathered.data_ptr() + 4 * nq * k, capi.u32(offs), S, nq, k, k,
                                                     2 * nq * k, o_i.data_ptr(), o_d.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    eg, ed = merge_topk_numpy(np.stack(exp_d), np.stack(exp_i), offs, k)
    assert_same_results(o_i.cpu().numpy().view(np.uint32), o_d.cpu().numpy(), eg, ed, what="deep, 8 shards")
    gt = oracle.bruteforce(base, q, k, threads=8)
    assert recall_at_k(eg, gt, k) > 0.95


def test_exchange_over_rccl_with_a_world_of_one():
    """The exchange step as the multi-GPU runs take it — torch.distributed backend "nccl" (= RCCL), all_gather_into_tensor of
    the packed [dist | ids] buffer on the device, then the device merge reading the gathered buffer with its stride — on the one
    GPU this box has: a process group of a single rank.  Against the oracle on the same shard."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from hnsw_amd.sharded import ShardedHNSW
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, d, nq, ef, k = 20000, 96, 257, 30, 10
        base = datagen.make("deep", n, seed=12)
        q = datagen.make("deep", nq, seed=12, stream=1)
        sh = ShardedHNSW(n, d, 0, 1, device=0, group=dist.group.WORLD)
        sh.build(base, threads=16)
        ids, ds = sh.search(torch.from_numpy(q).cuda(), ef, k)
        torch.cuda.synchronize()
        oi, od, _ = oracle.OracleIndex(sh.index.export_flat(), base).search(q, ef, out_k=k)
        assert_same_results(ids.cpu().numpy().view(np.uint32), ds.cpu().numpy(), oi, od, what="one rank over RCCL")
        # the same in two halves, two batches in flight: the second batch searches while the first is gathered and merged
        q2 = datagen.make("deep", nq, seed=13, stream=1)
        qa, qb = torch.from_numpy(q).cuda(), torch.from_numpy(q2).cuda()
        for _ in range(2):
            ha = sh.search_begin(qa, ef, k, slot=0)
            hb = sh.search_begin(qb, ef, k, slot=1)
            ia, da, ea = sh.search_finish(ha)
            ib, db, eb = sh.search_finish(hb)
            ea.synchronize(); eb.synchronize()
            oi2, od2, _ = oracle.OracleIndex(sh.index.export_flat(), base).search(q2, ef, out_k=k)
            assert_same_results(ia.cpu().numpy().view(np.uint32), da.cpu().numpy(), oi, od, what="pipelined, first batch")
            assert_same_results(ib.cpu().numpy().view(np.uint32), db.cpu().numpy(), oi2, od2, what="pipelined, second batch")
    finally:
        dist.destroy_process_group()
