"""Sharded multi-GPU search: one process per GPU, each holding an independently built
sub-graph over a contiguous slice of the base set; queries are replicated, every rank
searches its shard, the per-shard top-k lists are all-gathered (RCCL over xGMI when the
process group's backend is "nccl") and merged on the device.

The reference is a single in-memory index (no counterpart); this is the layout the
multi-GPU configurations of the benchmark use.
"""
from __future__ import annotations

import numpy as np

from . import capi
from .index import HNSW


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Rows [lo, hi) of the base set owned by `rank`: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * n_total // world, (rank + 1) * n_total // world


def shard_offsets(n_total: int, world: int) -> np.ndarray:
    return np.array([shard_range(n_total, r, world)[0] for r in range(world)], dtype=np.uint32)


def merge_topk_numpy(dist: np.ndarray, ids: np.ndarray, offsets: np.ndarray, out_k: int):
    """Host statement of the merge the device kernel performs: dist/ids are [S, nq, k] with
    shard-local ids (0xFFFFFFFF = padding); returns the out_k nearest per query as global
    ids, ascending distance, ties by id."""
    S, nq, k = dist.shape
    gid = ids.astype(np.int64) + offsets.astype(np.int64)[:, None, None]
    pad = ids == np.uint32(0xFFFFFFFF)
    gid[pad] = 0xFFFFFFFF
    d = np.where(pad, np.inf, dist).transpose(1, 0, 2).reshape(nq, S * k)
    g = gid.transpose(1, 0, 2).reshape(nq, S * k)
    order = np.lexsort((g, d), axis=1)[:, :out_k]
    od = np.take_along_axis(d, order, 1).astype(np.float32)
    og = np.take_along_axis(g, order, 1).astype(np.uint32)
    if od.shape[1] < out_k:
        padn = out_k - od.shape[1]
        od = np.concatenate([od, np.full((nq, padn), np.inf, np.float32)], 1)
        og = np.concatenate([og, np.full((nq, padn), 0xFFFFFFFF, np.uint32)], 1)
    return og, od


class ShardedHNSW:
    """The rank-local part of a sharded index plus the exchange step.

    search(q, ef, k) is collective: every rank of `group` must call it with the same
    queries.  `local_search` and `merge` can be replaced (CPU tests run the exchange
    over gloo with host tensors and a host merge); by default they are the GPU kernels.
    """

    def __init__(self, n_total: int, d: int, rank: int, world: int, ctor=(16, 16, 200), device: int = 0, group=None):
        self.n_total, self.d, self.rank, self.world = n_total, d, rank, world
        self.lo, self.hi = shard_range(n_total, rank, world)
        self.offsets = shard_offsets(n_total, world)
        self.group = group
        self.index = HNSW(*ctor, device=device)

    def build(self, shard_rows: np.ndarray, threads: int = 1) -> None:
        if shard_rows.shape != (self.hi - self.lo, self.d):
            raise ValueError(f"rank {self.rank} owns rows [{self.lo}, {self.hi}): got {shard_rows.shape}")
        self.index.init(self.d, self.hi - self.lo)
        self.index.insert_batch(shard_rows, threads=threads)

    # -- exchange buffers, allocated once per (nq, k) and reused by every batch ---------------------------
    def _slot(self, which: int, nq: int, k: int, device):
        """Buffers (and, on a GPU, the stream) of pipeline slot `which`: the local result [2, nq, k] (distances, then
        ids, as 32-bit words, so that the exchange is ONE all-gather), the gathered results [world, 2, nq, k], and the
        merged output.  Allocated on first use of a shape, then reused by every batch."""
        import torch
        slots = self.__dict__.setdefault("_slots", {})
        key = (which, nq, k, str(device))
        sl = slots.get(key)
        if sl is None:
            host_side = device.type == "cuda" and self._host_exchange()
            sl = {"loc": torch.empty((2, nq, k), dtype=torch.int32, device=device),
                  "gathered": torch.empty((self.world, 2, nq, k), dtype=torch.int32, device=device),
                  "out_i": torch.empty((nq, k), dtype=torch.int32, device=device),
                  "out_d": torch.empty((nq, k), dtype=torch.float32, device=device),
                  "stream": torch.cuda.Stream(device=device) if device.type == "cuda" else None,
                  "loc_h": torch.empty((2, nq, k), dtype=torch.int32).pin_memory() if host_side else None,
                  "gathered_h": torch.empty((self.world, 2, nq, k), dtype=torch.int32).pin_memory() if host_side else None}
            slots[key] = sl
        return sl

    def _host_exchange(self) -> bool:
        """A host-side backend (gloo: rehearsals of the multi-GPU path on one GPU): the exchange travels as host copies."""
        import torch.distributed as dist
        return (self.world > 1 or self.group is not None) and dist.get_backend(self.group) != "nccl"

    def search_begin(self, q_t, ef: int, k: int, slot: int = 0):
        """First half of a pipelined search on the GPU: launches this rank's search of the batch on the slot's own
        stream and returns at once.  search_finish(handle) completes it.  Batches in different slots overlap: while
        one is exchanged and merged, the next one is already searching."""
        import torch
        nq = q_t.shape[0]
        sl = self._slot(slot, nq, k, q_t.device)
        sl["stream"].wait_stream(torch.cuda.current_stream(q_t.device))          # (the queries were produced there)
        loc_d, loc_i = sl["loc"][0].view(torch.float32), sl["loc"][1]
        call = self.index.search_batch_device_begin(q_t.data_ptr(), nq, ef, k, loc_i.data_ptr(), loc_d.data_ptr(), 0, 0,
                                                    sl["stream"].cuda_stream)
        return (sl, call, nq, k)

    def search_finish(self, handle):
        """Second half: waits for this rank's search (re-running what outgrew its scratch), then queues the all-gather
        and the merge on the slot's stream.  Returns (ids, dist, event): the tensors are the slot's own output buffers,
        complete once `event` has passed (wait on it, or synchronise, before reading; they are overwritten by the next
        batch in the same slot)."""
        import torch
        import torch.distributed as dist
        sl, call, nq, k = handle
        self.index.search_batch_device_finish(call)
        stream = sl["stream"]
        with torch.cuda.stream(stream):
            if self.world == 1 and self.group is None:
                out_i, out_d = sl["loc"][1], sl["loc"][0].view(torch.float32)
            else:
                if sl["loc_h"] is not None:
                    sl["loc_h"].copy_(sl["loc"], non_blocking=True)
                    stream.synchronize()
                    dist.all_gather_into_tensor(sl["gathered_h"].view(-1), sl["loc_h"].view(-1), group=self.group)
                    sl["gathered"].copy_(sl["gathered_h"], non_blocking=True)
                else:
                    dist.all_gather_into_tensor(sl["gathered"].view(-1), sl["loc"].view(-1), group=self.group)
                g = sl["gathered"]
                capi.check(capi.lib().hnsw_amd_merge_topk_device(g.data_ptr(), g.data_ptr() + 4 * nq * k, capi.u32(self.offsets),
                                                                 self.world, nq, k, k, 2 * nq * k, sl["out_i"].data_ptr(),
                                                                 sl["out_d"].data_ptr(), stream.cuda_stream))
                out_i, out_d = sl["out_i"], sl["out_d"]
            ev = torch.cuda.Event()
            ev.record(stream)
        return out_i, out_d, ev

    def search(self, q_t, ef: int, k: int, *, local_search=None, merge=None):
        """q_t: torch tensor [nq, d] float32 (on this rank's GPU, or on the CPU when the hooks
        are given).  Returns (ids [nq,k] int32 holding global ids, dist [nq,k] float32).  The tensors are buffers of
        this object, reused by the next call with the same shape."""
        import torch
        import torch.distributed as dist

        if local_search is None and merge is None and q_t.is_cuda:
            out_i, out_d, ev = self.search_finish(self.search_begin(q_t, ef, k))
            torch.cuda.current_stream(q_t.device).wait_event(ev)
            return out_i, out_d
        nq = q_t.shape[0]
        sl = self._slot(0, nq, k, q_t.device)
        loc = sl["loc"]
        loc_d, loc_i = loc[0].view(torch.float32), loc[1]
        if local_search is None:
            self.index.search_batch_device(q_t.data_ptr(), nq, ef, k, loc_i.data_ptr(), loc_d.data_ptr(), 0, 0,
                                           torch.cuda.current_stream().cuda_stream)
        else:
            li, ld = local_search(q_t, ef, k)
            loc_i.copy_(li); loc_d.copy_(ld)
        if self.world == 1 and self.group is None:
            return loc_i, loc_d            # (with a process group the exchange runs even for a world of one: same code path)
        gathered = sl["gathered"]
        if sl["loc_h"] is not None:
            sl["loc_h"].copy_(loc)
            dist.all_gather_into_tensor(sl["gathered_h"].view(-1), sl["loc_h"].view(-1), group=self.group)
            gathered.copy_(sl["gathered_h"])
        else:
            dist.all_gather_into_tensor(gathered.view(-1), loc.view(-1), group=self.group)
        if merge is not None:
            return merge(gathered[:, 0].view(torch.float32).contiguous(), gathered[:, 1].contiguous(), self.offsets, k)
        stride = 2 * nq * k
        capi.check(capi.lib().hnsw_amd_merge_topk_device(gathered.data_ptr(), gathered.data_ptr() + 4 * nq * k, capi.u32(self.offsets),
                                                         self.world, nq, k, k, stride, sl["out_i"].data_ptr(), sl["out_d"].data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream))
        return sl["out_i"], sl["out_d"]
