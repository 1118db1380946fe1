"""Python mirror of the reference's `class HNSW` (src/hnsw.h:217-603 of the reference) over
the C ABI.  Same member names and argument meaning: HNSW(M, Mmax, efConstruction), init,
insert, aproximateKnn (result left in W_), saveGraph, loadGraph; plus the batched
search the GPU is for.  Searches run on the GPU through libhnsw_amd.so, never on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import numpy as np

from . import capi
from .flatgraph import FlatGraph


class Neighbors(NamedTuple):
    """Element of W_ (reference: struct Neighbors, src/node.h:17-40, without the node pointer)."""
    distance: float
    node_order: int


class HNSW:
    def __init__(self, M: int, Mmax: int, efConstruction: int, device: int = 0):
        self._h = C.c_void_p()
        capi.check(capi.lib().hnsw_amd_create(M, Mmax, efConstruction, C.byref(self._h)))
        self._destroy = capi.lib().hnsw_amd_destroy      # (kept here: module globals may be gone when __del__ runs at exit)
        self.device = device
        self.W_: list[Neighbors] = []
        self.flags = capi.DIST_SSE | capi.KEYS_INT32

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        destroy = getattr(self, "_destroy", None)
        if h and destroy is not None:
            destroy(h)

    # -- reference API ----------------------------------------------------------------
    def init(self, vector_size: int, max_node_count: int) -> None:
        capi.check(capi.lib().hnsw_amd_init(self._h, vector_size, max_node_count))

    def insert(self, q: np.ndarray) -> None:
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.shape != (self.vector_size_,):
            raise ValueError("insert: vector has the wrong length")
        capi.check(capi.lib().hnsw_amd_insert(self._h, capi.f32(q)))

    def aproximateKnn(self, q: np.ndarray, k: int, ef: int) -> None:
        """Result in self.W_: up to ef entries, ascending distance (k is ignored, as upstream)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.shape != (self.vector_size_,):
            raise ValueError("aproximateKnn: vector has the wrong length")
        ids = np.empty(max(ef, 1), np.uint32)
        ds = np.empty(max(ef, 1), np.float32)
        cnt = C.c_uint32(0)
        capi.check(capi.lib().hnsw_amd_aproximate_knn(self._h, capi.f32(q), k, ef, self.flags, capi.u32(ids), capi.f32(ds),
                                                      C.byref(cnt)))
        self.W_ = [Neighbors(float(ds[i]), int(ids[i])) for i in range(cnt.value)]

    def saveGraph(self, filename: str) -> None:
        capi.check(capi.lib().hnsw_amd_save_graph(self._h, filename.encode()))

    def loadGraph(self, filename: str) -> None:
        capi.check(capi.lib().hnsw_amd_load_graph(self._h, filename.encode()))

    # -- batched / device entry points ---------------------------------------------------
    def insert_batch(self, rows: np.ndarray, threads: int = 1, gpu=False) -> None:
        """threads > 1: concurrent build.  gpu=True: the construction searches of layer-0-only vectors run on the GPU
        (hnsw_amd_insert_batch_assisted); gpu="link": selection and linking of layer 0 run there as well
        (hnsw_amd_insert_batch_linked)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.vector_size_:
            raise ValueError("insert_batch: rows must be n x vector_size")
        if gpu == "link":
            capi.check(capi.lib().hnsw_amd_insert_batch_linked(self._h, capi.f32(rows), rows.shape[0], threads, self.device))
        elif gpu:
            capi.check(capi.lib().hnsw_amd_insert_batch_assisted(self._h, capi.f32(rows), rows.shape[0], threads, self.device))
        else:
            capi.check(capi.lib().hnsw_amd_insert_batch(self._h, capi.f32(rows), rows.shape[0], threads))

    def set_build_float_keys(self, on: bool = True) -> None:
        """Construction searches keyed by the float distance (not the reference's int32 truncation): a regular HNSW
        graph, several times faster to build on unit-norm data, but no longer the reference's adjacency."""
        capi.check(capi.lib().hnsw_amd_set_build_float_keys(self._h, 1 if on else 0))

    def set_build_distance(self, dist: str) -> None:
        """"sse": insert() computes distances as the shipped reference build does; "scalar": as its
        -DNO_MANUAL_VECTORIZATION build does (all d dims, one running sum)."""
        capi.check(capi.lib().hnsw_amd_set_build_distance(self._h, {"sse": capi.DIST_SSE, "scalar": capi.DIST_SCALAR}[dist]))

    def sync_device(self, device: int | None = None) -> None:
        if device is not None:
            self.device = device
        capi.check(capi.lib().hnsw_amd_sync_device(self._h, self.device))

    def search_batch(self, queries: np.ndarray, ef: int, out_k: int | None = None, *, flags: int | None = None,
                     stats: bool = False):
        """Host arrays in, host arrays out: (ids [nq,out_k] u32, dists f32, counts [nq] u32[, stats])."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.vector_size_:
            raise ValueError("search_batch: queries must be nq x vector_size")
        nq = q.shape[0]
        out_k = ef if out_k is None else out_k
        ids = np.empty((nq, out_k), np.uint32)
        ds = np.empty((nq, out_k), np.float32)
        cnt = np.empty(nq, np.uint32)
        st = np.zeros(nq, capi.STATS_DTYPE) if stats else None
        self.sync_device()
        capi.check(capi.lib().hnsw_amd_search_batch(self._h, capi.f32(q), nq, ef, out_k, self.flags if flags is None else flags,
                                                    capi.u32(ids), capi.f32(ds), capi.u32(cnt),
                                                    st.ctypes.data if stats else None))
        return (ids, ds, cnt, st) if stats else (ids, ds, cnt)

    def search_batch_device(self, d_queries: int, nq: int, ef: int, out_k: int, d_ids: int, d_dist: int, d_count: int = 0,
                            d_stats: int = 0, stream: int = 0, flags: int | None = None) -> int:
        """Raw device pointers (ints) of buffers on this index's GPU; returns the number of re-run queries."""
        retried = C.c_uint32(0)
        capi.check(capi.lib().hnsw_amd_search_batch_device(self._h, d_queries, nq, ef, out_k,
                                                           self.flags if flags is None else flags, d_ids, d_dist,
                                                           d_count or None, d_stats or None, stream or None, C.byref(retried)))
        return retried.value

    def search_batch_device_noalloc(self, d_queries: int, nq: int, ef: int, out_k: int, d_ids: int, d_dist: int, d_count: int = 0,
                                    d_stats: int = 0, stream: int = 0, flags: int | None = None) -> int:
        """search_batch_device through the entry point that allocates nothing (the index must be on its device already)."""
        retried = C.c_uint32(0)
        capi.check(capi.lib().hnsw_amd_search_batch_device_noalloc(self._h, d_queries, nq, ef, out_k,
                                                                   self.flags if flags is None else flags, d_ids, d_dist,
                                                                   d_count or None, d_stats or None, stream or None, C.byref(retried)))
        return retried.value

    def search_batch_device_begin(self, d_queries: int, nq: int, ef: int, out_k: int, d_ids: int, d_dist: int, d_count: int = 0,
                                  d_stats: int = 0, stream: int = 0, flags: int | None = None):
        """Launches the batch and returns a handle at once; search_batch_device_finish(handle) waits for it (and re-runs
        what outgrew its scratch).  Up to four calls may be in flight on one index."""
        call = C.c_void_p()
        capi.check(capi.lib().hnsw_amd_search_batch_device_begin(self._h, d_queries, nq, ef, out_k,
                                                                 self.flags if flags is None else flags, d_ids, d_dist,
                                                                 d_count or None, d_stats or None, stream or None, C.byref(call)))
        return call

    def search_batch_device_finish(self, call) -> int:
        retried = C.c_uint32(0)
        capi.check(capi.lib().hnsw_amd_search_batch_device_finish(self._h, call, C.byref(retried)))
        return retried.value

    def reload_env(self) -> None:
        """The library reads its diagnostic environment switches when the index is copied to the GPU; read them again."""
        capi.check(capi.lib().hnsw_amd_reload_env(self._h))

    def gather_ceiling(self, n_waves: int, rows_per_wave: int, lds_bytes_per_wave: int, stream: int = 0, scalar: bool = False):
        """(ms, bytes) of one launch of the row-gather probe on this index's vector table (see include/hnsw_amd.h)."""
        ms, nbytes = C.c_float(0), C.c_uint64(0)
        capi.check(capi.lib().hnsw_amd_gather_ceiling_device(self._h, n_waves, rows_per_wave, lds_bytes_per_wave,
                                                             capi.DIST_SCALAR if scalar else capi.DIST_SSE, stream or None,
                                                             C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def last_launch_info(self) -> dict:
        """Configuration of the last search call's first launch (see include/hnsw_amd.h)."""
        v = [C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)]
        capi.check(capi.lib().hnsw_amd_last_launch_info(self._h, *[C.byref(x) for x in v]))
        return {"lds_bytes_per_query": v[0].value, "resident_queries_per_cu": v[1].value,
                "visited_form": ("32-bit slots", "16-bit probing regions", "16-bit bucket regions")[min(v[2].value, 2)]}

    # -- flat graph exchange ---------------------------------------------------------------
    def import_flat(self, g: FlatGraph, vectors: np.ndarray) -> None:
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        if v.shape != (g.n, g.d):
            raise ValueError("import_flat: vectors must be n x d")
        keep = [np.ascontiguousarray(getattr(g, a), dtype=np.uint32) for a in
                ("adj0", "cnt0", "up_off", "up_order", "up_down", "up_cnt", "up_adj_slot", "up_adj_order", "ep_slot", "ep_order")]
        keep = [a if a.size else np.zeros(1, np.uint32) for a in keep]
        cg = capi.FlatGraphC(g.n, g.d, g.nlayers, g.deg0, g.degu, *[capi.u32(a) for a in keep])
        capi.check(capi.lib().hnsw_amd_import_flat(self._h, C.byref(cg), capi.f32(v)))

    def export_flat(self) -> FlatGraph:
        cg = capi.FlatGraphC()
        capi.check(capi.lib().hnsw_amd_export_flat(self._h, C.byref(cg), None))
        U = int(cg.up_off[cg.nlayers]) if cg.nlayers else 0

        def arr(p, count):
            return np.ctypeslib.as_array(p, shape=(max(count, 1),))[:count].copy()

        return FlatGraph(n=cg.n, d=cg.d, M=self.M_, Mmax=self.Mmax_, deg0=cg.deg0, degu=cg.degu, nlayers=cg.nlayers,
                         adj0=arr(cg.adj0, cg.n * cg.deg0), cnt0=arr(cg.cnt0, cg.n), up_off=arr(cg.up_off, cg.nlayers + 1),
                         up_order=arr(cg.up_order, U), up_down=arr(cg.up_down, U), up_cnt=arr(cg.up_cnt, U),
                         up_adj_slot=arr(cg.up_adj_slot, U * cg.degu), up_adj_order=arr(cg.up_adj_order, U * cg.degu),
                         ep_slot=arr(cg.ep_slot, cg.nlayers), ep_order=arr(cg.ep_order, cg.nlayers))

    # -- public data members of the reference object -----------------------------------------
    def _info(self):
        v = [C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint32(), C.c_int32(), C.c_int32(), C.c_uint32()]
        capi.check(capi.lib().hnsw_amd_get_info(self._h, *[C.byref(x) for x in v]))
        return [x.value for x in v]

    M_ = property(lambda s: s._info()[0])
    Mmax_ = property(lambda s: s._info()[1])
    Mmax0_ = property(lambda s: s._info()[2])
    efConstruction_ = property(lambda s: s._info()[3])
    vector_size_ = property(lambda s: s._info()[4])
    max_node_count_ = property(lambda s: s._info()[5])
    actual_node_count_ = property(lambda s: s._info()[6])
    nlayers = property(lambda s: s._info()[7])
