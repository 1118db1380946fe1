"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference search).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under hnsw_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DIST_SSE, DIST_SCALAR, DIST_IP = 0, 1, 2      # DIST_IP: this library's inner-product metric (no reference counterpart)
KEYS_INT32, KEYS_FLOAT = 0, 1

_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)


class _Graph(C.Structure):
    _fields_ = [("n", C.c_uint32), ("d", C.c_uint32), ("nlayers", C.c_uint32),
                ("deg0", C.c_uint32), ("degu", C.c_uint32),
                ("vectors", _f32p), ("adj0", _u32p), ("cnt0", _u32p), ("up_off", _u32p),
                ("up_order", _u32p), ("up_down", _u32p), ("up_cnt", _u32p),
                ("up_adj_slot", _u32p), ("up_adj_order", _u32p), ("ep_slot", _u32p), ("ep_order", _u32p)]


STATS_DTYPE = np.dtype([("n_dist", "<u4"), ("n_expand", "<u4"), ("n_sweeps", "<u4"), ("sum_deg0", "<u4"),
                        ("sum_degu", "<u4"), ("max_c", "<u4"), ("n_visited0", "<u4"), ("n_visited_up", "<u4")])


def build(force: bool = False) -> str:
    """Compile liboracle.so (and oracle/_ref when the reference tree is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hnsw_search_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.oracle_knn_batch.argtypes = [C.POINTER(_Graph), _f32p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                       _u32p, _f32p, _u32p, C.c_void_p, C.c_int]
        L.oracle_knn_batch.restype = None
        L.oracle_bruteforce.argtypes = [_f32p, C.c_uint32, C.c_uint32, _f32p, C.c_uint32, C.c_uint32, _u32p, C.c_int]
        L.oracle_bruteforce.restype = None
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


class OracleIndex:
    """A flat graph plus its vectors, searchable by the CPU restatement."""

    def __init__(self, graph, vectors: np.ndarray):
        self.g = graph
        self.vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        assert self.vectors.shape == (graph.n, graph.d)
        self._keep = [np.ascontiguousarray(getattr(graph, a), dtype=np.uint32) for a in
                      ("adj0", "cnt0", "up_off", "up_order", "up_down", "up_cnt", "up_adj_slot", "up_adj_order",
                       "ep_slot", "ep_order")]
        # ctypes wants non-null pointers even for empty arrays
        self._keep = [a if a.size else np.zeros(1, np.uint32) for a in self._keep]
        self.cg = _Graph(graph.n, graph.d, graph.nlayers, graph.deg0, graph.degu, _p(self.vectors, _f32p),
                         *[_p(a, _u32p) for a in self._keep])

    def search(self, queries: np.ndarray, ef: int, out_k: int | None = None, *, dist_mode: int = DIST_SSE,
               key_mode: int = KEYS_INT32, threads: int = 1, stats: bool = False):
        """Returns (ids [nq,out_k] u32, dists [nq,out_k] f32, counts [nq] u32[, stats])."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        assert q.ndim == 2 and q.shape[1] == self.g.d
        out_k = ef if out_k is None else out_k
        ids = np.empty((nq, out_k), np.uint32)
        ds = np.empty((nq, out_k), np.float32)
        cnt = np.empty(nq, np.uint32)
        st = np.zeros(nq, STATS_DTYPE) if stats else None
        lib().oracle_knn_batch(C.byref(self.cg), _p(q, _f32p), nq, ef, dist_mode, key_mode, out_k,
                               _p(ids, _u32p), _p(ds, _f32p), _p(cnt, _u32p),
                               st.ctypes.data if stats else None, threads)
        return (ids, ds, cnt, st) if stats else (ids, ds, cnt)


def bruteforce(base: np.ndarray, queries: np.ndarray, k: int, threads: int = 8) -> np.ndarray:
    b = np.ascontiguousarray(base, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    out = np.empty((q.shape[0], k), np.uint32)
    lib().oracle_bruteforce(_p(b, _f32p), b.shape[0], b.shape[1], _p(q, _f32p), q.shape[0], k, _p(out, _u32p), threads)
    return out


def parse_ref_results(path: str):
    """Read a results file written by oracle/_ref/ref_driver_* (modes query / lquery):
    {ef: (ids [nq,ef] u32, dists [nq,ef] f32, counts [nq] u32)}, rows padded with 0xFFFFFFFF / +inf."""
    import struct
    with open(path, "rb") as f:
        buf = f.read()
    magic, nq, nef = struct.unpack_from("<III", buf, 0)
    assert magic == 0x31535248, "not a ref_driver results file"
    pos = 12
    out = {}
    for _ in range(nef):
        (ef,) = struct.unpack_from("<I", buf, pos); pos += 4
        ids = np.full((nq, ef), 0xFFFFFFFF, np.uint32)
        ds = np.full((nq, ef), np.inf, np.float32)
        cnt = np.zeros(nq, np.uint32)
        for i in range(nq):
            (c,) = struct.unpack_from("<I", buf, pos); pos += 4
            ids[i, :c] = np.frombuffer(buf, "<u4", c, pos); pos += 4 * c
            ds[i, :c] = np.frombuffer(buf, "<f4", c, pos); pos += 4 * c
            cnt[i] = c
        out[ef] = (ids, ds, cnt)
    return out


def ref_driver_path(build: str = "sse"):
    """The unmodified reference compiled by oracle/Makefile, if it is present and executable."""
    p = os.path.join(_HERE, "_ref", f"ref_driver_{build}")
    return p if os.path.exists(p) and os.access(p, os.X_OK) else None
