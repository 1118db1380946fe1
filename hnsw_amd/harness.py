"""Measurement helpers: exact ground truth, the two recall definitions, algorithmic bytes.

The recall loop of the reference's driver (src/main.cpp:236-280 of the reference) counts, per
query, how many of the first k ground-truth ids occur among the first k+1 results (its loop
breaks after pushing k+1 items); `reference_precision` reproduces that, `recall_at_k` is
the usual definition.
"""
from __future__ import annotations

import numpy as np

# the reference's ef sweep: 20,30,...,100,120,...,200 (src/main.cpp:232-234)
REFERENCE_EF_SWEEP = [20, 30, 40, 50, 60, 70, 80, 90, 100, 120, 140, 160, 180, 200]


def cpu_model() -> str:
    """Model name of the host CPU (first "model name" line of /proc/cpuinfo), or "unknown"."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores() -> int:
    """Host threads a CPU leg may use: the cores this process may run on, capped by OMP_NUM_THREADS when the
    environment sets it (a shared machine hands every job a share of its cores that way)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    if cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def recall_at_k(ids: np.ndarray, gt: np.ndarray, k: int = 10) -> float:
    ids = np.asarray(ids)[:, :k]
    gt = np.asarray(gt)[:, :k]
    hit = (ids[:, :, None] == gt[:, None, :]).any(axis=1).sum()
    return float(hit) / (ids.shape[0] * k)


def reference_precision(ids: np.ndarray, gt: np.ndarray, k: int = 10) -> float:
    ids = np.asarray(ids)[:, :k + 1]
    gt = np.asarray(gt)[:, :k]
    hit = (gt[:, :, None] == ids[:, None, :]).any(axis=2).sum()
    return float(hit) / (ids.shape[0] * k)


def algorithmic_bytes(stats: np.ndarray, d: int, d_used: int, k: int, row_elem_bytes: int = 4) -> np.ndarray:
    """Bytes a query must move, independent of layout: one row of d_used elements per distance
    (floats; bytes when the byte-row table is searched), 4 bytes per neighbour id of every list
    read, the query itself, the result."""
    s = stats
    return (s["n_dist"].astype(np.int64) * row_elem_bytes * d_used + 4 * (s["sum_deg0"].astype(np.int64) + s["sum_degu"])
            + 4 * d + 8 * k)


def exact_topk_torch(base_t, queries_t, k: int, chunk: int = 1 << 18, id_offset: int = 0):
    """Exact k nearest rows of base_t for every query by squared L2, on the tensors' device.
    ||q||^2 + ||x||^2 - 2 q.x in float32 (exact for the integer-valued sets), chunked over
    the base, running top-k.  Returns (dist [nq,k] f32, ids [nq,k] int64)."""
    import torch
    nq = queries_t.shape[0]
    qn = (queries_t * queries_t).sum(1, keepdim=True)
    best_d = torch.full((nq, k), float("inf"), device=queries_t.device)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=queries_t.device)
    for lo in range(0, base_t.shape[0], chunk):
        x = base_t[lo:lo + chunk]
        d2 = qn + (x * x).sum(1)[None, :] - 2.0 * (queries_t @ x.T)
        kk = min(k, x.shape[0])
        dd, ii = torch.topk(d2, kk, dim=1, largest=False)
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + lo + id_offset], 1)
        dd2, sel = torch.topk(cat_d, k, dim=1, largest=False)
        best_d, best_i = dd2, torch.gather(cat_i, 1, sel)
    return best_d, best_i


F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X, f32-input MFMA (spec)


def exact_topk_mfma(base_t, queries_t, k: int, id_offset: int = 0):
    """Exact k nearest rows with the library's own brute-force kernels (f32-input MFMA for the
    dot products).  Tensors on the current GPU, float32, contiguous.  Returns
    (dist [nq,k] f32 tensor, ids [nq,k] int64 tensor, info dict with the MFMA kernel's rate)."""
    import ctypes as C

    import torch

    from . import capi
    n, d = base_t.shape
    nq = queries_t.shape[0]
    out_i = torch.empty((nq, k), dtype=torch.int32, device=queries_t.device)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=queries_t.device)
    ms = C.c_float(0.0)
    capi.check(capi.lib().hnsw_amd_bruteforce_knn_device(base_t.data_ptr(), n, queries_t.data_ptr(), nq, d, k, id_offset,
                                                         out_i.data_ptr(), out_d.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream, C.byref(ms)))
    tflops = 2.0 * nq * n * d / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    info = {"mfma_kernel_ms": round(ms.value, 3), "tflops": round(tflops, 2), "peak_tflops": F32_MFMA_PEAK_TFLOPS,
            "mfma_frac": round(tflops / F32_MFMA_PEAK_TFLOPS, 4)}
    return out_d, out_i.to(torch.int64) & 0xFFFFFFFF, info
