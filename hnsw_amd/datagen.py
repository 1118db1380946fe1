"""Deterministic synthetic datasets shaped like the ann-benchmarks sets the
reference lists (README.md:13-23 of the reference): SIFT-128, Fashion-MNIST-784,
GloVe-100, GIST-960, DEEP-96.

No real dataset is reachable offline, so every run uses these generators.  All
randomness is a counter-based 64-bit mixer evaluated with numpy integer
arithmetic, so a given (kind, seed, n) is bit-identical on every machine: the
golden fixtures under tests/golden/ only store a checksum of the data, not the
data itself.

The vectors have a low intrinsic dimension (16 latent factors plus per-element
noise) so that an HNSW graph behaves on them roughly as it does on real
descriptor data (recall grows smoothly with ef instead of collapsing as it does
on i.i.d. noise).
"""
from __future__ import annotations

import hashlib

import numpy as np

_LATENT = 16
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return x


def _irwin_hall16(h: np.ndarray) -> np.ndarray:
    """Sum of the four 16-bit fields of h, centred: integer, roughly normal,
    standard deviation 65536/sqrt(3)."""
    m = np.uint64(0xFFFF)
    s = (h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48))
    return s.astype(np.int64) - 2 * 65535


def _irwin_hall8(h: np.ndarray) -> np.ndarray:
    """Sum of four 8-bit fields of h, centred; standard deviation 256/sqrt(3)."""
    m = np.uint64(0xFF)
    s = (h & m) + ((h >> np.uint64(8)) & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(24)) & m)
    return s.astype(np.int64) - 510


def _stream(seed: int, tag: int) -> np.uint64:
    with np.errstate(over="ignore"):
        return _mix64(np.array([np.uint64(seed) * _GOLD + np.uint64(tag)], dtype=np.uint64))[0]


def integer_vectors(n: int, d: int, seed: int, stream: int = 0, *, row0: int = 0,
                    hi: int = 255, chunk: int = 1 << 16) -> np.ndarray:
    """n x d float32 matrix of integer values in [0, hi].

    `seed` fixes the mixing matrix and offsets (shared by base and query sets),
    `stream` selects an independent row stream (0 = base, 1 = queries),
    `row0` is the index of the first row (so shards can be generated apart).
    """
    # mixing matrix A[d][16] in [-32, 31], offsets b[d] in [40, 103]
    ja = np.arange(d * _LATENT, dtype=np.uint64)
    with np.errstate(over="ignore"):
        A = (_mix64(_stream(seed, 1) + ja) & np.uint64(63)).astype(np.int64) - 32
        b = (_mix64(_stream(seed, 2) + np.arange(d, dtype=np.uint64)) & np.uint64(63)).astype(np.int64) + 40
    A = A.reshape(d, _LATENT).astype(np.float64)
    zkey = _stream(seed, 16 + 2 * stream)
    ekey = _stream(seed, 17 + 2 * stream)
    out = np.empty((n, d), dtype=np.float32)
    for lo in range(0, n, chunk):
        hi_r = min(n, lo + chunk)
        rows = np.arange(row0 + lo, row0 + hi_r, dtype=np.uint64)
        with np.errstate(over="ignore"):
            zc = rows[:, None] * np.uint64(_LATENT) + np.arange(_LATENT, dtype=np.uint64)[None, :]
            z = _irwin_hall16(_mix64(zkey + zc * _GOLD)).astype(np.float64)
            ec = rows[:, None] * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :]
            e = _irwin_hall8(_mix64(ekey + ec * _GOLD))
        # |A z| < 16*32*131070 < 2^27: exact in float64 whatever the BLAS order
        t = (z @ A.T).astype(np.int64)
        v = b[None, :] + (t >> 16) + (e >> 4)
        np.clip(v, 0, hi, out=v)
        out[lo:hi_r] = v.astype(np.float32)
    return out


def _normalize_rows(x: np.ndarray, centre: float) -> np.ndarray:
    y = x.astype(np.float64) - centre
    nrm = np.sqrt((y * y).sum(axis=1, keepdims=True))
    nrm[nrm == 0] = 1.0
    return (y / nrm).astype(np.float32)


KINDS = {
    # kind: (d, value range / transform)
    "sift": 128,     # integer 0..255, Euclidean
    "fmnist": 784,   # integer 0..255, Euclidean
    "glove": 100,    # float, rows L2-normalised (angular as L2 on unit vectors)
    "gist": 960,     # float in [0, 1]
    "deep": 96,      # float, rows L2-normalised
}


def make(kind: str, n: int, seed: int = 1, stream: int = 0, *, row0: int = 0, d: int | None = None, threads: int = 1) -> np.ndarray:
    """Rows [row0, row0+n) of the synthetic set `kind` as float32 (n x d).  threads > 1: slices of the rows are generated
    side by side (every row depends on its own index only, so the result is the same); large sets are generated in slices
    in any case, which bounds the temporaries."""
    d = KINDS[kind] if d is None else d
    if n >= (1 << 18) and (threads > 1 or n > (1 << 21)):
        # (large sets in slices even on one thread: the float64 temporaries of a slice instead of those of the whole set)
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, threads)
        out = np.empty((n, d), dtype=np.float32)
        step = min(-(-n // (4 * threads)), 1 << 20)
        step = -(-step // (1 << 16)) * (1 << 16)

        def part(lo: int) -> None:
            out[lo:lo + step] = make(kind, min(step, n - lo), seed, stream, row0=row0 + lo, d=d)

        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(part, range(0, n, step)))
        return out
    x = integer_vectors(n, d, seed=seed * 1000 + sorted(KINDS).index(kind), stream=stream, row0=row0)
    if kind in ("sift", "fmnist"):
        return x
    if kind in ("glove", "deep"):
        return _normalize_rows(x, 72.0)
    if kind == "gist":
        return (x.astype(np.float64) / 255.0).astype(np.float32)
    raise ValueError(kind)


def checksum(x: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()[:16]
