"""Flat (pointer-free) form of an HNSW graph, as numpy arrays.

This is the layout the search kernels read (see hnsw_amd/csrc/flat_index.h) and
the form graphs are exchanged in between the host index, the fixtures under
tests/golden/ and the test oracle.  It replaces the reference's per-node heap
objects (`Node`, `Layer`, `Neighbors`: src/node.h:17-135, src/layer.h:9-38 of the
reference): neighbour lists keep their order, which the search results depend on.

Layer 0: node i is the vector with node_order i; `adj0[i, :cnt0[i]]` are its
neighbours' node_orders, the row padded to deg0 = 2*Mmax with 0xFFFFFFFF.
Upper layers 1..nlayers-1 are concatenated: layer l owns the slots
[up_off[l], up_off[l+1]); for slot s, `up_order[s]` is the vector it stands for,
`up_down[s]` its slot in the layer below (relative to that layer; for layer 1 the
node_order), `up_adj_slot[s]` / `up_adj_order[s]` its neighbours as (slot inside
the same layer, node_order) pairs.  `ep_slot[l]`, `ep_order[l]`: entry point of
layer l (the node that created the layer, reference src/hnsw.cpp:324-333).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, fields

import numpy as np

PAD = np.uint32(0xFFFFFFFF)

_ARRAYS = ("adj0", "cnt0", "up_off", "up_order", "up_down", "up_cnt",
           "up_adj_slot", "up_adj_order", "ep_slot", "ep_order")


@dataclass
class FlatGraph:
    n: int
    d: int
    M: int
    Mmax: int
    deg0: int
    degu: int
    nlayers: int
    adj0: np.ndarray
    cnt0: np.ndarray
    up_off: np.ndarray
    up_order: np.ndarray
    up_down: np.ndarray
    up_cnt: np.ndarray
    up_adj_slot: np.ndarray
    up_adj_order: np.ndarray
    ep_slot: np.ndarray
    ep_order: np.ndarray

    def __post_init__(self):
        for name in _ARRAYS:
            a = np.ascontiguousarray(getattr(self, name), dtype=np.uint32)
            setattr(self, name, a)
        self.adj0 = self.adj0.reshape(self.n, self.deg0)
        nu = int(self.up_off[-1]) if self.up_off.size else 0
        self.up_adj_slot = self.up_adj_slot.reshape(nu, self.degu)
        self.up_adj_order = self.up_adj_order.reshape(nu, self.degu)
        self.validate()

    # -- integrity: everything a kernel indexes with must be in range ---------
    def validate(self) -> None:
        n, nl = self.n, self.nlayers
        if nl == 0:
            return
        assert self.up_off.shape == (nl + 1,) and self.up_off[0] == 0 and self.up_off[1] == 0
        assert self.cnt0.shape == (n,) and int(self.cnt0.max(initial=0)) <= self.deg0
        cols = np.arange(self.deg0, dtype=np.uint32)[None, :]
        live = cols < self.cnt0[:, None]
        assert bool((self.adj0[live] < n).all()), "layer-0 neighbour out of range"
        assert bool((self.adj0[~live] == PAD).all()), "layer-0 padding is not 0xFFFFFFFF"
        nu = int(self.up_off[-1])
        assert self.up_order.shape == (nu,) and self.up_down.shape == (nu,) and self.up_cnt.shape == (nu,)
        assert int(self.up_cnt.max(initial=0)) <= self.degu
        assert bool((self.up_order < n).all())
        for l in range(1, nl):
            lo, hi = int(self.up_off[l]), int(self.up_off[l + 1])
            assert hi > lo, "empty upper layer"
            size = hi - lo
            below = n if l == 1 else int(self.up_off[l] - self.up_off[l - 1])
            assert bool((self.up_down[lo:hi] < below).all())
            colsu = np.arange(self.degu, dtype=np.uint32)[None, :]
            liveu = colsu < self.up_cnt[lo:hi, None]
            assert bool((self.up_adj_slot[lo:hi][liveu] < size).all())
            assert bool((self.up_adj_order[lo:hi][liveu] < n).all())
            assert int(self.ep_slot[l]) < size and int(self.ep_order[l]) < n
        assert int(self.ep_slot[0]) < n and int(self.ep_order[0]) < n

    def layer_sizes(self) -> list[int]:
        return [self.n] + [int(self.up_off[l + 1] - self.up_off[l]) for l in range(1, self.nlayers)]

    def adjacency_hash(self) -> str:
        h = hashlib.sha256()
        for name in _ARRAYS:
            h.update(getattr(self, name).tobytes())
        return h.hexdigest()[:16]

    def same_adjacency(self, other: "FlatGraph") -> bool:
        return all(np.array_equal(getattr(self, a), getattr(other, a)) for a in _ARRAYS) and \
            (self.n, self.d, self.deg0, self.degu, self.nlayers) == (other.n, other.d, other.deg0, other.degu, other.nlayers)

    def to_npz_dict(self, prefix: str = "g_") -> dict:
        out = {}
        for f in fields(self):
            v = getattr(self, f.name)
            out[prefix + f.name] = v if isinstance(v, np.ndarray) else np.int64(v)
        return out

    @classmethod
    def from_npz(cls, z, prefix: str = "g_") -> "FlatGraph":
        kw = {}
        for f in fields(cls):
            v = z[prefix + f.name]
            kw[f.name] = v if f.name in _ARRAYS else int(v)
        return cls(**kw)
