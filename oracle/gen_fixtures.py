#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE
(oracle/_ref/ref_driver_{sse,scalar}, built by oracle/Makefile from /root/reference/src).

Runs only in the development container (the reference tree does not travel).  For each
case it generates a synthetic base and query set (hnsw_amd.datagen; only a checksum of
the data is stored), lets the reference build the index with its own insert(), dumps the
graph as flat arrays, and records the complete W_ (ids and float distances) that the
reference's aproximateKnn produced for every query and ef.

    python oracle/gen_fixtures.py            # all cases
"""
from __future__ import annotations

import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnsw_amd import datagen  # noqa: E402
from hnsw_amd.flatgraph import FlatGraph, PAD  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")

# name: kind, d, n, nq, (M, Mmax, efC), driver build, ef list
CASES = {
    "int_d32_n2000": dict(kind="sift", d=32, n=2000, nq=200, ctor=(16, 16, 200), build="sse", efs=[10, 64]),
    "int_d128_n4000": dict(kind="sift", d=128, n=4000, nq=100, ctor=(16, 16, 200), build="sse", efs=[10, 64, 200]),
    "unit_d100_n2000_sse": dict(kind="glove", d=100, n=2000, nq=200, ctor=(16, 16, 200), build="sse", efs=[10, 64]),
    "unit_d100_n2000_scalar": dict(kind="glove", d=100, n=2000, nq=100, ctor=(16, 16, 200), build="scalar", efs=[10, 64]),
    "unit_d96_n3000": dict(kind="deep", d=96, n=3000, nq=100, ctor=(16, 16, 200), build="sse", efs=[20, 100]),
    "int_d16_n1000_m5": dict(kind="sift", d=16, n=1000, nq=100, ctor=(5, 5, 20), build="sse", efs=[3, 10, 40],
                             savegraph=True),
    "int_d32_n1500_m32": dict(kind="sift", d=32, n=1500, nq=100, ctor=(32, 32, 60), build="sse", efs=[10, 64]),   # lists of 64 / 32
    "gistlike_d960_n600": dict(kind="gist", d=960, n=600, nq=40, ctor=(16, 16, 200), build="sse", efs=[10, 64]),
}
# adjacency hashes only (pins the restated builder at a larger size)
HASH_CASES = {
    "int_d128_n20000": dict(kind="sift", d=128, n=20000, ctor=(16, 16, 200), build="sse"),
    "unit_d96_n10000": dict(kind="deep", d=96, n=10000, ctor=(16, 16, 200), build="sse"),
    "unit_d100_n10000_scalar": dict(kind="glove", d=100, n=10000, ctor=(16, 16, 200), build="scalar"),
}


def parse_graph(path: str) -> FlatGraph:
    raw = np.fromfile(path, dtype=np.uint32)
    magic, n, d, M, Mmax, Mmax0, nl = (int(x) for x in raw[:7])
    assert magic == 0x31464748
    pos = 7
    adj0 = np.full((n, Mmax0), PAD, np.uint32)
    cnt0 = np.zeros(n, np.uint32)
    up_off = [0, 0]
    up_order, up_down, up_cnt, up_slot, up_ord = [], [], [], [], []
    ep_slot, ep_order = [], []
    for l in range(nl):
        nn, eps, epo = (int(x) for x in raw[pos:pos + 3]); pos += 3
        ep_slot.append(eps); ep_order.append(epo)
        if l == 0:
            assert nn == n
        for s in range(nn):
            order, down, deg = (int(x) for x in raw[pos:pos + 3]); pos += 3
            pairs = raw[pos:pos + 2 * deg].reshape(deg, 2); pos += 2 * deg
            if l == 0:
                assert order == s, "layer-0 slot is not the node_order"
                cnt0[s] = deg
                adj0[s, :deg] = pairs[:, 1]
            else:
                up_order.append(order); up_down.append(down); up_cnt.append(deg)
                rs = np.full(Mmax, PAD, np.uint32); ro = np.full(Mmax, PAD, np.uint32)
                rs[:deg] = pairs[:, 0]; ro[:deg] = pairs[:, 1]
                up_slot.append(rs); up_ord.append(ro)
        if l >= 1:
            up_off.append(up_off[-1] + nn)
    assert pos == raw.size
    u = lambda a: np.array(a, dtype=np.uint32)
    return FlatGraph(n=n, d=d, M=M, Mmax=Mmax, deg0=Mmax0, degu=Mmax, nlayers=nl, adj0=adj0, cnt0=cnt0,
                     up_off=u(up_off[:nl + 1] if nl >= 1 else [0]), up_order=u(up_order), up_down=u(up_down),
                     up_cnt=u(up_cnt), up_adj_slot=u(up_slot).reshape(-1, Mmax), up_adj_order=u(up_ord).reshape(-1, Mmax),
                     ep_slot=u(ep_slot), ep_order=u(ep_order))


from oracle.oracle import parse_ref_results as parse_results  # noqa: E402


def make_data(case):
    base = datagen.make(case["kind"], case["n"], seed=7, stream=0, d=case["d"])
    q = datagen.make(case["kind"], case.get("nq", 1), seed=7, stream=1, d=case["d"])
    return base, q


def run_case(name: str, case: dict, tmp: str) -> None:
    base, q = make_data(case)
    bpath, qpath = os.path.join(tmp, "b.bin"), os.path.join(tmp, "q.bin")
    base.tofile(bpath); q.tofile(qpath)
    drv = os.path.join(REF, f"ref_driver_{case['build']}")
    M, Mmax, efC = case["ctor"]
    common = [bpath, str(case["n"]), str(case["d"]), str(M), str(Mmax), str(efC)]
    gpath, rpath, spath = (os.path.join(tmp, x) for x in ("g.out", "r.out", "s.out"))
    subprocess.check_call([drv, "build"] + common + [gpath] + ([spath] if case.get("savegraph") else []))
    subprocess.check_call([drv, "query"] + common + [qpath, str(case["nq"]), rpath] + [str(e) for e in case["efs"]])
    g = parse_graph(gpath)
    res = parse_results(rpath)
    out = g.to_npz_dict()
    meta = dict(name=name, kind=case["kind"], seed=7, n=case["n"], d=case["d"], nq=case["nq"], ctor=list(case["ctor"]),
                build=case["build"], efs=case["efs"], base_sha=datagen.checksum(base), query_sha=datagen.checksum(q),
                adjacency_hash=g.adjacency_hash(), layer_sizes=g.layer_sizes())
    for ef, (ids, ds, cnt) in res.items():
        out[f"ids_ef{ef}"] = ids; out[f"dist_ef{ef}"] = ds; out[f"cnt_ef{ef}"] = cnt
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    if case.get("savegraph"):
        # the reference's own saveGraph output (src/hnsw.cpp:1178-1240): input for the loadGraph reader test
        os.replace(spath, os.path.join(GOLD, name + ".savegraph.bin"))
    print(name, "layers", g.layer_sizes(), "hash", meta["adjacency_hash"],
          "size", os.path.getsize(os.path.join(GOLD, name + ".npz")))


def run_hash_case(name: str, case: dict, tmp: str) -> dict:
    base = datagen.make(case["kind"], case["n"], seed=7, stream=0, d=case["d"])
    bpath, gpath = os.path.join(tmp, "b.bin"), os.path.join(tmp, "g.out")
    base.tofile(bpath)
    M, Mmax, efC = case["ctor"]
    subprocess.check_call([os.path.join(REF, f"ref_driver_{case['build']}"), "build", bpath, str(case["n"]), str(case["d"]),
                           str(M), str(Mmax), str(efC), gpath])
    g = parse_graph(gpath)
    rec = dict(case, seed=7, base_sha=datagen.checksum(base), adjacency_hash=g.adjacency_hash(),
               layer_sizes=g.layer_sizes(), ctor=list(case["ctor"]))
    print(name, rec)
    return rec


def main():
    os.makedirs(GOLD, exist_ok=True)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)
    only = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        for name, case in CASES.items():
            if only and name not in only:
                continue
            run_case(name, case, tmp)
        if not only or "hashes" in only:
            recs = {name: run_hash_case(name, case, tmp) for name, case in HASH_CASES.items()}
            with open(os.path.join(GOLD, "adjacency_hashes.json"), "w") as f:
                json.dump(recs, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
