"""GPU: the on-device k-way merge against its host statement, and the C++ drop-in class
(include/hnsw_amd.hpp) driven like the reference's own driver."""
import os
import subprocess

import numpy as np
import pytest

from hnsw_amd import HNSW, capi, datagen
from hnsw_amd.sharded import merge_topk_numpy

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("S,nq,k", [(2, 50, 10), (8, 33, 10), (3, 7, 100), (1, 5, 4)])
def test_merge_kernel_matches_host_merge(S, nq, k):
    import torch
    rng = np.random.default_rng(S * 100 + k)
    dist = np.sort(rng.integers(0, 60, size=(S, nq, k)).astype(np.float32), axis=2)   # many ties across shards
    ids = rng.integers(0, 1000, size=(S, nq, k)).astype(np.uint32)
    ids[:, ::5, k - 2:] = 0xFFFFFFFF            # short lists
    dist[:, ::5, k - 2:] = np.inf
    offsets = (np.arange(S) * 1000).astype(np.uint32)
    eg, ed = merge_topk_numpy(dist, ids, offsets, k)
    d_t = torch.from_numpy(dist).cuda()
    i_t = torch.from_numpy(ids.view(np.int32)).cuda()
    o_i = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    o_d = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    capi.check(capi.lib().hnsw_amd_merge_topk_device(d_t.data_ptr(), i_t.data_ptr(), capi.u32(offsets), S, nq, k, k, 0,
                                                     o_i.data_ptr(), o_d.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(o_d.cpu().numpy(), ed)
    assert np.array_equal(o_i.cpu().numpy().view(np.uint32), eg)


def test_cpp_drop_in_class_runs_the_reference_style_driver(tmp_path):
    exe = os.path.join(ROOT, "examples", "sift_test")
    if not os.path.exists(exe):
        pytest.fail("examples/sift_test is not built (python -c 'import __graft_entry__ as g; g.build()')")
    n, d, nq = 3000, 128, 200
    base = datagen.make("sift", n, seed=2)
    q = datagen.make("sift", nq, seed=2, stream=1)
    from oracle import oracle
    gt = oracle.bruteforce(base, q, 100).astype(np.uint32)
    bp, qp, gp = (str(tmp_path / x) for x in ("base.bin", "q.bin", "gt.bin"))
    base.tofile(bp); q.tofile(qp); gt.tofile(gp)
    out = subprocess.run([exe, bp, str(n), str(d), qp, str(nq), gp, "100"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("Precision:")]
    print("\n".join(lines))                       # includes the time of one query per call (-s shows it)
    assert len(lines) == 14                      # the reference's sweep: 20..100 step 10, 120..200 step 20
    last = lines[-1]
    one, batched = float(last.split()[1].rstrip(",")), float(last.split("batched precision:")[1].split(",")[0])
    assert one == batched and one > 0.99         # per-query calls and the batch agree; ef=200 finds everything
    # same graph through the Python mirror gives the same figure
    h = HNSW(16, 16, 200)
    h.init(d, n)
    h.insert_batch(base)
    ids, _, _ = h.search_batch(q, 200, out_k=11)
    from hnsw_amd.harness import reference_precision
    assert abs(reference_precision(ids, gt, 10) - one) < 1e-9


def test_cpp_class_serves_the_reference_public_members(tmp_path):
    """stat_, visited_, vector_data_, layers_, print(int), file-call behaviour (src/hnsw.h:221-245) through the C++ class."""
    exe = os.path.join(ROOT, "examples", "facade_check")
    if not os.path.exists(exe):
        pytest.fail("examples/facade_check is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    assert "facade ok" in out.stdout
    assert "No. precise distance computations:" in out.stdout and "was not sucesfully opened" in out.stdout


def test_cpp_hdf5_create_and_query(tmp_path):
    """The reference's HDF5 mode (create(file, "train") + query(file, "test", "neighbors", ef), src/main.cpp:140-160)
    through include/hnsw_amd.hpp on an ann-benchmarks-shaped file: the precision it prints equals the one the
    Python mirror computes on the same data."""
    from tests import hdf5_writer
    if not hdf5_writer.available():
        pytest.skip("libhdf5 not found")
    exe = os.path.join(ROOT, "examples", "hdf5_test")
    if not os.path.exists(exe):
        pytest.fail("examples/hdf5_test is not built (python -c 'import __graft_entry__ as g; g.build()')")
    n, nq = 4000, 150
    base = datagen.make("sift", n, seed=4)
    q = datagen.make("sift", nq, seed=4, stream=1)
    from oracle import oracle
    gt = oracle.bruteforce(base, q, 100).astype(np.int32)
    path = str(tmp_path / "set.hdf5")
    hdf5_writer.write(path, {"train": base, "test": q, "neighbors": gt})
    out = subprocess.run([exe, path, "1", "l2", "20", "100"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [float(l.split()[1].rstrip(",")) for l in out.stdout.splitlines() if l.startswith("Precision:")]
    assert len(got) == 2 and got[1] > 0.99 and got[0] <= got[1]
    h = HNSW(16, 16, 200)
    h.init(128, n)
    h.insert_batch(base)            # one thread, like the driver: the same graph
    from hnsw_amd.harness import reference_precision
    for ef, p in zip((20, 100), got):
        ids, _, _ = h.search_batch(q, ef, out_k=11)
        assert abs(reference_precision(ids, gt, 10) - p) < 1e-6


def test_cpp_hdf5_create_with_the_index_built_on_the_gpu(tmp_path):
    """create(..., gpu_build = true): layer 0 searched, selected and linked on the GPU; the driver's precision is that
    of a sound graph."""
    from tests import hdf5_writer
    if not hdf5_writer.available():
        pytest.skip("libhdf5 not found")
    exe = os.path.join(ROOT, "examples", "hdf5_test")
    n, nq = 20_000, 150
    base = datagen.make("sift", n, seed=6)
    q = datagen.make("sift", nq, seed=6, stream=1)
    from oracle import oracle
    gt = oracle.bruteforce(base, q, 100, threads=8).astype(np.int32)
    path = str(tmp_path / "set.hdf5")
    hdf5_writer.write(path, {"train": base, "test": q, "neighbors": gt})
    out = subprocess.run([exe, path, "8", "l2+gpu", "100"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [float(l.split()[1].rstrip(",")) for l in out.stdout.splitlines() if l.startswith("Precision:")]
    assert len(got) == 1 and got[0] > 0.99


def test_cpp_hdf5_angular_set_is_normalised_on_load(tmp_path):
    """An angular file (rows of arbitrary length, "neighbors" by cosine): with `angular` the driver scales rows and
    queries to unit length, on which squared L2 ranks like the angle; without it the raw-L2 search misses them."""
    from tests import hdf5_writer
    if not hdf5_writer.available():
        pytest.skip("libhdf5 not found")
    exe = os.path.join(ROOT, "examples", "hdf5_test")
    n, nq = 4000, 100
    rng = np.random.default_rng(5)
    unit = datagen.make("glove", n + nq, seed=5)
    raw = (unit * rng.uniform(0.2, 5.0, size=(n + nq, 1))).astype(np.float32)       # same directions, other lengths
    base, q = raw[:n], raw[n:]
    cos = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float64) @ (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float64).T
    gt = np.argsort(-cos, axis=1)[:, :100].astype(np.int32)
    path = str(tmp_path / "angular.hdf5")
    hdf5_writer.write(path, {"train": base, "test": q, "neighbors": gt})
    prec = {}
    for metric in ("angular", "l2"):
        out = subprocess.run([exe, path, "4", metric, "100"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        prec[metric] = [float(l.split()[1].rstrip(",")) for l in out.stdout.splitlines() if l.startswith("Precision:")][0]
    assert prec["angular"] > 0.9 and prec["l2"] < 0.7, prec
