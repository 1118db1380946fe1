"""GPU: the brute-force exact k-NN (f32 MFMA + selection) against a CPU double-precision scan."""
import numpy as np
import pytest

from hnsw_amd import datagen, harness
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,n,nq,d,k", [("sift", 5000, 70, 128, 10), ("sift", 40000, 300, 128, 100), ("glove", 7000, 33, 100, 11),
                                            ("fmnist", 3000, 20, 784, 10), ("sift", 130, 5, 19, 128),
                                            ("deep", 70001, 200, 96, 10), ("glove", 50000, 130, 100, 10), ("sift", 20000, 129, 30, 5),
                                            # d % 32 == 4: the last four dims ride along with the last full k step
                                            ("sift", 9000, 50, 100, 10), ("sift", 3000, 17, 36, 7), ("sift", 4000, 20, 132, 10)])
@pytest.mark.parametrize("tiles_per_worker", [None, "3"])
def test_bruteforce_matches_cpu_scan(kind, n, nq, d, k, tiles_per_worker, monkeypatch):
    import torch
    if tiles_per_worker:
        monkeypatch.setenv("HNSW_AMD_BF_TILES_PER_WORKER", tiles_per_worker)      # (a block walks several column tiles)
    base = datagen.make(kind, n, seed=9, d=d)
    q = datagen.make(kind, nq, seed=9, stream=1, d=d)
    dist, ids, info = harness.exact_topk_mfma(torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda(), k, id_offset=1000)
    ids = ids.cpu().numpy() - 1000
    dist = dist.cpu().numpy()
    gt = oracle.bruteforce(base, q, k, threads=8).astype(np.int64)
    ref_d = ((q[:, None, :].astype(np.float64) - base[gt].astype(np.float64)) ** 2).sum(-1)
    kk = min(k, n)
    if kind in ("sift", "fmnist"):
        # integer-valued data: every distance exact, so the sorted distance lists agree exactly
        assert np.array_equal(dist[:, :kk].astype(np.float64), ref_d[:, :kk])
        got_d = ((q[:, None, :].astype(np.float64) - base[ids[:, :kk]].astype(np.float64)) ** 2).sum(-1)
        assert np.array_equal(got_d, ref_d[:, :kk])      # ids may differ only inside exact ties
    else:
        assert np.allclose(dist[:, :kk], ref_d[:, :kk], rtol=1e-4, atol=1e-5)
        same = (ids[:, :kk] == gt[:, :kk]).mean()
        assert same > 0.98
    assert info["mfma_kernel_ms"] > 0
