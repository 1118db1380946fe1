"""Compile libhnsw_amd.so (host index + HIP kernels for gfx950 + C ABI) in-tree."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhnsw_amd.so")


def build(verbose: bool = False) -> str:
    out = None if verbose else subprocess.DEVNULL
    # `checks`: the same library with the visited sets' invariants re-read and counted, for one GPU test (never loaded otherwise)
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "-j2", "all", "checks"], stdout=out)
    if not os.path.exists(LIB):
        raise RuntimeError("libhnsw_amd.so was not produced")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
