#!/usr/bin/env python3
"""Builds one index the device-linked way and prints the time (for profiling the build's kernels):
    python tools/build_only.py 2000000 deep"""
import os
import sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from hnsw_amd import HNSW, datagen
n = int(sys.argv[1]); kind = sys.argv[2]
base = datagen.make(kind, n, seed=1, threads=8)
h = HNSW(16, 16, 200); h.init(base.shape[1], n)
if kind != 'sift': h.set_build_float_keys(True)
t = time.time(); h.insert_batch(base, threads=16, gpu='link'); print('build', time.time() - t, flush=True)
