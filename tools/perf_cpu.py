#!/usr/bin/env python3
"""Thread scaling of the CPU oracle's batched KNN scan on this host (dev tool behind bench.py's cpu_baseline)."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "active")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from locations_recommender_amd import synth  # noqa: E402
import oracle_binding as ob  # noqa: E402

n = 1_000_000
d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
avail = len(os.sched_getaffinity(0))
for th in (1, 8, 32, 64, 128, 256):
    if th > avail:
        break
    nq = max(4, 2 * th if th <= 32 else th)
    rows = np.linspace(0, n - 1, nq).astype(np.int64)
    t0 = time.perf_counter()
    ob.knn_similar_batch(d, rows, 0.5, 0.5, 50, nthreads=th)
    dt = time.perf_counter() - t0
    print(f"{th:3d} threads, {nq} queries: {dt:.2f} s -> {nq * (n - 1) / dt / 1e6:.1f} M pairs/s ({nq * (n - 1) / dt / th / 1e6:.2f} M per thread)", flush=True)
