#!/usr/bin/env python3
"""KNN batch throughput across the query-length spectrum (rows are sorted by length, so batch b
holds the b-th length quantile).  Dev tool behind the choice of bench.py's batch order."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

n, batch, k = 1_000_000, 16_384, 50
d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                  d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
nb = n // batch
ix.topk_range_async(0, batch, 0.5, 0.5, k)
ix.synchronize()
tot = 0.0
sel = list(range(0, nb, max(1, nb // 12))) + [nb - 1]
for b in sel:
    t0 = time.perf_counter()
    ix.topk_range_async(b * batch, batch, 0.5, 0.5, k)
    ix.synchronize()
    dt = time.perf_counter() - t0
    _, _, cnt = ix.fetch_topk(batch, k)
    ix.synchronize()
    print(f"batch {b:3d} of {nb}: {dt * 1e3:7.2f} ms -> {batch * (n - 1) / dt / 1e9:6.1f} G pairs/s", flush=True)
t0 = time.perf_counter()
for b in range(nb):
    ix.topk_range_async(b * batch, batch, 0.5, 0.5, k)
ix.synchronize()
dt = time.perf_counter() - t0
print(f"all {nb} batches ({nb * batch} queries): {dt:.2f} s -> {nb * batch * (n - 1) / dt / 1e9:.1f} G pairs/s", flush=True)
ix.close()
