#!/usr/bin/env python3
"""Batched KNN scan time for a few settings of the insertion-mode parameters (dev tool)."""
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
nb = n // batch
for flush, enter in ((4, 16), (8, 16), (8, 32), (16, 32)):
    os.environ["LOCREC_KNN_FLUSH"] = str(flush)      # (read when the index is created)
    os.environ["LOCREC_KNN_ENTER"] = str(enter)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
    ix.topk_range_async(0, batch, 0.5, 0.5, k)
    ix.synchronize()
    r0 = ix.replayed_intervals()
    per = []
    t_all = time.perf_counter()
    for b in range(nb):
        t0 = time.perf_counter()
        ix.topk_range_async(b * batch, batch, 0.5, 0.5, k)
        ix.synchronize()
        per.append(time.perf_counter() - t0)
    dt = time.perf_counter() - t_all
    print(f"flush every {flush:2d}, enter at {enter:2d}: all {nb} batches {dt:.3f} s -> {nb * batch * (n - 1) / dt / 1e9:.1f} G pairs/s; "
          f"batch 0/1/2 {per[0] * 1e3:.1f}/{per[1] * 1e3:.1f}/{per[2] * 1e3:.1f} ms, median {np.median(per) * 1e3:.1f}, last {per[-1] * 1e3:.1f}; "
          f"replayed intervals {ix.replayed_intervals() - r0}", flush=True)
    ix.close()
