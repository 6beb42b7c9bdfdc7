#!/usr/bin/env python3
"""The longest-row query batch of cfg2 (the tile plan's worst case) with and without the 16-wave block (dev tool)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

n, batch, k = 1_000_000, 16_384, 50
d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
for env in ({"LOCREC_KNN_NO_WIDE_BLOCK": "1"}, {}):
    os.environ.pop("LOCREC_KNN_NO_WIDE_BLOCK", None)
    os.environ.update(env)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
    ix.topk_range_async(0, batch, 0.5, 0.5, k)
    ix.synchronize()
    for b in (57, 58, 59, 60):
        t0 = time.perf_counter()
        ix.topk_range_async(b * batch, batch, 0.5, 0.5, k)
        ix.synchronize()
        print(f"{env} batch {b}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    ix.close()
