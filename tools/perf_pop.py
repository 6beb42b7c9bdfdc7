#!/usr/bin/env python3
"""Batched scan and single-request scan for several sizes of the popular-index table (dev tool)."""
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
for h in os.environ.get("POP_HS", "0,2048,4096,8192,16384").split(","):
    os.environ.pop("LOCREC_KNN_NO_POP", None)
    os.environ.pop("LOCREC_KNN_POP_H", None)
    if h == "0":
        os.environ["LOCREC_KNN_NO_POP"] = "1"
    else:
        os.environ["LOCREC_KNN_POP_H"] = h
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
    ix.topk_range_async(0, batch, 0.5, 0.5, k)
    ix.synchronize()
    t0 = time.perf_counter()
    for b in (3, 20, 37, 54):
        ix.topk_range_async(b * batch, batch, 0.5, 0.5, k)
    ix.synchronize()
    dt = (time.perf_counter() - t0) / 4
    ix.topk_range_async(12345, 1, 0.5, 0.5, k)
    ix.synchronize()
    ix.profile_enable(True)
    for i in range(10):
        ix.topk_range_async(1000 + i * 777, 1, 0.5, 0.5, k)
    ms, launches = ix.profile_read()
    print(f"pop table {h:>5}: batch {dt * 1e3:.2f} ms -> {batch * (n - 1) / dt / 1e9:.1f} G pairs/s; single scan {ms / launches * 1e3:.1f} us", flush=True)
    ix.close()
