#!/usr/bin/env python3
"""Where locrec_knn_query_batch's time goes for a batch of arbitrary persons (dev tool)."""
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
ids = np.ascontiguousarray(d["person_ids"][:batch])
ix.query_batch(ids, 0.5, 0.5, k)
ix.profile_enable(True)
t0 = time.perf_counter()
ix.query_batch(ids, 0.5, 0.5, k)
dt = time.perf_counter() - t0
ms, launches = ix.profile_read()
print(f"arbitrary persons: wall {dt * 1e3:.1f} ms, scan kernel {ms:.1f} ms in {launches} launches", flush=True)
rows_ids = ix.row_person_ids(30 * batch, batch)   # one length quantile
ix.query_batch(rows_ids, 0.5, 0.5, k)
t0 = time.perf_counter()
ix.query_batch(rows_ids, 0.5, 0.5, k)
dt = time.perf_counter() - t0
ms, launches = ix.profile_read()
print(f"one length quantile: wall {dt * 1e3:.1f} ms, scan kernel {ms:.1f} ms in {launches} launches", flush=True)
ix.close()
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                  d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
ix.profile_enable(True)
for b in (0, 30, 55, 60):
    ix.topk_range_async(b * batch, batch, 0.5, 0.5, k)
    ix.fetch_topk(batch, k)
    ms, launches = ix.profile_read()
    print(f"range form, batch {b}: scan {ms:.1f} ms in {launches} launches (2 = the fast insertion path overflowed and the scan was redone); overflow blocks so far {ix.replayed_intervals()}", flush=True)
rows_ids = ix.row_person_ids(30 * batch, batch)
ix.query_batch(rows_ids, 0.5, 0.5, k)
ms, launches = ix.profile_read()
print(f"query_batch of the same rows as batch 30: scan {ms:.1f} ms in {launches} launches", flush=True)
ix.query_batch(rows_ids[:4096], 0.5, 0.5, k)
ms, launches = ix.profile_read()
print(f"query_batch of 4096 of them: scan {ms:.1f} ms in {launches} launches", flush=True)
ix.close()
