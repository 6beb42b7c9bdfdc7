#!/usr/bin/env python3
"""Latency of the reference's own operator (one person per call) with ratings in the index,
before and after a batched aggregation has grown the workspaces (dev tool)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

n, k = 1_000_000, 50
d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
r_place = d["p_idx"].astype(np.int64)
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                  d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["p_rowptr"], r_place, 1 + r_place % 5)
pid = int(ix.row_person_ids(n // 2, 1)[0])


def timeit(label, fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print(f"{label}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms", flush=True)


timeit("query (fresh index)", lambda: ix.query(pid, 0.5, 0.5, k))
timeit("recommend (fresh index)", lambda: ix.recommend(pid, 0.5, 0.5, k))
ix.topk_range_async(0, 16384, 0.5, 0.5, k)
ix.synchronize()
timeit("recommend (after a topk batch)", lambda: ix.recommend(pid, 0.5, 0.5, k))
ix.recommend_range_async(0, 16384, 0.5, 0.5, k)
ix.synchronize()
timeit("recommend (after a recommend batch)", lambda: ix.recommend(pid, 0.5, 0.5, k))
ix.recommend_range_async(0, 16384, 0.5, 0.5, k)
_ = ix.fetch_topk(16384, k)
roff, rpl, rest = ix.fetch_recommend(16384)
print("rows", roff[-1], flush=True)
timeit("recommend (after fetch_recommend of the batch)", lambda: ix.recommend(pid, 0.5, 0.5, k))
del roff, rpl, rest
timeit("recommend (after freeing the fetched rows)", lambda: ix.recommend(pid, 0.5, 0.5, k))
ix.profile_enable(True)
for i in range(6):
    ix.recommend_range_async(((i * 23) % 61) * 16384, 16384, 0.5, 0.5, k)
ix.synchronize()
print("profile", ix.profile_read(), flush=True)
timeit("recommend (profiling still enabled)", lambda: ix.recommend(pid, 0.5, 0.5, k))
ix.profile_enable(False)
timeit("recommend (profiling disabled again)", lambda: ix.recommend(pid, 0.5, 0.5, k))
import torch  # noqa: E402
torch.cuda.synchronize()
timeit("recommend (after torch.cuda init)", lambda: ix.recommend(pid, 0.5, 0.5, k))
t0 = time.perf_counter()
ix.recommend_range_async(16384, 16384, 0.5, 0.5, k)
ix.synchronize()
t1 = time.perf_counter()
ix.topk_range_async(16384, 16384, 0.5, 0.5, k)
ix.synchronize()
t2 = time.perf_counter()
print(f"batch of 16384: scan+aggregate {1e3 * (t1 - t0):.2f} ms, scan only {1e3 * (t2 - t1):.2f} ms", flush=True)
ix.close()
