#!/usr/bin/env python3
"""The benchmarked cfg2 step on an index with 0.1 % wide rows (per-row format fallback), for a kernel trace:
   rocprofv3 --kernel-trace --stats -- python3 tools/perf_wide.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft

pkg = graft.load_package()
from locations_recommender_amd import shard, synth

n, places, k, batch = 1_000_000, 100_000, 50, 16_384
d = synth.knn_dataset(n, places, 0x5EED0002)
frac = float(os.environ.get("WIDE_FRAC", "0.001"))
rng = np.random.default_rng(77)
wide = np.sort(rng.choice(n, int(n * frac), replace=False)) if frac > 0 else np.array([], np.int64)
v = d["p_val"].copy()
for r in wide:
    v[d["p_rowptr"][r] + rng.integers(0, d["p_rowptr"][r + 1] - d["p_rowptr"][r])] = 300.0
d["p_val"] = v
d["r_rowptr"], d["r_place"] = d["p_rowptr"], d["p_idx"].astype(np.int64)
d["r_rating"] = 1 + d["r_place"] % 5
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"], d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"],
                  d["r_rowptr"], d["r_place"], d["r_rating"])
nb = n // batch
start = int(os.environ.get("START", "1"))
ix.recommend_range_async(shard.query_batch_of(start, 0, 1, nb) * batch, batch, 0.5, 0.5, k)
ix.synchronize()
steps = int(os.environ.get("STEPS", "4"))
per = []
ix.profile_enable(True)
for i in range(steps):
    b = shard.query_batch_of(start + 1 + i, 0, 1, nb)
    t0 = time.perf_counter()
    ix.recommend_range_async(b * batch, batch, 0.5, 0.5, k)
    ix.synchronize()
    nw = int(np.count_nonzero(np.isin(ix.row_person_ids(b * batch, batch), d["person_ids"][wide]))) if len(wide) else 0
    per.append((b, (time.perf_counter() - t0) * 1e3, nw))
scan_ms, launches = ix.profile_read()
print(f"scan kernel {scan_ms / max(1, launches):.2f} ms x {launches}")
print(f"{len(wide)} wide rows: " + ", ".join(f"batch {b}: {ms:.2f} ms ({nw} wide queries)" for b, ms, nw in per) +
      f"; mean {np.mean([p[1] for p in per]):.2f} ms per step, plan {ix.scan_kernel_name()}, image {ix.ht_image_info()}")
ix.close()
