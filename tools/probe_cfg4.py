#!/usr/bin/env python3
"""cfg4-scale check on ONE GPU (BASELINE.json configs[3] shape: 10 M persons x 1 M places, seed
0x5EED0004; one GPU holds the full candidate set exactly as each of the 8 ranks would): index
creation, one batch of queries, sampled parity against the oracle, throughput."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

n = int(os.environ.get("PROBE_N", "10000000"))
places = int(os.environ.get("PROBE_PLACES", "1000000"))
batch = int(os.environ.get("PROBE_BATCH", "16384"))
workers = int(os.environ.get("PROBE_WORKERS", "16"))
k = 50
t0 = time.time()
d = synth.knn_dataset_parallel(n, places, 0x5EED0004, workers=workers)
print(f"generated {n} persons x {places} places, nnz {len(d['p_idx'])} + {len(d['c_idx'])} in {time.time() - t0:.0f} s", flush=True)
t0 = time.time()
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                  d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
info = ix.info()
print(f"locrec_knn_create {time.time() - t0:.1f} s; info {info}", flush=True)
ix.topk_range_async(0, batch, 0.5, 0.5, k)
ix.synchronize()
ix.profile_enable(True)
t0 = time.perf_counter()
steps = 2
for b in range(1, 1 + steps):
    ix.topk_range_async(b * batch * 7, batch, 0.5, 0.5, k)
ix.synchronize()
dt = (time.perf_counter() - t0) / steps
ms, launches = ix.profile_read()
ix.profile_enable(False)
print(f"batch of {batch} queries vs {n} candidates: {dt * 1e3:.1f} ms/step -> {batch * (n - 1) / dt / 1e9:.1f} G pairs/s; "
      f"scan kernel {ms / launches:.1f} ms, effective {batch * info['scan_bytes'] / (ms / launches) / 1e6:.0f} GB/s "
      f"({info['scan_bytes'] / n:.1f} B/pair)", flush=True)
ids, sims, counts = ix.fetch_topk(batch, k)
rows_pid = ix.row_person_ids(steps * batch * 7, batch)
import oracle_binding as ob  # noqa: E402
pid_to_row = None
sample = np.linspace(0, batch - 1, 16).astype(np.int64)
qrows = (rows_pid[sample] - d["person_ids"][0]).astype(np.int64)  # person ids are contiguous: row = id - first id
t0 = time.time()
oi, os_, oc = ob.knn_similar_batch(d, qrows, 0.5, 0.5, k, nthreads=min(16, len(os.sched_getaffinity(0))))
print(f"oracle: 16 queries in {time.time() - t0:.1f} s", flush=True)
bad = 0
for j, s in enumerate(sample):
    c = int(counts[s])
    if c != int(oc[j]) or not np.array_equal(ids[s, :c], oi[j, :c]) or not np.array_equal(sims[s, :c], os_[j, :c]):
        bad += 1
print(f"sampled parity vs oracle: {16 - bad}/16 queries bit-identical (ids and similarities)", flush=True)
# the reference's own operator: one person per call
pid = int(d["person_ids"][n // 2])
ix.query(pid, 0.5, 0.5, k)
t0 = time.perf_counter()
for _ in range(10):
    a, b = ix.query(pid, 0.5, 0.5, k)
dt = (time.perf_counter() - t0) / 10
o1, o2, o3 = ob.knn_similar_batch(d, np.array([n // 2], np.int64), 0.5, 0.5, k, nthreads=1)
ok = np.array_equal(a, o1[0, :int(o3[0])]) and np.array_equal(b, o2[0, :int(o3[0])])
print(f"locrec_knn_query: {dt * 1e3:.3f} ms per request ({(n - 1) / dt / 1e9:.1f} G pairs/s), matches oracle: {ok}", flush=True)
ix.close()
assert bad == 0 and ok
print("CFG4_OK", flush=True)
