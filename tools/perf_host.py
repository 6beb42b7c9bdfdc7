#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points: results handed back in
host memory, as the JNI shim receives them.  Never bench.py's `value`."""
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
t0 = time.time()
d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
t1 = time.time()
r_place = d["p_idx"].astype(np.int64)  # placeRatings: one row per (person, visited place), rating 1..5
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                  d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["p_rowptr"], r_place, 1 + r_place % 5)
print(f"knn: generate {t1 - t0:.1f} s, locrec_knn_create (host CSR -> device layout, PCIe included) {time.time() - t1:.2f} s", flush=True)
ids = np.ascontiguousarray(d["person_ids"][:3 * batch])
ix.query_batch(ids[:batch], 0.5, 0.5, k)
t0 = time.perf_counter()
for b in range(1, 3):
    ix.query_batch(ids[b * batch:(b + 1) * batch], 0.5, 0.5, k)
dt = (time.perf_counter() - t0) / 2
print(f"locrec_knn_query_batch: {batch} person ids in, {batch}x{k} (id, similarity) out to host: {dt * 1e3:.1f} ms "
      f"-> {batch * (n - 1) / dt / 1e9:.1f} G pairs/s PCIe-inclusive", flush=True)
t0 = time.perf_counter()
off, places, est = ix.recommend_batch(ids[:batch], 0.5, 0.5, k)
dt = time.perf_counter() - t0
print(f"locrec_knn_recommend_batch: {batch} person ids in, {off[-1]} (place, rating) rows out to host: {dt * 1e3:.1f} ms "
      f"-> {batch * (n - 1) / dt / 1e9:.1f} G pairs/s PCIe-inclusive", flush=True)
pid = int(d["person_ids"][5000])
for name, fn in (("locrec_knn_query", lambda: ix.query(pid, 0.5, 0.5, k)),
                 ("locrec_knn_recommend", lambda: ix.recommend(pid, 0.5, 0.5, k))):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt * 1e3:.3f} ms per request -> {(n - 1) / dt / 1e9:.2f} G pairs/s", flush=True)
ix.close()

g = synth.sg_dataset(seed=0x5EED0003)
t1 = time.time()
sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
print(f"sg: locrec_sg_create {time.time() - t1:.2f} s", flush=True)
v = int(g["first_person"])
sg.recommend(v, 0.15, 0.0, 100)
t0 = time.perf_counter()
for _ in range(5):
    ids_, probs, it, conv = sg.recommend(v, 0.15, 0.0, 100)
dt = (time.perf_counter() - t0) / 5
print(f"locrec_sg_recommend(eps=0, 100 iterations), {len(ids_)} rows back to host: {dt * 1e3:.2f} ms "
      f"-> {100 / dt:.0f} iterations/s PCIe-inclusive", flush=True)
t0 = time.perf_counter()
for _ in range(5):
    ids_, probs, it, conv = sg.recommend(v, 0.15, 0.01, 1000)
dt = (time.perf_counter() - t0) / 5
print(f"locrec_sg_recommend(eps=0.01): converged={conv} in {it} iterations, {dt * 1e3:.2f} ms per request", flush=True)
sg.close()
