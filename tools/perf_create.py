#!/usr/bin/env python3
"""locrec_knn_create at cfg2 / cfg4 size: device build vs host build, and create_from_device (dev tool).
PERF_N / PERF_PLACES select the size; LOCREC_DEBUG_TIMING=1 with the DEBUG_SWITCHES library prints phases."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()
from locations_recommender_amd import synth

n = int(os.environ.get("PERF_N", "1000000"))
places = int(os.environ.get("PERF_PLACES", "100000"))
t0 = time.perf_counter()
d = synth.knn_dataset_parallel(n, places, 0x5EED0002 if n <= 2_000_000 else 0x5EED0004, workers=16)
print(f"generated {n} x {places} in {time.perf_counter() - t0:.1f} s", flush=True)
args = (d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"], d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
for label, env in (("device build, host arrays in", {}), ("device build again", {}), ("host build", {"LOCREC_KNN_HOST_BUILD": "1"})):
    if label == "host build" and os.environ.get("PERF_SKIP_HOST"):
        continue
    os.environ.pop("LOCREC_KNN_HOST_BUILD", None)
    os.environ.update(env)
    t0 = time.perf_counter()
    ix = pkg.KnnIndex(*args)
    dt = time.perf_counter() - t0
    print(f"{label}: locrec_knn_create {dt:.2f} s", flush=True)
    ix.close()
os.environ.pop("LOCREC_KNN_HOST_BUILD", None)
import torch
t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items() if isinstance(v, np.ndarray)}
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    ix = pkg.KnnIndex.from_device(t["person_ids"], t["p_rowptr"], t["p_idx"], t["p_val"], d["p_dim"],
                                  t["c_rowptr"], t["c_idx"], t["c_val"], d["c_dim"])
    dt = time.perf_counter() - t0
    print(f"locrec_knn_create_from_device {dt:.2f} s", flush=True)
    ix.close()
