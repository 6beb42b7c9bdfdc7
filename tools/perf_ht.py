#!/usr/bin/env python3
"""cfg2 batch step: head / tail form against the row scan (dev tool).  PERF_ENVS = ';'-separated env sets."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()
from locations_recommender_amd import shard, synth

n = int(os.environ.get("PERF_N", "1000000"))
places = int(os.environ.get("PERF_PLACES", "100000"))
batch = int(os.environ.get("PERF_BATCH", "16384"))
steps = int(os.environ.get("PERF_STEPS", "4"))
d = synth.knn_dataset(n, places, seed=0x5EED0002)
d["r_rowptr"], d["r_place"] = d["p_rowptr"], d["p_idx"].astype(np.int64)
d["r_rating"] = 1 + d["r_place"] % 5
envs = [dict(kv.split("=") for kv in e.split(",") if kv) for e in os.environ.get("PERF_ENVS", ";LOCREC_KNN_BLOCKS=1024;LOCREC_KNN_BLOCKS=4096;LOCREC_KNN_HT_H=256;LOCREC_KNN_HT_H=1024").split(";")]
ref = None
for env in envs:
    for k in list(os.environ):
        if k.startswith("LOCREC_KNN_") or k.startswith("LOCREC_DEBUG_"):
            del os.environ[k]
    os.environ.update(env)
    t0 = time.perf_counter()
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    create = time.perf_counter() - t0
    nb = n // batch
    ix.recommend_range_async(shard.query_batch_of(0, 0, 1, nb) * batch, batch, 0.5, 0.5, 50)
    ix.synchronize()
    ix.profile_enable(True)
    t0 = time.perf_counter()
    same = os.environ.get("PERF_SAME_BATCH") is not None     # every step scans batch 0 again (seeding experiments)
    for i in range(steps):
        ix.recommend_range_async(shard.query_batch_of(0 if same else 1 + i, 0, 1, nb) * batch, batch, 0.5, 0.5, 50)
    ix.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms, launches = ix.profile_read()
    ix.profile_enable(False)
    ids, sims, cnt = ix.fetch_topk(batch, 50)
    same = None
    if ref is None:
        ref = (ids, sims, cnt)
    else:
        same = bool(np.array_equal(ids, ref[0]) and np.array_equal(sims, ref[1]) and np.array_equal(cnt, ref[2]))
    print(f"{env}: {ix.scan_kernel_name()} create {create:.2f} s, step {dt * 1e3:.2f} ms = {batch * (n - 1) / dt / 1e9:.0f} G pairs/s, "
          f"scan kernel {ms / max(1, launches):.2f} ms, replays {ix.replayed_intervals()}, same as first: {same}", flush=True)
    ix.close()
