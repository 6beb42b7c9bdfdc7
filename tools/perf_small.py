#!/usr/bin/env python3
"""Request latency at the reference's own sample scale (cfg1: 10 k persons x ~1 k places) and a few sizes up (dev tool)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

for n, places in ((2_000, 1_000), (10_000, 1_000), (50_000, 10_000), (200_000, 100_000)):
    d = synth.knn_dataset(n, places, seed=0x5EED0001)
    d["r_rowptr"], d["r_place"] = d["p_rowptr"], d["p_idx"].astype(np.int64)
    d["r_rating"] = 1 + d["r_place"] % 5
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    pid = int(d["person_ids"][n // 3])
    out = {}
    for name, fn in (("query", lambda: ix.query(pid, 0.5, 0.5, 50)), ("recommend", lambda: ix.recommend(pid, 0.5, 0.5, 50))):
        fn()
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            fn()
            lat.append((time.perf_counter() - t0) * 1e3)
        out[name] = (float(np.median(lat)), float(np.max(lat)))
    print(f"{n} persons x {places} places: query median {out['query'][0]:.4f} ms (max {out['query'][1]:.3f}), "
          f"recommend median {out['recommend'][0]:.4f} ms (max {out['recommend'][1]:.3f})", flush=True)
    ix.close()
