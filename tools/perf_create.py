#!/usr/bin/env python3
"""Phase times of locrec_knn_create at cfg2 size (LOCREC_DEBUG_TIMING=1 prints them) (dev tool)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LOCREC_DEBUG_TIMING"] = "1"
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

n = int(os.environ.get("PROBE_N", "1000000"))
d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
r_place = d["p_idx"].astype(np.int64)
for _ in range(2):
    t0 = time.time()
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["p_rowptr"], r_place, 1 + r_place % 5)
    print(f"create total {time.time() - t0:.2f} s", flush=True)
    ix.close()
