#!/usr/bin/env python3
"""Single requests (knn_scan1 and its tail kernels) for counter collection under rocprofv3 --pmc."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

d = synth.knn_dataset(1_000_000, 100_000, seed=0x5EED0002)
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                  d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
for i in range(10):
    ix.query(int(d["person_ids"][1000 + 77_777 * i]), 0.5, 0.5, 50)
ix.close()
