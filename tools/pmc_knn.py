#!/usr/bin/env python3
"""One KNN batch launch (and optionally SG sweeps) for counter collection under rocprofv3 --pmc."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()
from locations_recommender_amd import synth

n = int(os.environ.get("PROBE_N", "1000000"))
batch = int(os.environ.get("PROBE_BATCH", "16384"))
what = os.environ.get("PROBE_WHAT", "knn")
if what == "knn":
    d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
    for b in range(2):
        ix.topk_range_async(b * batch, batch, 0.5, 0.5, 50)
    ix.synchronize()
    if os.environ.get("PROBE_OUT"):
        import json
        with open(os.path.join(os.environ["PROBE_OUT"], "image.json"), "w") as f:
            json.dump({**ix.ht_image_info(), "tiles": (batch + ix.query_tile() - 1) // ix.query_tile()}, f)
    ix.close()
else:
    g = synth.sg_dataset()
    sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
    sg.sweeps_async(int(g["first_person"]), 0.15, 10)
    sg.synchronize()
    if os.environ.get("PROBE_OUT"):
        import json
        info = sg.info()
        with open(os.path.join(os.environ["PROBE_OUT"], "workload.json"), "w") as f:
            json.dump({"edges": info["edges"], "vertices": info["vertices"]}, f)
    sg.close()
