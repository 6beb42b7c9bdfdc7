#!/usr/bin/env python3
"""locrec_sg_create wall time at cfg3 size (dev tool): id table vs sort (LOCREC_SG_NO_DENSE_IDS), with and without the
weight dictionary (LOCREC_SG_NO_DICT)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

g = synth.sg_dataset(seed=0x5EED0003)
for label, env in (("id table", None), ("sort + bisection", "1"), ("id table", None), ("id table, fp64 weights streamed", "nodict"),
                   ("id table", None)):
    os.environ.pop("LOCREC_SG_NO_DENSE_IDS", None)
    os.environ.pop("LOCREC_SG_NO_DICT", None)
    if env == "nodict":
        os.environ["LOCREC_SG_NO_DICT"] = "1"
    elif env:
        os.environ["LOCREC_SG_NO_DENSE_IDS"] = env
    t0 = time.perf_counter()
    h = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
    dt = time.perf_counter() - t0
    print(f"locrec_sg_create, {h.info()['edges']} edges, {label}: {dt * 1e3:.1f} ms", flush=True)
    h.close()
