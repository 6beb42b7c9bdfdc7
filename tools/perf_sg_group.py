#!/usr/bin/env python3
"""Independent graphs iterated together: every graph on its own stream vs one locrec_sg_group (dev tool).
PERF_GRAPHS graphs of PERF_PERSONS persons each, 100 sweeps per request."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

CASES = ((2000, 200, 16), (2000, 200, 64), (20000, 1000, 16), (280000, 10000, 8))
if os.environ.get("PERF_CASES"):   # e.g. PERF_CASES=1 under rocprofv3: only the 64 small graphs
    CASES = tuple(CASES[int(i)] for i in os.environ["PERF_CASES"].split(","))
for persons, places, n in CASES:
    specs = [synth.sg_dataset(n_persons=persons, n_places=places, n_categories=20, seed=0x700 + i) for i in range(n)]
    graphs = [pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"]) for g in specs]
    targets = [int(g["first_person"]) for g in specs]
    streams = [torch.cuda.Stream() for _ in graphs]
    for h, st in zip(graphs, streams):
        h.set_stream(st.cuda_stream)
    grp = pkg.SgGroup(graphs)

    def run_streams():
        for h, v in zip(graphs, targets):
            h.sweeps_async(v, 0.15, 100)
        for h in graphs:
            h.synchronize()

    def run_group():
        grp.sweeps_async(targets, 0.15, 100)
        grp.synchronize()

    out = {}
    for name, run in (("per-graph streams", run_streams), ("one group", run_group)):
        run()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        out[name] = n * 500 / (time.perf_counter() - t0)
    edges = graphs[0].info()["edges"]
    print(f"{n} graphs of {edges} edges: " + ", ".join(f"{k} {v / 1e3:.1f} k graph-iterations/s" for k, v in out.items()), flush=True)
    grp.close()
    for h in graphs:
        h.close()
