#!/usr/bin/env python3
"""Where the time of one SG request at the shipped parameters goes (dev tool): enqueue (set-up + iterations with
the convergence polls), the wait, the read-back; with hipGraph replay of the runs and without (LOCREC_SG_NO_GRAPH)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

g = synth.sg_dataset(seed=0x5EED0003)
rng = np.random.default_rng(7)
persons = [int(g["first_person"]) + int(x) for x in rng.integers(0, 280_000, size=60)]
for label, env in (("graph replay", None), ("single launches", "1"), ("graph replay", None), ("single launches", "1")):
    if env:
        os.environ["LOCREC_SG_NO_GRAPH"] = env
    else:
        os.environ.pop("LOCREC_SG_NO_GRAPH", None)
    h = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
    for eps, max_it in ((0.01, 20), (1e-6, 20)):
        h.recommend(persons[0], 0.15, eps, max_it)
        a, b, c, its = [], [], [], []
        for v in persons:
            t0 = time.perf_counter()
            h.iterate_async(v, 0.15, eps, max_it)
            t1 = time.perf_counter()
            h.synchronize()
            t2 = time.perf_counter()
            r = h.fetch()
            t3 = time.perf_counter()
            a.append(t1 - t0); b.append(t2 - t1); c.append(t3 - t2); its.append(r[2])
        print(f"{label:16s} eps {eps:g}: iterate_async {np.median(a) * 1e6:6.1f} us, synchronize {np.median(b) * 1e6:6.1f} us, "
              f"fetch {np.median(c) * 1e6:6.1f} us, total {np.median(np.array(a) + b + c) * 1e6:6.1f} us, iterations {np.median(its):.0f}", flush=True)
    h.close()
