#!/usr/bin/env python3
"""cfg3: the two-launch iteration against the fused iteration (LOCREC_SG_FUSED=1; its three kernels on one stream or on
two): us per iteration, the main sweep kernel's own duration, and agreement of the results."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
from locations_recommender_amd import synth
g = synth.sg_dataset()
v = int(g["first_person"])
res = {}
variants = [("two launches", {}), ("fused 1 stream", {"LOCREC_SG_FUSED": "1", "LOCREC_SG_FUSED_ONE_STREAM": "1"}),
            ("fused 2 streams", {"LOCREC_SG_FUSED": "1"})]
for name, env in variants:
    os.environ.pop("LOCREC_SG_FUSED", None)
    os.environ.pop("LOCREC_SG_FUSED_ONE_STREAM", None)
    os.environ.update(env)
    sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
    sg.sweeps_async(v, 0.15, 100); sg.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        sg.sweeps_async(v, 0.15, 100)
    sg.synchronize()
    us = (time.perf_counter() - t0) / 1000 * 1e6
    sg.profile_enable(True)
    for _ in range(3):
        sg.sweeps_async(v, 0.15, 100)
    sg.synchronize()
    ms, launches = sg.profile_read()
    sg.profile_enable(False)
    sg.sweeps_async(v, 0.15, 100)
    res[name] = sg.fetch()
    lat = []
    for pv in range(v, v + 20):
        t0 = time.perf_counter()
        r = sg.recommend(pv, 0.15, 0.01, 20)
        lat.append(time.perf_counter() - t0)
    res[name + " eps"] = r
    print(f"{name:15s}: {us:6.2f} us/iteration ({1e6 / us / 1e3:.1f} k it/s), sweep kernel {ms / launches * 1e3:.2f} us x {launches}, "
          f"request at eps 0.01: {np.median(lat) * 1e3:.3f} ms, {r[2]} iterations, converged {r[3]}", flush=True)
    sg.close()
last = variants[-1][0]
a, b = res["two launches"], res[last]
print("ids equal", np.array_equal(a[0], b[0]), "iterations", a[2:], b[2:], "max rel diff", float(np.max(np.abs(a[1] - b[1]) / a[1])))
a, b = res["two launches eps"], res[last + " eps"]
print("eps run: ids equal", np.array_equal(a[0], b[0]), "iterations", a[2:], b[2:], "max rel diff", float(np.max(np.abs(a[1] - b[1]) / a[1])))
