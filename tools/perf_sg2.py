#!/usr/bin/env python3
"""SG wall time per iteration with and without per-launch event profiling (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
from locations_recommender_amd import synth
g = synth.sg_dataset()
v = int(g["first_person"])
sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
sg.sweeps_async(v, 0.15, 100); sg.synchronize()
for prof in (False, True, False):
    sg.profile_enable(prof)
    t0 = time.perf_counter()
    for _ in range(10):
        sg.sweeps_async(v, 0.15, 100)
    t1 = time.perf_counter()
    sg.synchronize()
    dt = time.perf_counter() - t0
    if prof:
        sg.profile_read()
    print(f"profiling {prof}: {dt/1000*1e6:.2f} us/iteration wall (host enqueue {(t1-t0)/1000*1e6:.2f} us/iteration)", flush=True)
sg.close()
