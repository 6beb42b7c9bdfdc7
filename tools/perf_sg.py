#!/usr/bin/env python3
"""SG sweep timing at several kernel settings (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
from locations_recommender_amd import synth
g = synth.sg_dataset()
v = int(g["first_person"])
envs = [{}] if os.environ.get("PERF_SG_SKIP_PERSIST") else [{}, {"LOCREC_SG_PERSIST": "1"}]
if os.environ.get("PERF_SG_PPW_SWEEP"):
    envs = [{"LOCREC_SG_PPW": str(k)} for k in (1, 2, 4)] + [{"LOCREC_SG_GS": str(k)} for k in (2, 3, 4, 6, 8)]
for env in envs:
    for k in ("LOCREC_SG_NO_COL16", "LOCREC_SG_PPW", "LOCREC_SG_PERSIST", "LOCREC_SG_GS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
    sg.sweeps_async(v, 0.15, 100); sg.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        sg.sweeps_async(v, 0.15, 100)
    sg.synchronize()
    print(f"{env}: {(time.perf_counter() - t0) / 1000 * 1e6:.2f} us/iteration unprofiled", flush=True)
    sg.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        sg.sweeps_async(v, 0.15, 100)
    sg.synchronize()
    dt = time.perf_counter() - t0
    ms, launches = sg.profile_read()
    print(f"{env}: {dt/500*1e6:.2f} us/iteration wall, sweep kernel {ms/launches*1e3:.2f} us, "
          f"{sg.info()['sweep_bytes']/(ms/launches)/1e6:.0f} GB/s", flush=True)
    sg.close()
