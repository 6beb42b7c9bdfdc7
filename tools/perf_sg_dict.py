#!/usr/bin/env python3
"""cfg3: the sweep with fp64 weights streamed (LOCREC_SG_NO_DICT=1) against its dictionary form (uint16 weight indices,
value table in LDS) at several block sizes / pieces per wave: us per iteration, the sweep kernel's own duration, and
bit-equality of the results."""
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
variants = [("fp64 stream", {"LOCREC_SG_NO_DICT": "1"})]
for thr in os.environ.get("THREADS", "256,512,1024").split(","):
    for ppw in os.environ.get("PPW", "1,2,4").split(","):
        variants.append((f"dict t={thr} ppw={ppw}", {"LOCREC_SG_DICT_THREADS": thr, "LOCREC_SG_DICT_PPW": ppw}))
for name, env in variants:
    for k in ("LOCREC_SG_NO_DICT", "LOCREC_SG_DICT_THREADS", "LOCREC_SG_DICT_PPW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
    create = time.perf_counter() - t0
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
    print(f"{name:28s}: {us:6.2f} us/iteration ({1e6 / us / 1e3:.1f} k it/s), sweep kernel {ms / launches * 1e3:.2f} us x {launches}, "
          f"create {create * 1e3:.0f} ms", flush=True)
    sg.close()
a = res["fp64 stream"]
for name, _ in variants[1:]:
    b = res[name]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], name
print("all results bit-identical")
