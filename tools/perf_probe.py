#!/usr/bin/env python3
"""Quick on-GPU timing of the KNN scan at several tile sizes and of the SG sweep (dev tool)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()
from locations_recommender_amd import synth

n = int(os.environ.get("PROBE_N", "1000000"))
batch = int(os.environ.get("PROBE_BATCH", "16384"))
qts = [int(x) for x in os.environ.get("PROBE_QTS", "16,8").split(",")]
if os.environ.get("PROBE_KNN", "1") == "1":
    t0 = time.time()
    d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
    print(f"gen {time.time()-t0:.1f}s", flush=True)
    for qt in qts:
        os.environ["LOCREC_KNN_QT"] = str(qt)
        t0 = time.time()
        ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                          d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
        tc = time.time() - t0
        info = ix.info()
        ix.topk_range_async(0, batch, 0.5, 0.5, 50)
        ix.synchronize()
        ix.profile_enable(True)
        for b in range(3):
            ix.topk_range_async((b + 1) * batch, batch, 0.5, 0.5, 50)
        ms, launches = ix.profile_read()
        per = ms / launches
        print(f"QT<={qt} mode {info['mode']}: create {tc:.1f}s scan {per:.2f} ms/launch -> {batch*(n-1)/per/1e6:.1f} G pairs/s "
              f"eff {batch*info['scan_bytes']/per/1e6:.0f} GB/s", flush=True)
        # single query latency
        ix.topk_range_async(12345, 1, 0.5, 0.5, 50); ix.synchronize()
        ix.profile_enable(True)
        t0 = time.perf_counter()
        for i in range(20):
            ix.topk_range_async(1000 + i * 777, 1, 0.5, 0.5, 50)
        ix.synchronize()
        wall = (time.perf_counter() - t0) / 20
        ms, launches = ix.profile_read()
        print(f"   single query: {wall*1e3:.3f} ms wall, scan kernel {ms/launches*1e3:.1f} us "
              f"({info['scan_bytes']/(ms/launches)/1e6:.0f} GB/s)", flush=True)
        pid = int(ix.row_person_ids(5000, 1)[0])
        t0 = time.perf_counter()
        for i in range(10):
            ix.query(pid, 0.5, 0.5, 50)
        print(f"   query() end-to-end: {(time.perf_counter()-t0)/10*1e3:.3f} ms; ", end="")
        t0 = time.perf_counter()
        for i in range(10):
            ix.recommend(pid, 0.5, 0.5, 50)
        print(f"recommend(): {(time.perf_counter()-t0)/10*1e3:.3f} ms", flush=True)
        ix.close()
if os.environ.get("PROBE_SG", "1") == "1":
    g = synth.sg_dataset()
    sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
    v = int(g["first_person"])
    sg.sweeps_async(v, 0.15, 100); sg.synchronize()
    sg.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        sg.sweeps_async(v, 0.15, 100)
    sg.synchronize()
    dt = time.perf_counter() - t0
    ms, launches = sg.profile_read()
    print(f"SG: {dt/500*1e6:.2f} us/iteration wall, sweep kernel {ms/launches*1e3:.2f} us, "
          f"{sg.info()['sweep_bytes']/(ms/launches)/1e6:.0f} GB/s", flush=True)
    sg.close()
