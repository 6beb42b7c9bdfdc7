#!/usr/bin/env python3
"""Large-K requests (K = 2,000,000 at cfg2) for a rocprofv3 kernel trace: which kernels a request runs.
Run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/prof_large_k.py`, then
`python3 tools/prof_large_k.py --reduce <dir>` prints the per-request kernel times (dispatches that start
after the marker file's timestamp gap, i.e. after the index build)."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), i + 1) for i, (a, b) in enumerate(zip(rows, rows[1:]))]
    big = sorted(g for g in gaps if g[0] > 0.4e9)                      # the two sleeps of 0.5 s below
    cuts = sorted(i for _, i in big[-2:])
    for label, part in (("find_similar_persons, K = 2,000,000 (5 requests)", rows[cuts[0]:cuts[1]]),
                        ("makeRecommendations, K = 2,000,000 (5 requests)", rows[cuts[1]:])):
        tot = {}
        for r in part:
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            name = "rocprim radix sort (hipcub::DeviceRadixSort)" if "radix_sort" in name or "rocprim" in name else name.split("(")[0][:60]
            tot[name] = tot.get(name, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        all_ns = sum(tot.values())
        print(f"{label}: {all_ns / 5e3:.1f} us of kernels per request")
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
            print(f"    {v / 5e3:9.1f} us  {100 * v / all_ns:5.1f} %  {k}")
    sys.exit(0)

import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from locations_recommender_amd import synth  # noqa: E402

d = synth.knn_dataset(1_000_000, 100_000, seed=0x5EED0002)
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                  d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
pid = [int(d["person_ids"][1000 + 77_777 * i]) for i in range(6)]
ix.query(pid[5], 0.5, 0.5, 2_000_000)
ix.recommend(pid[5], 0.5, 0.5, 2_000_000, capacity=200_000)          # warm: workspaces, segment table
ix.synchronize()
time.sleep(0.5)
t0 = time.perf_counter()
for p in pid[:5]:
    ix.query(p, 0.5, 0.5, 2_000_000)
t1 = time.perf_counter()
ix.synchronize()
time.sleep(0.5)
t2 = time.perf_counter()
for p in pid[:5]:
    ix.recommend(p, 0.5, 0.5, 2_000_000, capacity=200_000)
t3 = time.perf_counter()
print(f"host wall: query {1e3 * (t1 - t0) / 5:.3f} ms, recommend {1e3 * (t3 - t2) / 5:.3f} ms per request (under the profiler)")
ix.close()
