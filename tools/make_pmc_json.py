#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/gpu/r2_profile.sh to profiles-ready files:

  <out>/r02_pmc.json       what bench.py's roofline reads (keyed by the hash of the kernel sources)
  <out>/r02_pmc_<leg>.txt  the per-kernel counter means of every pass, human readable

usage: make_pmc_json.py <pmc dir with knn/ sg/ scan1/ sub-directories> <out dir>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash)

LEGS = {
    # leg -> (kernel-name prefix of the dominant kernel, json key, workload)
    "knn": ("knn_scan_ht", "knn_scan", {"persons": 1_000_000, "places": 100_000, "batch": 16_384, "k": 50}),
    "sg": ("sg_sweep", "sg_sweep", None),
    "scan1": ("knn_scan1", "knn_scan1", {"persons": 1_000_000, "places": 100_000, "k": 50}),
}


def reduce_leg(d):
    """{kernel: {counter: mean per dispatch}}, {kernel: dispatches} over every pass directory of a leg."""
    means, counts = collections.defaultdict(dict), {}
    for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = collections.defaultdict(set)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            counts[k] = len(seen[k])
            for c, x in v.items():
                means[k][c] = x / len(seen[k])
    return means, counts


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80]


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    rec = {"source_hash": bench.kernel_source_hash(), "hash_covers": list(bench.PMC_SOURCES),
           "how": "rocprofv3 --pmc, one pass per counter set (tools/gpu/r2_profile.sh); means per dispatch"}
    for leg, (prefix, key, workload) in LEGS.items():
        d = os.path.join(src, leg)
        if not os.path.isdir(d):
            continue
        means, counts = reduce_leg(d)
        with open(os.path.join(out, f"r02_pmc_{leg}.txt"), "w") as fh:
            fh.write(f"rocprofv3 --pmc passes, leg '{leg}', kernel sources {rec['source_hash']}; means per dispatch\n")
            own = [k for k in means if "rocprim" not in k and "hipcub" not in k]   # (library sorts of the index build: omitted)
            for k in sorted(own, key=lambda k: -means[k].get("GRBM_GUI_ACTIVE", 0))[:14]:
                fh.write(f"\n{short(k)}  x{counts[k]}\n")
                for c in sorted(means[k]):
                    fh.write(f"    {c:28s} {means[k][c]:18.0f}\n")
        hit = [k for k in means if short(k).startswith(prefix)]
        if not hit:
            continue
        k = max(hit, key=lambda k: means[k].get("GRBM_GUI_ACTIVE", 0))
        m = means[k]
        if workload is None:
            with open(os.path.join(d, "workload.json")) as fh:
                workload = json.load(fh)
        rec[key] = {"kernel": short(k), "dispatches": counts[k], "workload": workload,
                    "insts_valu": m.get("SQ_INSTS_VALU"), "insts_salu": m.get("SQ_INSTS_SALU"),
                    "insts_lds": m.get("SQ_INSTS_LDS"), "fetch_kib": m.get("FETCH_SIZE"),
                    "write_kib": m.get("WRITE_SIZE"), "gui_active_cycles": m.get("GRBM_GUI_ACTIVE"),
                    "tcc_hit": m.get("TCC_HIT_sum"), "tcc_miss": m.get("TCC_MISS_sum")}
    with open(os.path.join(out, "r02_pmc.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
