#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_tr
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tr -- python bench.py --no-cpu --no-sg --steps 2 > gpurun_out/prof_tr.log 2>&1
echo "rc=$?"
f=$(find gpurun_out/prof_tr -name "*kernel_trace.csv" | head -1)
python - "$f" 2>&1 <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    full = r["Kernel_Name"]
    name = next((k for k in ("knn_aggregate", "knn_scan1", "knn_final1", "knn_agg_compact", "knn_collect1", "knn_select1") if k in full), "other")
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"))
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    if k[0] != "other":
        print(k, "n", len(v), "avg us", sum(v) / len(v) / 1e3, "min", min(v) / 1e3, "max", max(v) / 1e3)
PY
find gpurun_out/prof_tr -name "*kernel_trace.csv" -delete
