#!/bin/bash
# kernel stats of a few cfg2 batch steps (which kernels the step spends its time in)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/stats
PERF_ENVS=";" PERF_STEPS=4 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -- python3 tools/perf_ht.py > gpurun_out/stats.log 2>&1
grep "G pairs" gpurun_out/stats.log | cut -c1-200
f=$(find gpurun_out/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    n = "rocprim/hipcub kernel" if "rocprim" in n else n.split("(")[0][:50]
    print(f'{int(r["Calls"]):6d} x {float(r["AverageNs"]) / 1e3:10.1f} us  {n}')
PY
rm -rf gpurun_out/stats
