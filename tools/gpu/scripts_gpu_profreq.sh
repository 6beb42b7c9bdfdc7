#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_req
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_req -- python tools/perf_host.py > gpurun_out/prof_req.log 2>&1
echo "rc=$?"; grep -E "locrec_knn|locrec_sg" gpurun_out/prof_req.log
f=$(find gpurun_out/prof_req -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-70,160-400
find gpurun_out/prof_req -name "*kernel_trace.csv" -delete
