#!/bin/bash
# groups of small graphs: throughput of both forms, then the kernel trace of the 64-small-graphs case
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 400 python tools/perf_sg_group.py > gpurun_out/sg_group.log 2>&1 || exit 1
grep graphs gpurun_out/sg_group.log
rm -rf gpurun_out/sgg_trace
PERF_CASES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sgg_trace -- python3 tools/perf_sg_group.py > gpurun_out/sgg_trace.log 2>&1
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
f=$(find gpurun_out/sgg_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/sgg_kernel_stats.csv && head -8 "$f"
rm -rf gpurun_out/sgg_trace
