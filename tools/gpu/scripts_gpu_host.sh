#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python tools/perf_host.py > gpurun_out/perf_host.log 2>&1
echo "rc=$?"; grep -E "locrec_knn|locrec_sg" gpurun_out/perf_host.log
