#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python tools/perf_request.py > gpurun_out/perf_request.log 2>&1
echo "rc=$?"; tail -8 gpurun_out/perf_request.log
