#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
PERF_N=1000000 PERF_PLACES=100000 timeout -k 10 300 python tools/perf_create.py > gpurun_out/perf_create_1m.log 2>&1
grep -v amdgpu.ids gpurun_out/perf_create_1m.log | tail -n 8
PERF_N=10000000 PERF_PLACES=1000000 PERF_SKIP_HOST=${SKIP_HOST:-} timeout -k 10 700 python tools/perf_create.py > gpurun_out/perf_create_10m.log 2>&1
grep -v amdgpu.ids gpurun_out/perf_create_10m.log | tail -n 8
