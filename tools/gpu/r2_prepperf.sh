#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python tools/perf_prep.py > gpurun_out/perf_prep.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/perf_prep.log | tail -n 14
exit $rc
