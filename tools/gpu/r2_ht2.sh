#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
export PERF_ENVS="${1:-;LOCREC_KNN_HT_W=6;LOCREC_KNN_HT_W=12}"
export PERF_STEPS=4
timeout -k 10 500 python tools/perf_ht.py > gpurun_out/perf_ht2.log 2>&1
grep -v amdgpu.ids gpurun_out/perf_ht2.log | tail -n 12
