#!/bin/bash
# release library: drain interval / entry threshold / waves of knn_scan_ht at the cfg2 batch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
export PERF_ENVS="${1:-;LOCREC_KNN_FLUSH=8;LOCREC_KNN_FLUSH=32;LOCREC_KNN_FLUSH=64;LOCREC_KNN_ENTER=64;LOCREC_KNN_ENTER=128;LOCREC_KNN_ENTER=16;LOCREC_KNN_FLUSH=32,LOCREC_KNN_ENTER=64}"
export PERF_STEPS=3
timeout -k 10 600 python tools/perf_ht.py > gpurun_out/perf_tune.log 2>&1
grep -v amdgpu.ids gpurun_out/perf_tune.log | tail -n 12 | cut -c1-260
