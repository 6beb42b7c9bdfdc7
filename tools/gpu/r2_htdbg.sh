#!/bin/bash
# section timing of knn_scan_ht with the DEBUG_SWITCHES build (results are wrong by design)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
export LOCREC_LIB_PATH=$PWD/locations-recommender_amd/liblocrec_dbg.so
export PERF_ENVS="${1:-;LOCREC_DEBUG_HT=1;LOCREC_DEBUG_HT=2;LOCREC_DEBUG_HT=4;LOCREC_DEBUG_HT=8;LOCREC_DEBUG_HT=15;LOCREC_DEBUG_HT=7;LOCREC_DEBUG_HT=11}"
export PERF_STEPS=2
timeout -k 10 500 python tools/perf_ht.py > gpurun_out/perf_htdbg.log 2>&1
grep -v amdgpu.ids gpurun_out/perf_htdbg.log | tail -n 12
