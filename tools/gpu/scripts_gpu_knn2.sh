#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_knn.py tests/test_mains.py -x -q -m gpu > gpurun_out/t_knn.log 2>&1
rc=$?; echo "rc=$rc"; tail -n 4 gpurun_out/t_knn.log
if [ $rc -ne 0 ]; then exit 1; fi
PROBE_SG=0 PROBE_QTS=16 timeout -k 10 600 python tools/perf_probe.py > gpurun_out/probe.log 2>&1
echo "rc=$?"; tail -n 3 gpurun_out/probe.log
