#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_knn.py tests/test_mains.py -x -q -m gpu > gpurun_out/t_knn.log 2>&1
rc=$?; echo "rc=$rc"; tail -n 8 gpurun_out/t_knn.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python tools/perf_request.py > gpurun_out/py.log 2>&1
echo "rc=$?"; head -n 5 gpurun_out/py.log
