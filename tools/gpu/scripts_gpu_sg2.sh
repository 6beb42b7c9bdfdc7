#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sg.py tests/test_mains.py -x -q -m gpu > gpurun_out/t_sg.log 2>&1
rc=$?; echo "rc=$rc"; tail -n 6 gpurun_out/t_sg.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python tools/perf_host.py > gpurun_out/perf_host.log 2>&1
echo "rc=$?"; grep -E "locrec_" gpurun_out/perf_host.log
