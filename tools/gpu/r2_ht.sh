#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_index" > gpurun_out/pytest_fuzz.log 2>&1; rc=$?
tail -n 8 gpurun_out/pytest_fuzz.log
[ $rc -eq 124 ] && exit 99
timeout -k 10 400 python tools/perf_ht.py > gpurun_out/perf_ht.log 2>&1; rc2=$?
cat gpurun_out/perf_ht.log | tail -n 12
exit $((rc + rc2))
