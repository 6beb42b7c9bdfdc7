#!/bin/bash
# full suite, bench (with cpu legs), producer timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -n 10 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/perf_prep.py > gpurun_out/perf_prep.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/perf_prep.log | tail -n 8
[ $rc -ne 0 ] && exit $rc
timeout -k 10 700 python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; rc=$?
tail -c 1500 gpurun_out/bench_full.log
exit $rc
