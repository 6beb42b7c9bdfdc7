#!/bin/bash
# full -m gpu suite with per-test durations
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 ${1:-1100} python -m pytest tests -m gpu -x -q --durations=15 ${2:-} > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 40 gpurun_out/pytest_gpu.log
exit $rc
