#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_prep.py -m gpu -x -q > gpurun_out/pytest_prep.log 2>&1; rc=$?
tail -n 40 gpurun_out/pytest_prep.log
exit $rc
