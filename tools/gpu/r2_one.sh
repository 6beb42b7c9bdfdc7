#!/bin/bash
# one test file / selection on the GPU box:  r2_one.sh "<pytest args>"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest $1 -m gpu -x -q > gpurun_out/pytest_one.log 2>&1; rc=$?
tail -n 40 gpurun_out/pytest_one.log
exit $rc
