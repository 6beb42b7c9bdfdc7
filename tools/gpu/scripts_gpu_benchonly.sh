#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python bench.py "$@" > gpurun_out/bench.log 2>&1
echo "rc=$?"; tail -n 4 gpurun_out/bench.log
