#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu -k "cfg1 or cfg4 or cfg5 or sharded" > gpurun_out/t_new.log 2>&1
echo "rc=$?"; tail -n 25 gpurun_out/t_new.log
