#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python tools/probe_cfg4.py > gpurun_out/cfg4.log 2>&1
echo "rc=$?"; tail -n 12 gpurun_out/cfg4.log
