#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PROBE_SG=0 timeout -k 10 800 python tools/perf_probe.py > gpurun_out/probe.log 2>&1
echo "rc=$?"; tail -n 16 gpurun_out/probe.log
