#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
LOCREC_SG_PERSIST=1 timeout -k 10 300 python tools/dbg_sg_phases.py > gpurun_out/phases.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/phases.log
