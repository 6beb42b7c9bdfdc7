#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOCREC_SG_GS=4 timeout -k 10 300 python -m pytest tests/test_gpu_sg.py -x -q -m gpu -k "not two_ranks" > gpurun_out/t_sg_gs.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/t_sg_gs.log
PERF_SG_PPW_SWEEP=1 timeout -k 10 300 python tools/perf_sg.py > gpurun_out/probe_sg.log 2>&1
echo "rc=$?"; tail -20 gpurun_out/probe_sg.log
