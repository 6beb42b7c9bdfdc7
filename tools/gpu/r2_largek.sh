#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/largek
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/largek -- python3 tools/prof_large_k.py > gpurun_out/largek.log 2>&1; rc=$?
grep "host wall" gpurun_out/largek.log
python3 tools/prof_large_k.py --reduce gpurun_out/largek > gpurun_out/r02_large_k_trace.txt 2>&1
cat gpurun_out/r02_large_k_trace.txt
rm -rf gpurun_out/largek
exit $rc
