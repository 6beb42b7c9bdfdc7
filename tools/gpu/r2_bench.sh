#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 ${1:-900} python bench.py ${2:-} > gpurun_out/bench.log 2> gpurun_out/bench.err; rc=$?
tail -c 6000 gpurun_out/bench.log; tail -n 5 gpurun_out/bench.err
exit $rc
