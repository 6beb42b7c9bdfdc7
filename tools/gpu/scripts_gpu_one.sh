#!/bin/bash
# run a pytest selection: TESTS="tests/test_mains.py" KEXPR="..." bash tools/gpu/scripts_gpu_one.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest ${TESTS:-tests} -x -q -m gpu ${KEXPR:+-k "$KEXPR"} > gpurun_out/t_one.log 2>&1
echo "rc=$?"; tail -n 25 gpurun_out/t_one.log
