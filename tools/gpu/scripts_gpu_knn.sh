#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "[$log] rc=$rc"; tail -n ${TAILN:-12} "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping"; exit 99; fi
  return 0
}
run 900 t_knn.log python -m pytest tests/test_gpu_knn.py -x -q -m gpu
run 300 t_sgmp.log python -m pytest tests/test_gpu_sg.py -x -q -m gpu -k two_ranks
