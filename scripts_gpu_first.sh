#!/bin/bash
# first GPU contact: smoke, parity tests, short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { # run <seconds> <log> <cmd...>; stops the whole script after a timeout
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "[$log] rc=$rc"; tail -n 25 "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping"; exit 99; fi
  return 0
}
rocminfo | grep -m2 gfx
run 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
run 600 sg.log python -m pytest tests/test_gpu_sg.py -x -q -m gpu
run 900 knn.log python -m pytest tests/test_gpu_knn.py -x -q -m gpu -k "not full_size"
