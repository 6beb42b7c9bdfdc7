#!/bin/bash
# full-size parity + bench + rocprof kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "[$log] rc=$rc"; tail -n ${TAILN:-15} "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping"; exit 99; fi
  return 0
}
run 900 knn_full.log python -m pytest tests/test_gpu_knn.py -x -q -m gpu -k "full_size"
run 900 bench.log python bench.py --steps 3 --warmup 1
rm -rf gpurun_out/prof_stats
run 900 prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 2 --warmup 1 --no-cpu
find gpurun_out/prof_stats -name "*kernel_stats*" | head; f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
find gpurun_out/prof_stats -name "*kernel_trace.csv" -size +20M -delete
