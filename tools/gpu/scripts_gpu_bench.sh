#!/bin/bash
# full parity suite + bench + rocprof kernel stats + traffic counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "[$log] rc=$rc"; tail -n ${TAILN:-6} "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping"; exit 99; fi
  return 0
}
run 1100 gpu_tests.log python -m pytest tests -x -q -m gpu
run 600 probe_sg.log python tools/perf_sg.py
run 900 bench.log python bench.py
rm -rf gpurun_out/prof_stats
run 900 prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 3 --warmup 1 --no-cpu
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
find gpurun_out/prof_stats -name "*kernel_trace.csv" -size +8M -delete
run 600 perf_host.log python tools/perf_host.py
