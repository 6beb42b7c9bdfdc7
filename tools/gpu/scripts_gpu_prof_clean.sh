#!/bin/bash
# kernel stats of the two headline legs alone (no concurrent-graphs leg, no cpu leg), then the
# traffic counters of both dominant kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_clean
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_clean -- python bench.py --steps 3 --warmup 1 --no-cpu --sg-graphs 0 > gpurun_out/prof_clean.log 2>&1
rc=$?; echo "prof_clean rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
f=$(find gpurun_out/prof_clean -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
find gpurun_out/prof_clean -name "*kernel_trace.csv" -delete
bash tools/gpu/scripts_gpu_pmc.sh knn || exit 99
bash tools/gpu/scripts_gpu_pmc.sh sg
