#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python bench.py --no-cpu "$@" > gpurun_out/bench_env.log 2>&1
echo "rc=$?"; tail -n 1 gpurun_out/bench_env.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('knn', d['value'] / 1e9, 'G pairs/s; request', d['knn_request'])
s = d['sg']
print('sg', s['value'], 'it/s', s['ms_per_iteration'] * 1e3, 'us; sweep', s['roofline']['avg_launch_ms'] * 1e3, 'us')
print('batched', s['batched']['value'], s['batched']['frac_of_hbm_peak'])
"
