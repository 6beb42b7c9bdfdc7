#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for flags in "--no-aggregate" ""; do
  LOCREC_DEBUG_TIMING=1 timeout -k 10 400 python bench.py --no-cpu --no-sg $flags > gpurun_out/bisect.log 2> gpurun_out/bisect.err || { echo "failed"; tail -5 gpurun_out/bisect.err; exit 1; }
  grep "locrec recommend" gpurun_out/bisect.err | tail -3
  tail -n 1 gpurun_out/bisect.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$flags', d['value'] / 1e9, d['knn_request'])"
done
