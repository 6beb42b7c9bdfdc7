#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_knn.py tests/test_gpu_fuzz.py -m gpu -x -q -k "single or random_index or sharded_request or cfg2_full_size or large_k" > gpurun_out/pytest_single.log 2>&1; rc=$?
tail -n 4 gpurun_out/pytest_single.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py --no-sg --no-formats --no-cpu --steps 2 > gpurun_out/bench_single.log 2> gpurun_out/bench_single.err
python - <<'PY'
import json
for line in open("gpurun_out/bench_single.log"):
    if line.startswith("{"):
        d = json.loads(line); r = d["knn_request"]
        print("step ms", d["ms_per_step"], "| recommend ms", r["ms_per_request"], "query ms", r["find_similar_persons_ms"], "scan kernel ms", r["scan_roofline"]["avg_launch_ms"], "frac", r["scan_roofline"]["frac"], "| large_k", r["large_k"]["recommend_ms"], r["large_k"]["find_similar_persons_ms"])
PY
