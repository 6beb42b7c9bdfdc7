#!/bin/bash
# full -m gpu suite, then the bench (no cpu leg) - the usual check after a kernel change
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -n 16 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py --no-cpu --steps 3 > gpurun_out/bench_quick.log 2> gpurun_out/bench_quick.err; rc=$?
python - <<'PY'
import json
for line in open("gpurun_out/bench_quick.log"):
    if line.startswith("{"):
        d = json.loads(line); r = d["knn_request"]; sg = d.get("sg") or {}
        print("KNN G pairs/s", round(d["value"] / 1e9, 1), "step ms", round(d["ms_per_step"], 2), "scan ms", round(d["roofline"]["avg_launch_ms"], 2),
              "valu frac", d["roofline"]["frac"], "create_s", d["config"]["create_s"])
        print("request: recommend ms", r["ms_per_request"], "query ms", r["find_similar_persons_ms"], "scan1 frac", r["scan_roofline"]["frac"])
        print("formats", d.get("knn_other_formats"))
        print("sg", {k: sg.get(k) for k in ("value", "iteration_frac_of_hbm_peak")}, (sg.get("roofline") or {}).get("frac"), (sg.get("batched") or {}).get("frac_of_hbm_peak"))
PY
exit $rc
