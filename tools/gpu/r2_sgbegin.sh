#!/bin/bash
# one set-up launch per request (sg_begin / sg_begin_group): SG parity tests, group throughput, SG bench legs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sg.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q -k "not knn and not random_index and not random_large" > gpurun_out/pytest_sg.log 2>&1; rc=$?
tail -n 5 gpurun_out/pytest_sg.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/perf_sg_group.py > gpurun_out/sg_group.log 2>&1 || exit 1
grep graphs gpurun_out/sg_group.log
timeout -k 10 300 python bench.py --no-cpu --no-formats --steps 2 > gpurun_out/sg_begin_bench.log 2>&1 || exit 1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/sg_begin_bench.log") if x.startswith("{")][-1]
j=json.loads(l); sg=j.get("sg",{})
print(j["value"], sg.get("value"), sg.get("ms_per_iteration"), sg.get("roofline",{}).get("frac"), sg.get("batched",{}).get("value"), sg.get("batched",{}).get("graph_iterations_per_s_by_form"))
PY
