#!/bin/bash
# SG iteration runs replayed as hipGraphs: parity tests, then the SG bench legs with and without replay
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_sg.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/pytest_sg.log 2>&1; rc=$?
tail -n 5 gpurun_out/pytest_sg.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu --no-formats --steps 2 > gpurun_out/sg_graph_$i.log 2>&1 || exit 1
LOCREC_SG_NO_GRAPH=1 timeout -k 10 300 python bench.py --no-cpu --no-formats --steps 2 > gpurun_out/sg_nograph_$i.log 2>&1 || exit 1
done
python - <<'PY'
import json
for f in ("sg_graph_1","sg_nograph_1","sg_graph_2","sg_nograph_2"):
    l=[x for x in open(f"gpurun_out/{f}.log") if x.startswith("{")][-1]
    j=json.loads(l); sg=j.get("sg",{})
    print(f, j["value"], sg.get("value"), sg.get("ms_per_iteration"), sg.get("roofline",{}).get("frac"), sg.get("batched",{}).get("value"), sg.get("batched",{}).get("graph_iterations_per_s_by_form"))
PY
