#!/bin/bash
# round 3, call B: the whole GPU suite on the current sources, then the bench line with the new legs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b; export TMPDIR=/tmp
O=gpurun_out/r3b
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?
tail -8 $O/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?
tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3b/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"])
for k in ("knn_host_inclusive","knn_large_k_batched"):
    print(k, json.dumps(d.get(k))[:900])
print("through_constructor", json.dumps(d["knn_request"].get("through_constructor")))
print("sg", d["sg"]["value"], d["sg"]["ms_per_iteration"])
PY
exit $rc
