#!/bin/bash
# small results gathered into pinned memory by one launch (knn_pack_host): KNN parity tests, then the request latencies with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_knn.py tests/test_gpu_long_queries.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not graph" > gpurun_out/pytest_pack.log 2>&1; rc=$?
tail -n 5 gpurun_out/pytest_pack.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu --no-formats --no-sg --steps 2 > gpurun_out/pack_on_$i.log 2>&1 || exit 1
LOCREC_KNN_NO_PACK=1 timeout -k 10 300 python bench.py --no-cpu --no-formats --no-sg --steps 2 > gpurun_out/pack_off_$i.log 2>&1 || exit 1
done
python - <<'PY'
import json
for f in ("pack_on_1","pack_off_1","pack_on_2","pack_off_2"):
    l=[x for x in open(f"gpurun_out/{f}.log") if x.startswith("{")][-1]
    j=json.loads(l); r=j["knn_request"]
    print(f, "recommend ms", round(r["ms_per_request"],4), "query ms", round(r["find_similar_persons_ms"],4), "large_k", round(r["large_k"]["recommend_ms"],3), round(r["large_k"]["find_similar_persons_ms"],3))
PY
