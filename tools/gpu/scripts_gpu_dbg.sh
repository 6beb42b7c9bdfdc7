#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python - > gpurun_out/dbg3.log 2>&1 <<'PY'
import sys, os, numpy as np
sys.path.insert(0,'tests')
import __graft_entry__ as g, oracle_binding as ob
pkg=g.load_package()
from locations_recommender_amd import synth
d0=synth.small_knn_dataset(n=3000,p_dim=200,seed=12)
d=synth.knn_dataset(2000, 500, seed=0x5EED0002)
oids,osims,ocnt=ob.knn_similar_batch(d,np.arange(2000),0.5,0.5,10,nthreads=8)
KEYS=("LOCREC_KNN_NO_PACK16","LOCREC_KNN_QT","LOCREC_KNN_WAVES","LOCREC_DEBUG_NOFILTER","LOCREC_KNN_FORCE_HASH")
def history():
    saved={k:os.environ.pop(k) for k in KEYS if k in os.environ}
    ix0=pkg.KnnIndex(d0["person_ids"],d0["p_rowptr"],d0["p_idx"],d0["p_val"],d0["p_dim"],d0["c_rowptr"],d0["c_idx"],d0["c_val"],d0["c_dim"])
    ix0.close()
    os.environ.update(saved)
for env in ({"LOCREC_KNN_QT":"8","LOCREC_KNN_FORCE_HASH":"1"}, {}, {"LOCREC_KNN_WAVES":"4"}, {"LOCREC_KNN_NO_PACK16":"1"}):
    for k in KEYS: os.environ.pop(k,None)
    os.environ.update(env)
    nbad=0; detail=[]
    for rep in range(10):
        history()
        ix=pkg.KnnIndex(d["person_ids"],d["p_rowptr"],d["p_idx"],d["p_val"],d["p_dim"],d["c_rowptr"],d["c_idx"],d["c_val"],d["c_dim"])
        ids,sims,cnt=ix.all_pairs_topk(0.5,0.5,10)
        bad=np.flatnonzero((ids!=oids).any(axis=1))
        nbad += len(bad)
        if len(bad): detail.append((rep, bad[:4].tolist()))
        mode=ix.info()["mode"]; ix.close()
    print(env, "mode", mode, "bad rows in 10 reps:", nbad, detail[:8], flush=True)
PY
echo "dbg3 rc=$?"; tail -6 gpurun_out/dbg3.log
