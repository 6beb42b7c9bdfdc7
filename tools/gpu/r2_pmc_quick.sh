#!/bin/bash
# two SQ counter passes over the batched KNN scan (PROBE_WHAT=knn) - quick look, not the committed profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmcq
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_INST_VMEM SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcq/p$i -- python tools/pmc_knn.py > gpurun_out/pmcq_$i.log 2>&1
  rc=$?; echo "pass $i rc=$rc"; [ $rc -eq 124 ] && exit 99
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcq/p*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); seen=set(); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        key=(k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k]+=1
    for k,v in agg.items():
        if "scan" in k or "ht_" in k:
            print(k, "x", cnt[k], {c: round(x/cnt[k]) for c,x in v.items()})
PY
find gpurun_out/pmcq -name "*.csv" -size +1M -delete
