#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_scan1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_scan1/p$i -- python tools/pmc_scan1.py > gpurun_out/pmc_scan1_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_scan1_$i.log; exit 99; }
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_scan1/p*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); seen=set(); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        key=(k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k]+=1
    for k,v in agg.items():
        if "scan1" in k: print(k, "dispatches", cnt[k], {c: round(x/cnt[k]) for c,x in v.items()})
PY
find gpurun_out/pmc_scan1 -name "*kernel_trace.csv" -delete
