#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
WHAT=${1:-knn}
run() {
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "[$log] rc=$rc"; tail -n 3 "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping"; exit 99; fi
  return 0
}
rm -rf gpurun_out/pmc_$WHAT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PROBE_WHAT=$WHAT run 600 pmc_${WHAT}_$i.log rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_$WHAT/p$i -- python tools/pmc_knn.py
done
python - <<'PY'
import csv, glob, collections, os
what=os.environ.get("WHAT_PY","")
for f in sorted(glob.glob("gpurun_out/pmc_*/p*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    seen=set()
    for r in csv.DictReader(open(f)):
        key=(r["Kernel_Name"][:60], r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[r["Kernel_Name"][:60]]+=1
    print("==", f)
    for k,v in agg.items():
        if "scan" in k or "sweep" in k or "finalize" in k:
            print(k, "dispatches", cnt[k], {c: round(x/cnt[k]) for c,x in v.items()})
PY
find gpurun_out/pmc_$WHAT -name "*kernel_trace.csv" -delete
