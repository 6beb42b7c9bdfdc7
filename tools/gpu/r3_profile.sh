#!/bin/bash
# The committed profile of round 3: (1) rocprofv3 --kernel-trace --stats of the bench command, (2) --pmc passes (own runs,
# no other trace domain) of the batched KNN scan, the SG sweep and the single-request scan - now with the per-CLASS VALU
# instruction counters (SQ_INSTS_VALU_INT32 / _CVT / _MUL_F16 / ...), (3) the per-class issue rates of tools/ubench_valu.hip
# and the same class counters on its kernels (which counter does each opcode land in), all reduced by
# tools/make_pmc_json.py to gpurun_out/profile/r03_pmc.json + text summaries.  (The counter passes launch the SG
# iterations one by one, LOCREC_SG_NO_GRAPH: same kernels, no hipGraph replay between the profiler and the dispatches.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${ROUND_TAG:-r03}
OUT=gpurun_out/profile; rm -rf $OUT; mkdir -p $OUT/pmc
if [ -z "$SKIP_TRACE" ]; then
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-formats > $OUT/${R}_bench_under_rocprof.log 2>&1
rc=$?; echo "kernel-trace rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${R}_kernel_stats.csv && head -12 "$f"
rm -rf $OUT/trace
fi
CLS1="SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_TRANS_F16"
CLS2="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
for leg in knn sg scan1; do
  mkdir -p $OUT/pmc/$leg; i=0
  case $leg in scan1) prog=tools/pmc_scan1.py;; *) prog=tools/pmc_knn.py;; esac
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "$CLS1" "$CLS2"; do
    i=$((i+1))
    if [ $leg != knn ] && [ $i -ge 6 ]; then continue; fi     # the class split is the batched scan's question
    LOCREC_SG_NO_GRAPH=1 PROBE_WHAT=$leg PROBE_OUT=$OUT/pmc/$leg timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc/$leg/p$i -- python3 $prog > $OUT/pmc/$leg/p$i.log 2>&1
    rc=$?; echo "pmc $leg pass $i rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
  done
done
# the issue-rate table, and which class counter every opcode of it lands in
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu 2>/dev/null || exit 98
timeout -k 10 200 /tmp/ubench_valu > $OUT/${R}_ubench_valu.log 2>&1 || exit 99
mkdir -p $OUT/pmc/ubench; i=0
for set in "$CLS1" "$CLS2"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc/ubench/p$i -- /tmp/ubench_valu classes > $OUT/pmc/ubench/p$i.log 2>&1
  rc=$?; echo "pmc ubench pass $i rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
done
python3 tools/make_pmc_json.py $OUT/pmc $OUT $R > $OUT/make_pmc.log 2>&1; tail -n 60 $OUT/make_pmc.log
rm -rf $OUT/pmc
