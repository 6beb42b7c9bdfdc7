#!/bin/bash
# The committed profile of a round: (1) rocprofv3 --kernel-trace --stats of the bench command,
# (2) --pmc passes (own runs, no other trace domain) of the batched KNN scan, the SG sweep and the
# single-request scan, reduced to gpurun_out/profile/r02_pmc.json + text summaries.  (The counter passes launch the SG
# iterations one by one, LOCREC_SG_NO_GRAPH: same kernels, no hipGraph replay between the profiler and the dispatches.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/profile; rm -rf $OUT; mkdir -p $OUT/pmc
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-formats > $OUT/r02_bench_under_rocprof.log 2>&1
rc=$?; echo "kernel-trace rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r02_kernel_stats.csv && head -12 "$f"
rm -rf $OUT/trace
for leg in knn sg scan1; do
  mkdir -p $OUT/pmc/$leg; i=0
  case $leg in scan1) prog=tools/pmc_scan1.py;; *) prog=tools/pmc_knn.py;; esac
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    LOCREC_SG_NO_GRAPH=1 PROBE_WHAT=$leg PROBE_OUT=$OUT/pmc/$leg timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc/$leg/p$i -- python3 $prog > $OUT/pmc/$leg/p$i.log 2>&1
    rc=$?; echo "pmc $leg pass $i rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
  done
done
python3 tools/make_pmc_json.py $OUT/pmc $OUT > $OUT/make_pmc.log 2>&1; tail -n 40 $OUT/make_pmc.log
rm -rf $OUT/pmc
