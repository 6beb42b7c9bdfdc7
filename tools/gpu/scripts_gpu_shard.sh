#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {
  local t=$1 log=$2; shift 2
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "[$log] rc=$rc"; tail -n ${TAILN:-12} "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping"; exit 99; fi
  return 0
}
run 400 t_sg.log python -m pytest tests/test_gpu_sg.py -x -q -m gpu
run 300 rccl1.log python tools/rccl_one_rank.py
LOCREC_BENCH_BACKEND=gloo run 600 bench_mp.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --persons 200000 --batch 8192 --no-cpu
