#!/bin/bash
# rehearse the 2-rank bench path on one GPU (gloo collectives, ranks share cuda:0), then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
LOCREC_BENCH_BACKEND=gloo timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --persons 200000 --batch 8192 --no-cpu > gpurun_out/bench_mp.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/bench_mp.log | cut -c1-2500
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
