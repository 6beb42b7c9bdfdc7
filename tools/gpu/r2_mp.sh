#!/bin/bash
# multi-process rehearsals on one GPU: 2 gloo ranks sharing cuda:0 through bench.py, the 1-rank RCCL paths, smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LOCREC_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --persons 200000 --batch 8192 --no-cpu > gpurun_out/bench_mp.log 2>&1
rc=$?; echo "bench 2 ranks rc=$rc"; tail -1 gpurun_out/bench_mp.log | cut -c1-900
[ $rc -ne 0 ] && { tail -20 gpurun_out/bench_mp.log; exit $rc; }
timeout -k 10 300 python tools/rccl_one_rank.py > gpurun_out/rccl_one_rank.log 2>&1; rc=$?
echo "rccl one rank rc=$rc"; grep -v amdgpu.ids gpurun_out/rccl_one_rank.log | tail -8
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
