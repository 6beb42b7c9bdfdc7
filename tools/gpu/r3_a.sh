#!/bin/bash
# round 3, call A: handle-cache + JNI-shim tests, the counter list of this box, the per-class VALU issue rates
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a; export TMPDIR=/tmp
O=gpurun_out/r3a
timeout -k 10 500 python -m pytest tests/test_handle_cache.py tests/test_jni_shim.py tests/test_abi.py -x -q > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log; [ $rc -eq 124 ] && exit 99
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1); grep -c . $O/counters.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu 2>/dev/null && timeout -k 10 200 /tmp/ubench_valu > $O/ubench_valu.log 2>&1
grep CLASS $O/ubench_valu.log | head -40
