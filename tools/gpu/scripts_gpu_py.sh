#!/bin/bash
# run one python tool: SCRIPT=tools/x.py bash tools/gpu/scripts_gpu_py.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 ${LIMIT:-800} python $SCRIPT > gpurun_out/py.log 2>&1
echo "rc=$?"; tail -n ${TAILN:-20} gpurun_out/py.log
