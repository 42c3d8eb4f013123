#!/bin/bash
# whole GPU suite with the phase-voting traversal forced, then the traversal sweep
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
RTCORE_TRAVERSAL=3 timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test_mode3.log 2>&1 || { tail -40 gpurun_out/test_mode3.log; exit 1; }
tail -3 gpurun_out/test_mode3.log
rm -f gpurun_out/sweep_results.txt
VAR=RTCORE_TRAVERSAL VALUES="3" bash scripts/gpu_sweep.sh
VAR=RTCORE_TRAVERSAL_DEEP VALUES="1 3" bash scripts/gpu_sweep.sh
