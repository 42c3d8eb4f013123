#!/bin/bash
# wide-form parity tests, then the traversal form swept over the workloads
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_wide.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test_wide.log 2>&1 || { tail -40 gpurun_out/test_wide.log; exit 1; }
tail -3 gpurun_out/test_wide.log
rm -f gpurun_out/sweep_results.txt
VAR=RTCORE_TRAVERSAL VALUES="2 3" bash scripts/gpu_sweep.sh
