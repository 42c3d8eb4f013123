#!/bin/bash
# GPU suite, then the wide tree's leaf size swept over the workloads
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test.log 2>&1 || { tail -40 gpurun_out/test.log; exit 1; }
tail -2 gpurun_out/test.log
rm -f gpurun_out/sweep_results.txt
VAR=RTCORE_WIDE_LEAF_TRIS VALUES="${VALUES:-8 4 3 2 1}" bash scripts/gpu_sweep.sh
