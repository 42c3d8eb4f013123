#!/bin/bash
# GPU-box check: parity tests, then a short bench.  Each GPU step has its own
# time limit; the script stops at the first failure.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread > gpurun_out/test.log 2>&1 || { tail -40 gpurun_out/test.log; exit 1; }
tail -3 gpurun_out/test.log
timeout -k 10 240 python -u bench.py --steps ${STEPS:-5} --warmup 1 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1 || { tail -40 gpurun_out/bench.log; exit 1; }
tail -2 gpurun_out/bench.log
