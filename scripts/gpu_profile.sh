#!/bin/bash
# rocprofv3 kernel-trace + stats of a short bench run (no PMC in this pass).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python3 bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/prof/bench_under_prof.log 2>&1 || { tail -30 gpurun_out/prof/bench_under_prof.log; exit 1; }
tail -1 gpurun_out/prof/bench_under_prof.log
find gpurun_out/prof -name "*stats*" | head
