#!/bin/bash
# two ranks on the one GPU of the box, gloo backend: exercises the N > 1 path of
# bench.py end to end (RCCL itself needs two GPUs).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/bench_2rank.log 2>&1 || { tail -30 gpurun_out/bench_2rank.log; exit 1; }
grep '^{' gpurun_out/bench_2rank.log | cut -c1-700
