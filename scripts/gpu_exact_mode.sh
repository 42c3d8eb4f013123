#!/bin/bash
# bench lines with the exact traversal forms (the reference's tree in the reference's order;
# the wave hand-off for stragglers stays on, it keeps distances exact)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
export RTCORE_TRAVERSAL_DEPTH0=0 RTCORE_TRAVERSAL_DEEP=1 RTCORE_FAST_TREE_BUCKETS=0
STEPS=${STEPS:-2} BENCH_ARGS="--no-cpu-baseline" WORKLOADS="${WORKLOADS:-sponza-class dragon-class san-miguel-class}" bash scripts/gpu_bench_some.sh
