#!/bin/bash
# knob sweeps after a kernel change (environment overrides of the context defaults)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
rm -f gpurun_out/sweep_results.txt
VAR=RTCORE_MIN_ACTIVE VALUES="32 40 56" STEPS=3 bash scripts/gpu_sweep.sh
VAR=RTCORE_SORT_EXTEND_FROM VALUES="1 99" STEPS=3 WORKLOADS="san-miguel-class dragon-class" bash scripts/gpu_sweep.sh
VAR=RTCORE_SORT_CONNECT_FROM VALUES="0 2 99" STEPS=3 WORKLOADS="san-miguel-class dragon-class" bash scripts/gpu_sweep.sh
VAR=RTCORE_WIDE_LEAF_TRIS VALUES="2 4" STEPS=3 bash scripts/gpu_sweep.sh
