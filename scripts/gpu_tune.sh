#!/bin/bash
# knob sweeps after a kernel change (environment overrides of the context defaults)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
rm -f gpurun_out/sweep_results.txt
VAR=RTCORE_MIN_ACTIVE VALUES="32 40 56 60" STEPS=3 bash scripts/gpu_sweep.sh
VAR=RTCORE_COOP_STEPS VALUES="256 1024" STEPS=3 WORKLOADS="san-miguel-class dragon-class" bash scripts/gpu_sweep.sh
VAR=RTCORE_SORT_BITS VALUES="15 18 24" STEPS=3 WORKLOADS="san-miguel-class dragon-class" bash scripts/gpu_sweep.sh
VAR=RTCORE_SORT_EXTEND_FROM VALUES="1 3" STEPS=3 WORKLOADS="san-miguel-class dragon-class" bash scripts/gpu_sweep.sh
VAR=RTCORE_TRAVERSAL_DEPTH0 VALUES="2" STEPS=3 bash scripts/gpu_sweep.sh
VAR=RTCORE_TRAVERSAL_DEEP VALUES="3" STEPS=3 WORKLOADS="san-miguel-class dragon-class" bash scripts/gpu_sweep.sh
