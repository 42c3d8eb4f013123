#!/bin/bash
# sweep an environment knob over the three workloads: VAR=name VALUES="a b c"
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in ${VALUES}; do
  for wl in ${WORKLOADS:-sponza-class san-miguel-class dragon-class}; do
    env ${VAR}=$v timeout -k 10 400 python -u bench.py --steps ${STEPS:-4} --warmup 1 --workload $wl --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/sweep.log 2>&1 || { tail -30 gpurun_out/sweep.log; exit 1; }
    grep '^{' gpurun_out/sweep.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('${VAR}=$v', '$wl', d['value'],'Mrays/s', d['ms_per_step'],'ms/step; extend', d['roofline']['mrays_per_s'], 'connect', d['kernels']['connect']['mrays_per_s'], 'other ms', d['kernels']['other_ms_per_step'])
" | tee -a gpurun_out/sweep_results.txt
  done
done
