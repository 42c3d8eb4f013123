#!/bin/bash
# sweep two environment knobs over the workloads: VAR1=.. VALUES1=".." VAR2=.. VALUES2=".."
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for a in ${VALUES1}; do
 for b in ${VALUES2}; do
  for wl in ${WORKLOADS:-sponza-class san-miguel-class dragon-class}; do
    env ${VAR1}=$a ${VAR2}=$b timeout -k 10 400 python -u bench.py --steps ${STEPS:-3} --warmup 1 --workload $wl --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/sweep.log 2>&1 || { tail -30 gpurun_out/sweep.log; exit 1; }
    grep '^{' gpurun_out/sweep.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('${VAR1}=$a ${VAR2}=$b', '$wl', d['value'],'Mrays/s', d['ms_per_step'],'ms/step; extend', d['roofline']['mrays_per_s'], 'connect', d['kernels']['connect']['mrays_per_s'], 'build s', d['config']['bvh_build_s'])
" | tee -a gpurun_out/sweep_results.txt
  done
 done
done
