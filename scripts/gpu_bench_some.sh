#!/bin/bash
# bench lines of the workloads named in $WORKLOADS (one GPU)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in ${WORKLOADS:-sponza-class}; do
  timeout -k 10 400 python -u bench.py --steps ${STEPS:-5} --warmup 1 --workload $wl ${BENCH_ARGS} > gpurun_out/bench_$wl.log 2>&1 || { tail -30 gpurun_out/bench_$wl.log; exit 1; }
  grep '^{' gpurun_out/bench_$wl.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$wl', d['value'],'Mrays/s', d['ms_per_step'],'ms/step; extend', d['roofline']['mrays_per_s'], 'Mr/s', 'connect', d['kernels']['connect']['mrays_per_s'], 'other ms', d['kernels']['other_ms_per_step'], 'ext ms', d['kernels']['extend_ms_per_step'], 'con ms', d['kernels']['connect_ms_per_step'])
"
done
