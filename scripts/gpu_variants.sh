#!/bin/bash
# compare library builds: VARIANTS="a b" (files rust-raytracer_amd/variants/librtcore_<name>.so; "default" = the shipped one)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in ${VARIANTS}; do
  if [ "$v" = default ]; then unset RTCORE_LIB; else export RTCORE_LIB=$PWD/rust-raytracer_amd/variants/librtcore_$v.so; fi
  for wl in ${WORKLOADS:-sponza-class san-miguel-class dragon-class}; do
    timeout -k 10 400 python -u bench.py --steps ${STEPS:-4} --warmup 1 --workload $wl --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/variant.log 2>&1 || { tail -30 gpurun_out/variant.log; exit 1; }
    grep '^{' gpurun_out/variant.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', '$wl', d['value'],'Mrays/s', d['ms_per_step'],'ms/step; extend', d['roofline']['mrays_per_s'], 'connect', d['kernels']['connect']['mrays_per_s'], 'other ms', d['kernels']['other_ms_per_step'])
" | tee -a gpurun_out/variant_results.txt
  done
done
