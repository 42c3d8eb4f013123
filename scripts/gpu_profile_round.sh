#!/bin/bash
# The measurement set of one round: TAG=r01_b bash scripts/gpu_profile_round.sh
#   1. bench line of the default workload, plain
#   2. the same command under rocprofv3 --kernel-trace --stats (per-kernel durations)
#   3. PMC passes (own runs, no tracing flags) for the workloads in WORKLOADS
# Everything lands in gpurun_out/$TAG/; copy what is to be kept into profiles/.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${TAG:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 python3 -u bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1 || { tail -30 $OUT/bench.log; exit 1; }
grep '^{' $OUT/bench.log > $OUT/sponza_b0_bench.json
rm -rf $OUT/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1 || { tail -30 $OUT/bench_prof.log; exit 1; }
grep '^{' $OUT/bench_prof.log > $OUT/sponza_b0_bench_under_rocprof.json
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/sponza_b0_kernel_stats.csv
rm -rf $OUT/prof
# per-kernel durations of the bounce workloads as well (where does "other" go?)
for wl in ${STATS_WORKLOADS:-san-miguel-class dragon-class}; do
  short=$(echo $wl | sed 's/-class//; s/-/_/g')
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 3 --warmup 1 --workload $wl --no-cpu-baseline > $OUT/bench_prof_$short.log 2>&1 || { tail -30 $OUT/bench_prof_$short.log; exit 1; }
  grep '^{' $OUT/bench_prof_$short.log > $OUT/${short}_b2_bench_under_rocprof.json
  cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/${short}_b2_kernel_stats.csv
  rm -rf $OUT/prof
done
for wl in ${WORKLOADS:-sponza-class san-miguel-class}; do
  WORKLOAD=$wl TAG=$TAG PASSES="${PASSES:-1 2 3 4 5 6 8}" bash scripts/gpu_pmc.sh > $OUT/pmc_$wl.log 2>&1 || { tail -30 $OUT/pmc_$wl.log; exit 1; }
  python3 scripts/pmc_to_json.py gpurun_out/pmc_${wl}_$TAG $wl > $OUT/pmc_$wl.json
  cp gpurun_out/pmc_${wl}_$TAG/summary.txt $OUT/pmc_${wl}_summary.txt
done
ls -la $OUT
cat $OUT/sponza_b0_bench.json | cut -c1-400
