#!/bin/bash
# One measurement set: TAG=name bash scripts/gpu_profile_round.sh
#   1. the default bench line (San-Miguel-class, with the other workloads, exact-mode rates and CPU baseline)
#   2. per workload: the bench command under rocprofv3 --kernel-trace --stats (per-kernel durations)
#   3. PMC passes (own runs, no tracing flags) for the workloads in WORKLOADS
# SKIP_BENCH / SKIP_STATS / SKIP_PMC=1 leave a part out; FETCH_FACTOR (and FETCH_FACTOR_SOURCE) is the calibrated
# bytes-per-FETCH_SIZE-unit factor handed to pmc_to_json.py.
# Everything lands in gpurun_out/$TAG/; copy what is to be kept into profiles/.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${TAG:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -z "$SKIP_BENCH" ]; then
  timeout -k 10 500 python3 -u bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1 || { tail -30 $OUT/bench.log; exit 1; }
  grep '^{' $OUT/bench.log > $OUT/san_miguel_b2_bench.json
fi
[ -n "$SKIP_STATS" ] && STATS_WORKLOADS=" "
for wl in ${STATS_WORKLOADS-san-miguel-class sponza-class dragon-class}; do
  short=$(echo $wl | sed 's/-class//; s/-/_/g')
  b=2; [ $wl = sponza-class ] && b=0
  rm -rf $OUT/prof
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 5 --warmup 1 --workload $wl --no-cpu-baseline --no-extras > $OUT/bench_prof_$short.log 2>&1 || { tail -30 $OUT/bench_prof_$short.log; exit 1; }
  grep '^{' $OUT/bench_prof_$short.log > $OUT/${short}_b${b}_bench_under_rocprof.json
  cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/${short}_b${b}_kernel_stats.csv
  rm -rf $OUT/prof
done
[ -n "$SKIP_PMC" ] && WORKLOADS=" "
for wl in ${WORKLOADS-san-miguel-class sponza-class}; do
  WORKLOAD=$wl TAG=$TAG PASSES="${PASSES:-1 2 3 4 5 6 8 11}" bash scripts/gpu_pmc.sh > $OUT/pmc_$wl.log 2>&1 || { tail -30 $OUT/pmc_$wl.log; exit 1; }
  python3 scripts/pmc_to_json.py gpurun_out/pmc_${wl}_$TAG $wl > $OUT/pmc_$wl.json
  cp gpurun_out/pmc_${wl}_$TAG/summary.txt $OUT/pmc_${wl}_summary.txt
done
ls -la $OUT
[ -f $OUT/san_miguel_b2_bench.json ] && cut -c1-600 $OUT/san_miguel_b2_bench.json
