#!/bin/bash
# PMC passes (one rocprofv3 run per counter set, no tracing flags besides the
# implicit kernel dispatch records).  WORKLOAD=..., sets are fixed below.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
WL=${WORKLOAD:-sponza-class}
OUT=gpurun_out/pmc_$WL${TAG:+_$TAG}
mkdir -p $OUT
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  case " ${PASSES:-1 2 3 4 5 6 7 8 9 10 11} " in *" $i "*) ;; *) continue;; esac
  rm -rf $OUT/p$i
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 2 --warmup 0 --workload $WL --no-cpu-baseline --no-extras ${BENCH_ARGS} > $OUT/p$i.log 2>&1 || { tail -20 $OUT/p$i.log; exit 1; }
  echo "pass $i ok: $set"
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE TCC_EA0_RDREQ_32B_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_COUNT
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_UTCL1_STALL_MULTI_MISS_sum GRBM_GUI_ACTIVE
SETS
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
