#!/bin/bash
# FETCH_SIZE calibration (scripts/micro/fetch_calib.hip) on the GPU box: known byte counts, then the
# counters of the same two dispatches, condensed into gpurun_out/fetch_calibration.txt.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
OUT=gpurun_out/fetch_calib
mkdir -p $OUT scripts/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/bin/fetch_calib scripts/micro/fetch_calib.hip
timeout -k 10 120 scripts/bin/fetch_calib > $OUT/known.txt
rm -rf $OUT/p1 $OUT/p2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -o pmc -- scripts/bin/fetch_calib > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p2 -o pmc -- scripts/bin/fetch_calib > $OUT/p2.log 2>&1
python3 scripts/micro/fetch_calib_report.py $OUT > gpurun_out/fetch_calibration.txt
cat gpurun_out/fetch_calibration.txt
