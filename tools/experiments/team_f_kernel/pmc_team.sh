#!/bin/bash
# fabric reads and matrix-pipe counters of the F kernels, plain (SEIGEN_HIP_TEAM=0) against the trace-sharing team kernels
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_team
mkdir -p $OUT
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for t in ${TEAMS:-0 4 8}; do
  export SEIGEN_HIP_TEAM=$t
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch$t -o run -- $BENCH > $OUT/fetch$t.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    -d $OUT/sq$t -o run -- $BENCH > $OUT/sq$t.log 2>&1
  echo "== TEAM=$t" >> $OUT/summary.txt
  python3 tools/pmc_summary.py $OUT/fetch$t $OUT/sq$t | grep -E "stage|^#" >> $OUT/summary.txt
done
cat $OUT/summary.txt
