#!/bin/bash
# All profile passes for config 3 (run on the GPU box, from the repo root):
#   bash tools/profile_config3.sh <tag>
# writes gpurun_out/<tag>/{trace,fetch,write,sq}/... and the calibration of the FETCH/WRITE counters.
# Counter passes are separate runs with --kernel-trace only (never --pmc together with other traces).
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# BENCH_EXTRA: more bench.py arguments (e.g. "--degree 1": the same passes for another degree on config 3's mesh)
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --configs none ${BENCH_EXTRA:-}"
# pass 1: kernel durations over a long timed region (100 steps; the 3 warm-up steps are 3 % of the launches), with the
# bench's own line - hipEvent averages of the same launches - kept next to it for the cross-check
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 bench.py --steps 100 --warmup 3 --no-cpu-baseline --configs none ${BENCH_EXTRA:-} > $OUT/trace_bench.json 2> $OUT/trace.log
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  -d $OUT/sq -o run -- $BENCH > $OUT/sq.log 2>&1
if [ -x build_tools/calib_fetch ]; then
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/calib_f -o run -- ./build_tools/calib_fetch > $OUT/calib_f.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/calib_w -o run -- ./build_tools/calib_fetch > $OUT/calib_w.log 2>&1
fi
python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/sq $OUT/calib_f $OUT/calib_w > $OUT/pmc_summary.txt 2>&1
python3 tools/make_traffic_json.py $OUT > $OUT/traffic.json 2> $OUT/traffic.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 tools/compare_rocprof_bench.py $OUT/kernel_trace.csv $OUT/trace_bench.json 3 > $OUT/rocprof_vs_bench.txt 2>&1
cat $OUT/rocprof_vs_bench.txt
