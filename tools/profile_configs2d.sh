#!/bin/bash
# Profile passes for the 2-D configs (2, 5, 1) on the MFMA tile kernels (run on the GPU box, from the repo root):
#   bash tools/profile_configs2d.sh <tag>
# pass 1: rocprofv3 --kernel-trace --stats of tools/bench_configs.py c2 c5 c1 (per-kernel durations);
# passes 2-3: FETCH_SIZE / WRITE_SIZE in their own runs (never --pmc together with other traces), + calibration.
set -u
TAG=${1:-prof2d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 tools/bench_configs.py c2 c5 c1 --steps 200 > $OUT/trace_bench.jsonl 2> $OUT/trace.log
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run -- python3 tools/bench_configs.py c2 c5 --steps 10 --warmup 2 > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run -- python3 tools/bench_configs.py c2 c5 --steps 10 --warmup 2 > $OUT/write.log 2>&1
if [ -x build_tools/calib_fetch ]; then
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/calib_f -o run -- ./build_tools/calib_fetch > $OUT/calib_f.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/calib_w -o run -- ./build_tools/calib_fetch > $OUT/calib_w.log 2>&1
fi
python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/calib_f $OUT/calib_w > $OUT/pmc_summary.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cat $OUT/trace_bench.jsonl | cut -c1-200
head -12 $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/pmc_summary.txt | cut -c1-220
