#!/bin/bash
# Per-kernel durations and HBM bytes of the hexahedral lane kernels (run on the GPU box from the repo root):
#   bash tools/experiments/hex_profile.sh <tag> <P> <N>
set -u
TAG=${1:-hexprof}; P=${2:-2}; N=${3:-96}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export SEIGEN_HIP_PATH=${SEIGEN_HIP_PATH:-lane}
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 tools/experiments/hex_throughput.py $P $N quadrilateral > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run -- python3 tools/experiments/hex_throughput.py $P $N quadrilateral > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run -- python3 tools/experiments/hex_throughput.py $P $N quadrilateral > $OUT/write.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 tools/pmc_summary.py $OUT/fetch $OUT/write > $OUT/pmc_summary.txt 2>&1
cat $OUT/trace.log
head -12 $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/pmc_summary.txt | head -30
