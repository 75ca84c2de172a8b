#!/bin/bash
# kernel-trace timeline of config 5 (run on the GPU box from the repo root): bash tools/profile_config5.sh <tag>
TAG=${1:-c5prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 tools/bench_configs.py c5 --steps 600 --warmup 20 > $OUT/bench.jsonl 2> $OUT/trace.log
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 tools/config5_timeline.py $OUT/kernel_trace.csv > $OUT/config5_timeline.txt 2>&1
cat $OUT/config5_timeline.txt; cut -c1-300 $OUT/bench.jsonl
