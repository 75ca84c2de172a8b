#!/bin/bash
# Socket power and clocks while bench.py runs (the FP64 stages hold the package at its power cap).
# usage (GPU box): bash tools/powerprobe.sh  ->  gpurun_out/power_samples.txt
mkdir -p gpurun_out
(for i in $(seq 1 40); do
   rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '
   echo
   sleep 0.25
 done) > gpurun_out/power_samples.txt &
SPID=$!
python bench.py --steps 600 --warmup 2 --no-cpu-baseline --configs none 2>&1 | tail -1 | cut -c1-220
wait $SPID
