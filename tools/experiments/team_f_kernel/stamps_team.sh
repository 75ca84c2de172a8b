#!/bin/bash
# per-phase cycle stamps (build_tools/libseigen_hip_stamps.so) of the F kernels, plain and team
for t in 0 4 8; do
  echo "== TEAM=$t"
  SEIGEN_HIP_TEAM=$t SEIGEN_HIP_STAMPS=1 SEIGEN_HIP_LIB=$PWD/build_tools/libseigen_hip_stamps.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>&1 | grep -E "stamps|ms_per_step" | sed -e 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/'
done
