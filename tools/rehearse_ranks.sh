#!/bin/bash
# Rehearsal of the multi-rank bench on a ONE-GPU box: N ranks share device 0, halos travel host-staged over gloo (RCCL
# refuses two ranks on one device).  Exercises partitioning, the self-launcher, the pipelined exchange with the second
# stream, and the halo block of the JSON line for the process grids of 2, 4 and 6 ranks (at most 6 processes may use the
# GPU on the test boxes; the 8-rank 1x2x4 grid is rehearsed on CPU by tests/test_dist_gloo.py).
#   bash tools/rehearse_ranks.sh [cubes per axis per rank, default 16]
N=${1:-16}
for R in 2 4 6; do
  SEIGEN_DIST_BACKEND=gloo SEIGEN_HIP_DEVICE=0 timeout -k 10 300 python bench.py --gpus $R --n $N --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l)
    print('ranks', d['n_gpus'], 'grid', d['config']['block_grid'], 'M DoF-updates/s', round(d['value']), 'ms/step', round(d['ms_per_step'], 3),
          'exchanges/step', d['halo']['exchanges_per_step'])
except Exception:
    print('FAILED:', l[-400:]); sys.exit(1)
" || exit 1
done
