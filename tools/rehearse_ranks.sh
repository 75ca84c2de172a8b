#!/bin/bash
# Rehearsal of the multi-rank bench on a ONE-GPU box: N ranks share device 0, halos travel host-staged over gloo (RCCL
# refuses two ranks on one device).  Exercises partitioning, the self-launcher, the pipelined exchange with the second
# stream, and the halo block of the JSON line for the process grids of 2 and 4 ranks (at most 6 processes may use the
# GPU on the test boxes; the 8-rank 1x2x4 grid is rehearsed on CPU by tests/test_dist_gloo.py).
#   bash tools/rehearse_ranks.sh [cubes per axis per rank, default 16]
# A second line per rank count: --workload c4 (config 4's set-up, global mesh of C4 cubes per axis, default 48).
N=${1:-16}
C4=${2:-48}
for R in 2 4; do   # (6 ranks + a straggler of the previous run trip the test boxes' limit of 6 GPU processes)
 for W in c3 c4; do
  SEIGEN_DIST_BACKEND=gloo SEIGEN_HIP_DEVICE=0 timeout -k 10 300 python bench.py --gpus $R --n $N --workload $W --c4-cubes $C4 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l)
    h = d['halo']
    print('ranks', d['n_gpus'], 'grid', d['config']['block_grid'], d['scaling'], 'M DoF-updates/s', round(d['value']), 'ms/step', round(d['ms_per_step'], 3),
          'rank ms/step min/max', round(d['rank_ms_per_step']['min'], 3), round(d['rank_ms_per_step']['max'], 3),
          'exchanges/step', h['exchanges_per_step'], 'kernel ms/step', [round(v, 3) for v in h['kernel_ms_per_step']],
          'waited for traces ms/step', [round(v, 3) for v in h['exposed_wait_ms_per_step']],
          'host blocked ms/step', [round(v, 3) for v in h['host_blocked_ms_per_step']], '|', d['config']['workload'])
except Exception:
    print('FAILED:', l[-400:]); sys.exit(1)
" || exit 1
  sleep 3
 done
done
