#!/bin/bash
# Upper bound of the "wave walks a y column, -y traces from a wave-private LDS stash" idea WITHOUT
# building it: variants of the F kernels in which the traces in question cost nothing (read from the own cell: WRONG
# results, timing and fabric bytes only).  Run from the repo root IN THE BUILD CONTAINER to build the variants
# (tools/experiments/column_walk/run.sh build), on the GPU box to measure them (... run.sh measure).
set -e
cd "$(dirname "$0")/../../.."
case "$1" in
build)
  cp seigen_amd/csrc/kernels_mfma.hip /tmp/kernels_mfma_keep.hip
  patch -p0 seigen_amd/csrc/kernels_mfma.hip < tools/experiments/column_walk/free_traces_upper_bound.patch
  for v in 1 2 3; do tools/build_variant.sh free$v -DSG_EXP_FREE_TRACES=$v; done
  cp /tmp/kernels_mfma_keep.hip seigen_amd/csrc/kernels_mfma.hip
  ;;
measure)
  out=gpurun_out/column_walk; mkdir -p $out; export TMPDIR=/tmp
  for rep in 1 2; do
    for v in base free1 free2 free3; do
      lib=""; [ $v != base ] && lib=$PWD/build_tools/libseigen_hip_$v.so
      SEIGEN_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --steps 150 --no-cpu-baseline --configs none > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err || { tail -5 $out/bench_${v}_$rep.err; exit 1; }
    done
  done
  for v in base free1 free2 free3; do
    lib=""; [ $v != base ] && lib=$PWD/build_tools/libseigen_hip_$v.so
    SEIGEN_HIP_LIB=$lib rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/fetch_$v -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --configs none > $out/fetch_$v.log 2>&1
  done
  python3 - <<PY
import json, glob, sys
sys.path.insert(0, "tools")
from pmc_summary import summarise
for v in ("base", "free1", "free2", "free3"):
    rows = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("$out/bench_%s_*.json" % v))]
    st = [[round(x, 3) for x in r["roofline"]["stage_avg_ms"]] for r in rows]
    fetch = {}
    for f in glob.glob("$out/fetch_%s/**/*counter_collection.csv" % v, recursive=True):
        for kernel, vals, n in summarise(f):
            if "stage_F" in kernel and "FETCH_SIZE" in vals:
                fetch[kernel.split("stage_")[1][:22]] = round(vals["FETCH_SIZE"] * 1024 * 2 / 1e9, 3)
    print("%-6s ms/step %s  stages %s  F fetch GB (2 x FETCH_SIZE) %s" % (v, ["%.3f" % r["ms_per_step"] for r in rows], st[-1], fetch))
PY
  ;;
*) echo "usage: run.sh build|measure"; exit 2;;
esac
