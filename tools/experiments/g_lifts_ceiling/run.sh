#!/bin/bash
# Ceiling of "prefetch the G stage's facet traces"  WITHOUT building the prefetch: variants of the
# shipped G kernel in which the traces cost an L2 hit or nothing (WRONG results; timing and phase stamps only).
#   gfree1  traces that leave the cube read from the own cell      gfree2  every neighbour trace from the own cell
#   gfree3  no trace loads at all (the lifts' matrix work and folds alone)
# build (container): tools/experiments/g_lifts_ceiling/run.sh build     measure (GPU box): ... run.sh measure
set -e
cd "$(dirname "$0")/../../.."
case "$1" in
build)
  cp seigen_amd/csrc/kernels_mfma.hip /tmp/kernels_mfma_keep.hip
  patch -p1 < tools/experiments/g_lifts_ceiling/g_free_traces.patch
  for v in 1 2 3; do tools/build_variant.sh gfree$v -DSG_EXP_G_FREE=$v; tools/build_variant.sh gfree${v}s -DSG_EXP_G_FREE=$v -DSG_STAMPS; done
  tools/build_variant.sh gstamp -DSG_STAMPS
  cp /tmp/kernels_mfma_keep.hip seigen_amd/csrc/kernels_mfma.hip
  ;;
measure)
  out=gpurun_out/g_lifts_ceiling; mkdir -p $out
  for rep in 1 2; do
    for v in base gfree1 gfree2 gfree3; do
      lib=""; [ $v != base ] && lib=$PWD/build_tools/libseigen_hip_$v.so
      SEIGEN_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --steps 150 --no-cpu-baseline --configs none > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err || { tail -5 $out/bench_${v}_$rep.err; exit 1; }
      echo "done $v $rep"
    done
  done
  for v in gstamp gfree1s gfree2s gfree3s; do
    SEIGEN_HIP_STAMPS=1 SEIGEN_HIP_LIB=$PWD/build_tools/libseigen_hip_$v.so timeout -k 10 200 python3 bench.py --steps 20 --no-cpu-baseline --configs none > $out/stamps_$v.json 2> $out/stamps_$v.err || { tail -5 $out/stamps_$v.err; exit 1; }
    grep "stamps" $out/stamps_$v.err > $out/stamps_$v.txt || true
  done
  python3 - <<PY
import json, glob
out = "$out"
print("variant   rep  ms/step   UH1    STEMP  U1     SH1    UTEMP  S1")
for v in ("base", "gfree1", "gfree2", "gfree3"):
    for rep in (1, 2):
        r = json.loads([l for l in open("%s/bench_%s_%d.json" % (out, v, rep)) if l.startswith("{")][0])
        st = r["roofline"]["stage_avg_ms"]
        print("%-9s %d    %.3f    %s" % (v, rep, r["ms_per_step"], "  ".join("%.3f" % x for x in st)))
for v in ("gstamp", "gfree1s", "gfree2s", "gfree3s"):
    print(v)
    print(open("%s/stamps_%s.txt" % (out, v)).read())
PY
  ;;
esac
