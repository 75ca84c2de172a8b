#!/bin/bash
# A/B of hexahedral lane-kernel variants (build_tools/libseigen_hip_<name>.so): per-kernel average durations, P1 and P2 at 96^3
#   bash tools/experiments/hex_ab.sh <tag> <variant> [<variant> ...]      ("base" = the in-tree library)
set -u
TAG=$1; shift
export TMPDIR=/tmp
export SEIGEN_HIP_PATH=lane
for v in "$@"; do
  if [ "$v" = base ]; then unset SEIGEN_HIP_LIB; else export SEIGEN_HIP_LIB=$PWD/build_tools/libseigen_hip_$v.so; fi
  for P in 1 2; do
    OUT=gpurun_out/$TAG/$v/p$P
    mkdir -p $OUT
    rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o run -- python3 tools/experiments/hex_throughput.py $P 96 quadrilateral > $OUT/log.txt 2>&1
    echo "== $v P$P: $(grep '^P' $OUT/log.txt)"
    python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "hex_stage" in r["Name"]]
    print("   " + "  ".join("%s %.0f us" % (r["Name"].split("hex_stage")[1].split("(")[0], float(r["AverageNs"]) / 1e3)
                            for r in sorted(rows, key=lambda r: r["Name"])))
PY
  done
done
