#!/bin/bash
# Fabric bytes per launch (FETCH_SIZE x 2, WRITE_SIZE; separate passes) and launch times of configs 2 and 5:
#   bash tools/experiments/fabric_view_2d.sh <tag>
set -u
TAG=${1:-fabric2d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in c2 c5; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${c}_trace -o run -- python3 tools/bench_configs.py $c --steps 100 > $OUT/${c}_trace.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/${c}_fetch -o run -- python3 tools/bench_configs.py $c --steps 20 > $OUT/${c}_fetch.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/${c}_write -o run -- python3 tools/bench_configs.py $c --steps 20 > $OUT/${c}_write.log 2>&1
  echo "== $c: $(grep -h '^{' $OUT/${c}_trace.log | cut -c1-200)"
  python3 - $OUT $c <<'PY'
import csv, glob, sys, collections
out, c = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob("%s/%s_trace/**/*kernel_stats.csv" % (out, c), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stage" in r["Name"]: dur[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
cnt = collections.defaultdict(dict)
for which in ("fetch", "write"):
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, c, which), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            cnt[k][which] = sum(v) / len(v)
for k in sorted(dur):
    if k in cnt and "fetch" in cnt[k]:
        fb, wb = 2 * cnt[k]["fetch"] * 1024, cnt[k].get("write", 0) * 1024
        us = dur[k][0]
        print("   %-62s %7.1f us  fetch %7.1f MB  write %7.1f MB  -> %5.2f TB/s" % (k.replace("void sg::", "").replace("(sg::StageArgs)", "")[:62], us, fb / 1e6, wb / 1e6, (fb + wb) / us / 1e6))
PY
done
