#!/bin/bash
export TMPDIR=/tmp
for n in "64 64 64" "4 256 256" "8 128 256" "16 64 256"; do
  tag=$(echo $n | tr ' ' x)
  out=gpurun_out/group_shape/$tag
  mkdir -p $out
  rocprofv3 --output-format csv --kernel-trace --stats -d $out/t -o run -- python3 tools/experiments/group_shape_probe.py $n > $out/t.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/f -o run -- python3 tools/experiments/group_shape_probe.py $n > $out/f.log 2>&1
  echo "== $(grep '^n =' $out/t.log)"
  python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mfma_stage" in r["Name"]: dur[r["Name"]] = float(r["AverageNs"]) / 1e3
acc = collections.defaultdict(list)
for f in glob.glob(out + "/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k in sorted(dur):
    if k in acc:
        print("   %-46s %7.1f us   fetch %6.2f GB" % (k.replace("void sg::", "").replace("(sg::StageArgs)", ""), dur[k], 2 * sum(acc[k]) / len(acc[k]) * 1024 / 1e9))
PY
done
