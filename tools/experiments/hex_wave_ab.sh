#!/bin/bash
# Cube-per-wave kernel (kernels.hip hex_wave_stage) against the thread-per-node generic kernel on hexahedra of degree 3 and 4:
# per-kernel rocprofv3 averages.   bash tools/experiments/hex_wave_ab.sh
export TMPDIR=/tmp
for PN in "3 48" "4 40"; do
  for hw in 1 0; do
    export SEIGEN_HIP_HEXWAVE=$hw
    out=gpurun_out/hexwave_ab/p${PN% *}_hw$hw
    mkdir -p $out
    rocprofv3 --output-format csv --kernel-trace --stats -d $out -o run -- python3 tools/experiments/hex_throughput.py $PN quadrilateral > $out/log.txt 2>&1
    echo "HEXWAVE=$hw $(grep '^P' $out/log.txt)"
    python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: r["Name"]):
        if "stage" in r["Name"]:
            print("    %-50s %4s launches  %6.0f us" % (r["Name"].replace("void sg::", "").replace("(sg::StageArgs)", ""), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
