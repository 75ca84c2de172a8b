#!/bin/bash
# GQ on / off at the other sizes: degree 3, and one rank's share of config 4
for rep in 1 2; do
for t in 0 1; do
  for args in "--degree 3" "--workload c4 --steps 10 --warmup 2"; do
    SEIGEN_HIP_GQ=$t timeout -k 10 400 python bench.py --steps 40 --no-cpu-baseline --configs none $args > gpurun_out/gq2.json 2> gpurun_out/gq2.err || { tail -5 gpurun_out/gq2.err; continue; }
    python - "$t" "$args" <<PY
import json, sys
d=json.loads(open("gpurun_out/gq2.json").read().strip().splitlines()[-1])
print("GQ=%s %-40s ms/step %.4f value %.0f stages" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"]), [round(x, 3) for x in d["roofline"]["stage_avg_ms"]], flush=True)
PY
  done
done
done
