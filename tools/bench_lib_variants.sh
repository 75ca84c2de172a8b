#!/bin/bash
# ms per stage of config 3 for library variants built with tools/build_variant.sh:  tools/bench_lib_variants.sh name...
# (STEPS=60 by default; extra environment, e.g. SEIGEN_HIP_ORDER_CHUNK, is passed through)
for v in "$@"; do
  lib=""; [ "$v" != "default" ] && lib="$PWD/build_tools/libseigen_hip_$v.so"
  SEIGEN_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --configs none --steps ${STEPS:-60} > gpurun_out/bench_var.json 2> gpurun_out/bench_var.err
  python - "$v" <<PY
import json, sys
try:
    d = json.loads(open("gpurun_out/bench_var.json").read().strip().splitlines()[-1])
    print("%-10s %8.0f M DoF/s  %6.3f ms/step  stages %s" % (sys.argv[1], d["value"], d["ms_per_step"], [round(x, 3) for x in d["roofline"]["stage_avg_ms"]]), flush=True)
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/bench_var.err").read()[-300:])
PY
done
