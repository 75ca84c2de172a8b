#!/bin/bash
# ms per stage of config 3 for several SEIGEN_HIP_ORDER_CHUNK values:  tools/bench_order_chunk.sh 0 96 192 ...
for v in "$@"; do
  SEIGEN_HIP_ORDER_CHUNK=$v timeout -k 10 200 python bench.py --no-cpu-baseline --configs none --steps 60 > gpurun_out/bench_var.json 2> gpurun_out/bench_var.err
  python - "$v" <<PY
import json, sys
try:
    d = json.loads(open("gpurun_out/bench_var.json").read().strip().splitlines()[-1])
    print("chunk %-6s %8.0f M DoF/s  %6.3f ms/step  stages %s" % (sys.argv[1], d["value"], d["ms_per_step"], [round(x, 3) for x in d["roofline"]["stage_avg_ms"]]))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/bench_var.err").read()[-300:])
PY
done
