#!/bin/bash
# the round's record runs on one box: default bench line, other degrees / FP32, one rank's share of config 4,
# the 2-D configs, the reference's harness runs end to end
out=gpurun_out/r04_numbers
mkdir -p $out
timeout -k 10 600 python bench.py > $out/config3_bench.json 2> $out/config3_bench.err; tail -c 600 $out/config3_bench.json | head -c 300; echo
: > $out/other_degrees_and_fp32_bench.jsonl
for a in "--degree 3" "--degree 2" "--degree 1" "--dtype f32" "--dtype f32 --degree 3"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 60 $a >> $out/other_degrees_and_fp32_bench.jsonl 2>> $out/other.err
done
timeout -k 10 600 python bench.py --no-cpu-baseline --workload c4 --steps 10 --warmup 2 > $out/bench_c4_share.json 2>> $out/other.err
timeout -k 10 600 python tools/bench_configs.py c2 c5 c1 --steps 200 > $out/secondary_configs_bench.jsonl 2>> $out/other.err
timeout -k 10 300 python tools/experiments/end_to_end_harness.py > $out/end_to_end_harness.txt 2>> $out/other.err
python - <<PY
import json
for f in ("other_degrees_and_fp32_bench.jsonl", "bench_c4_share.json", "secondary_configs_bench.jsonl"):
    for ln in open("$out/" + f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f, d.get("config", {}).get("workload", d.get("config"))if isinstance(d.get("config"), dict) else d.get("config"), round(d.get("value", 0)), d.get("ms_per_step"))
PY
cat $out/end_to_end_harness.txt
