#!/bin/bash
# A/B of the factorised G volume (SEIGEN_HIP_GQ = 0 / 1) on one box: parity subset, then config 3
set -o pipefail
out=gpurun_out/ab_gq
mkdir -p $out
SEIGEN_HIP_GQ=1 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_harness_gpu.py -x -q -m gpu \
    -k "apply_F_and_G or full_steps or golden or multiblock or eigenmode_3d or 3d_source" > $out/pytest_gq.txt 2>&1 || { tail -30 $out/pytest_gq.txt; exit 1; }
tail -3 $out/pytest_gq.txt | tee -a $out/log.txt
for rep in 1 2; do
for t in 0 1; do
  SEIGEN_HIP_GQ=$t timeout -k 10 300 python bench.py --steps 60 --no-cpu-baseline --configs none > $out/bench_gq${t}_$rep.json 2>$out/bench_gq${t}_$rep.err || { tail -20 $out/bench_gq${t}_$rep.err; exit 1; }
  python - <<PY | tee -a $out/log.txt
import json
d=json.loads(open("$out/bench_gq${t}_$rep.json").read().strip().splitlines()[-1])
print("GQ=$t ms/step %.4f value %.0f stages" % (d["ms_per_step"], d["value"]), [round(x, 4) for x in d["roofline"]["stage_avg_ms"]])
PY
done
done
