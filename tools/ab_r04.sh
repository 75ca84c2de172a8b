#!/bin/bash
# A/B on ONE box: the round-4 tree (a copy of commit 8b10c97 built under build_tools/r04, see profiles/r05/README.md)
# against this tree, BASELINE config 3, alternating runs - the check behind "kernels_mfma.hip consolidated, step unchanged".
out=gpurun_out/ab_r04; mkdir -p $out
for rep in 1 2 3; do
  (cd build_tools/r04 && timeout -k 10 200 python bench.py --steps ${STEPS:-150} --no-cpu-baseline) > $out/old_$rep.json 2> $out/old_$rep.err || { tail -5 $out/old_$rep.err; exit 1; }
  timeout -k 10 200 python bench.py --steps ${STEPS:-150} --no-cpu-baseline --configs none > $out/new_$rep.json 2> $out/new_$rep.err || { tail -5 $out/new_$rep.err; exit 1; }
done
python - <<PY
import json, glob
for tag in ("old", "new"):
    rows = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("$out/%s_*.json" % tag))]
    print(tag, "ms/step", ["%.4f" % r["ms_per_step"] for r in rows], "stages", [[round(x, 3) for x in r["roofline"]["stage_avg_ms"]] for r in rows][-1])
PY
