for deg in 1 2; do
  for rep in 1 2; do
  for v in base low5 low6 low8; do
    lib=""; [ $v != base ] && lib=$PWD/build_tools/libseigen_hip_$v.so
    r=$(SEIGEN_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --degree $deg --steps 300 --no-cpu-baseline --configs none 2>/dev/null | python3 -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(r['value']), round(r['ms_per_step'],4), [round(x,4) for x in r['roofline']['stage_avg_ms']])")
    echo "P$deg $v rep$rep: $r"
  done; done
done
