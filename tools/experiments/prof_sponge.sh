#!/bin/bash
# per-kernel durations of the sponge3d probe (tetrahedra P4 at 64^3, hexahedra DQ_3 at 48^3, DQ_2 at 96^3) with a ramp sponge
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/sponge_prof; mkdir -p $out
for c in "tets 64 0" "hex3 48 3" "hex2 96 2"; do
  set -- $c
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1 -o run -- python3 tools/experiments/sponge3d_probe.py $2 $3 ramp > $out/$1.log 2>&1
  echo "== $1"; find $out/$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-160'
done
