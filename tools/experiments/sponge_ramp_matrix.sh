#!/bin/bash
# ramp sponges, every 3-D family: the affine path against the same cells through their matrices (SEIGEN_HIP_SPONGE_AFFINE=0)
for c in "64 0" "48 3" "40 4" "96 2" "96 1"; do
  set -- $c
  for aff in default 0 1; do
    if [ $aff = default ]; then unset SEIGEN_HIP_SPONGE_AFFINE; else export SEIGEN_HIP_SPONGE_AFFINE=$aff; fi
    echo "SEIGEN_HIP_SPONGE_AFFINE=$aff: $(python3 tools/experiments/sponge3d_probe.py $1 $2 ramp 2>&1 | tail -1)"
  done
done
