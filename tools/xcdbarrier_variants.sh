#!/bin/bash
# tools/ubench_xcdbarrier.hip with the three scopes of the L1 invalidate that follows the barrier
for inv in "buffer_inv sc0" "buffer_inv sc1" "buffer_inv sc0 sc1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DINV="\"$inv\"" -o /tmp/xb tools/ubench_xcdbarrier.hip 2>/dev/null || { echo "$inv: does not assemble"; continue; }
  echo "== $inv"
  for nb in 16 32 50 64; do timeout -k 5 60 /tmp/xb $nb 8192 | tail -1; done
done
echo "== agent-scope loads (sc1), no invalidate"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DAGENT_LOADS -DINV="\"s_nop 0\"" -o /tmp/xb tools/ubench_xcdbarrier.hip 2>/dev/null
for nb in 16 32 50 64; do timeout -k 5 60 /tmp/xb $nb 8192 | tail -1; done
for nb in 50; do timeout -k 5 60 /tmp/xb $nb 65536 | tail -2; done
