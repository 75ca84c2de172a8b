#!/bin/bash
# the native exchange on one device (a rank that is its own z+- neighbour) against RCCL's point-to-point channel count:
# every channel is a workgroup that competes with the persistent stage kernels for a CU
for o in 0 1; do
for ch in default 1 2 4; do
  if [ "$ch" = default ]; then unset NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS; else export NCCL_MAX_P2P_NCHANNELS=$ch NCCL_MIN_P2P_NCHANNELS=$ch; fi
  echo "== ordered=$o p2p channels=$ch"
  SEIGEN_HALO_ORDERED=$o SEIGEN_BENCH_GRIDS=2 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$o tools/bench_rccl_self.py 2>&1 | grep -E "NATIVE|single block"
done
done
