#!/usr/bin/env python
"""What the multi-GPU schedule costs with the REAL transport, on one GPU: a 64^3 P4 block whose z- and
z+ neighbour is the rank itself (RCCL send/receive to self: same calls, same kernels, same stream
choreography as between GPUs; the bytes cross HBM instead of xGMI).  Compares the pipelined exchange
(FIRST, pack, RCCL || SECOND) with the plain single-block step.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/bench_rccl_self.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    from seigen_amd.mesh import Partition
    from seigen_amd.parallel import HaloExchanger
    torch.cuda.set_device(0)
    # started without torchrun (e.g. under rocprofv3, which must not see a launcher): a one-rank rendezvous of our own
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))

    class SelfNeighbour(Partition):
        def neighbour(self, side):
            return 0 if side >> 1 == 2 else None

    n, P, steps = (64, 64, 64), 4, int(os.environ.get("SEIGEN_BENCH_STEPS", "60"))
    h = [1.0 / 64] * 3
    rng = np.random.default_rng(0)
    grids = os.environ.get("SEIGEN_BENCH_GRID_LIST")
    for grid_env in ((None, "512", "496")[:int(os.environ.get("SEIGEN_BENCH_GRIDS", "3"))] if not grids else tuple(grids.split(","))):
        if grid_env:
            os.environ["SEIGEN_HIP_GRID_BLOCKS"] = grid_env
        part = SelfNeighbour(n, 0, 1)
        blk = HipBlock(3, P, n, h, [0.0] * 3, "left", part.nbr_mask)
        blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
        layer = 64 * 64 * 6
        u = rng.uniform(-1, 1, (layer,) + blk.field_shape(_lib.FIELD_U)[1:]) * 1e-3
        for k in range(64):
            blk.set_field_range(_lib.FIELD_U, k * layer, u)
        stream = torch.cuda.ExternalStream(blk.stream_ptr(), device=0)
        ex = HaloExchanger(blk, part, torch.device("cuda", 0), stream=stream)
        ex.step(3)
        blk.sync()
        torch.cuda.synchronize()
        ex.reset_stats(timing=True)
        t0 = time.perf_counter()
        ex.step(steps)
        blk.sync()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        st = ex.stats()
        print("z+- neighbours over RCCL (self), persistent grid %s blocks while exchanging: %.3f ms/step, stream waited %.3f ms/step "
              "for traces, %.1f MB sent per step" % (grid_env or "480 (default)", dt, st["exposed_wait_ms"] / steps,
                                                    st["bytes_sent"] / steps / 1e6), flush=True)
        blk.close()
    # the same schedule driven from inside the library (csrc/comm.cpp): one sg_step call for all the steps
    from seigen_amd.backend import comm_unique_id
    for grid_env in ((None, "512", "496")[:int(os.environ.get("SEIGEN_BENCH_GRIDS", "3"))] if not grids else tuple(grids.split(","))):
        if grid_env:
            os.environ["SEIGEN_HIP_GRID_BLOCKS"] = grid_env
        else:
            os.environ.pop("SEIGEN_HIP_GRID_BLOCKS", None)
        part = SelfNeighbour(n, 0, 1)
        blk = HipBlock(3, P, n, h, [0.0] * 3, "left", part.nbr_mask)
        blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
        layer = 64 * 64 * 6
        u = rng.uniform(-1, 1, (layer,) + blk.field_shape(_lib.FIELD_U)[1:]) * 1e-3
        for k in range(64):
            blk.set_field_range(_lib.FIELD_U, k * layer, u)
        blk.comm_init(comm_unique_id(), 0, 1, [None, None, None, None, 0, 0])
        blk.step(3)
        blk.sync()
        blk.comm_stats(reset=True)
        blk.enable_timing(True)
        t0 = time.perf_counter()
        blk.step(steps)
        blk.sync()
        dt = (time.perf_counter() - t0) / steps * 1e3
        st = blk.comm_stats()
        blk.enable_timing(False)
        print("NATIVE exchange (csrc/comm.cpp), z+- neighbours over RCCL (self), persistent grid %s blocks while exchanging: %.3f ms/step, "
              "receives lasted %.3f ms/step beyond SECOND, %.1f MB sent per step"
              % (grid_env or "480 (default)", dt, st["exposed_wait_ms"] / steps, st["bytes_sent"] / steps / 1e6), flush=True)
        blk.close()
    os.environ.pop("SEIGEN_HIP_GRID_BLOCKS", None)
    blk = HipBlock(3, P, n, h, [0.0] * 3)
    blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
    # the SAME data as the blocks above: on all-zero fields the matrix pipe draws less power, the package clocks higher
    # and the step takes 5-8 % less - rounds 2 and 3 compared the exchanging blocks with such a block and overstated
    # the cost of the exchange accordingly (profiles/r04/neighbour_overhead.txt)
    layer = 64 * 64 * 6
    u = rng.uniform(-1, 1, (layer,) + blk.field_shape(_lib.FIELD_U)[1:]) * 1e-3
    for k in range(64):
        blk.set_field_range(_lib.FIELD_U, k * layer, u)
    blk.step(3)
    blk.sync()
    t0 = time.perf_counter()
    blk.step(steps)
    blk.sync()
    print("single block, no neighbours: %.3f ms/step" % ((time.perf_counter() - t0) / steps * 1e3))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
