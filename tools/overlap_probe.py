#!/usr/bin/env python
"""Does a kernel on another stream run WHILE a persistent stage kernel is in flight?

Launches one config-3 stage (1.3 ms, every usable CU filled by the persistent grid) and, right
behind it on a second stream, a small torch kernel (stand-in for RCCL's send/receive kernel).
Prints when the small kernel finished relative to the stage, for persistent grids that fill all
block slots of the device (512) or leave some empty (SEIGEN_HIP_GRID_BLOCKS; a block with halo
neighbours uses 480, api.cpp)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from seigen_amd import _lib
from seigen_amd.backend import HipBlock


def probe(grid_blocks, n=48):
    os.environ["SEIGEN_HIP_GRID_BLOCKS"] = str(grid_blocks)
    blk = HipBlock(3, 4, (n, n, n), (1.0 / n,) * 3, (0.0,) * 3)
    blk.set_params(1.0, 1e-4, 0.5, 0.25)
    main = torch.cuda.ExternalStream(blk.stream_ptr())
    side = torch.cuda.Stream()
    x = torch.ones(1 << 20, device="cuda")
    res = []
    for rep in range(5):
        blk.sync()
        torch.cuda.synchronize()
        e0, e1, es = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(main)
        blk.run_stage(_lib.STAGE_STEMP, _lib.REGION_ALL)
        with torch.cuda.stream(side):
            y = x * 2.0
            es.record(side)
        e1.record(main)
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1), e0.elapsed_time(es)))
    blk.close()
    return res


if __name__ == "__main__":
    for grid_blocks in (512, 496, 480, 448):
        r = probe(grid_blocks)
        print("persistent grid %d: stage ms / small-kernel-done ms after stage start: %s"
              % (grid_blocks, ", ".join("%.3f/%.3f" % t for t in r[1:])))
