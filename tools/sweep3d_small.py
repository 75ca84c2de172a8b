#!/usr/bin/env python
"""ms per LF4 step of small 3-D blocks (the reference's own 3-D sweep sizes, eigenmode_3d.py:72-88: N <= 8, and 16)
on the production path, P1..P4."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seigen_amd.backend import HipBlock


def ms_per_step(degree, n, steps=200):
    blk = HipBlock(3, degree, (n, n, n), (1.0 / n,) * 3, (0.0,) * 3)
    blk.set_params(1.0, 1e-5, 0.5, 0.25)
    blk.step(20)
    blk.sync()
    t0 = time.perf_counter()
    blk.step(steps)
    blk.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    blk.close()
    return dt


if __name__ == "__main__":
    for p in (1, 2, 3, 4):
        print("P%d " % p + "  ".join("N=%-2d %.4f ms" % (n, ms_per_step(p, n)) for n in (2, 4, 8, 16)))
