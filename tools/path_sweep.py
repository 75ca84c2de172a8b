#!/usr/bin/env python
"""Lane-per-cell vs generic stage kernels over block sizes (2-D and 3-D low order): ms per LF4 step.
Chooses the size threshold in api.cpp (use_lane)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seigen_amd import _lib
from seigen_amd.backend import HipBlock


def ms_per_step(path, dim, degree, n, steps=40):
    os.environ["SEIGEN_HIP_PATH"] = path
    blk = HipBlock(dim, degree, n, tuple(1.0 / x for x in n), (0.0,) * dim)
    blk.set_params(1.0, 1e-5, 0.5, 0.25)
    blk.step(5)
    blk.sync()
    t0 = time.perf_counter()
    blk.step(steps)
    blk.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    cells = blk.ncells
    blk.close()
    return dt, cells


if __name__ == "__main__":
    cases = [(2, p, (m, m)) for p in (1, 2, 3, 4) for m in (128, 192, 256, 384, 512)]
    if "--3d" in sys.argv:
        cases = []
    cases += [(3, p, (m, m, m)) for p in (1, 2) for m in (4, 8, 16, 24, 32, 48)]
    for dim, p, n in cases:
        g, cells = ms_per_step("generic", dim, p, n)
        l, _ = ms_per_step("lane", dim, p, n)
        line = "dim %d P%d n=%-4d cells %8d: generic %.4f ms, lane %.4f ms" % (dim, p, n[0], cells, g, l)
        best = "lane" if l < g else "generic"
        if dim == 3:
            m, _ = ms_per_step("mfma", dim, p, n)
            line += ", mfma %.4f ms" % m
            if m < min(g, l):
                best = "mfma"
        print(line + "  -> " + best)
