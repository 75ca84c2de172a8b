"""Generic (thread-per-node, table-driven) against lane (sum-factorised, lane-per-cell) kernels on small hexahedral blocks:
where SG_HEX_LANE_MIN_CELLS (csrc/kernels.hpp) should sit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from seigen_amd import _lib
from seigen_amd.backend import HipBlock


def run(P, N, path, steps=200):
    os.environ["SEIGEN_HIP_PATH"] = path
    blk = HipBlock(3, P, (N, N, N), [1.0 / N] * 3, [0.0] * 3, "quadrilateral")
    blk.set_params(1.0, 0.02 / N / P ** 2, 0.5, 0.25)
    r = np.random.default_rng(0)
    blk.set_field(_lib.FIELD_U, r.uniform(-1, 1, blk.field_shape(_lib.FIELD_U)))
    s = r.uniform(-1, 1, blk.field_shape(_lib.FIELD_S))
    blk.set_field(_lib.FIELD_S, 0.5 * (s + np.swapaxes(s, -1, -2)))
    blk.step(20)
    blk.sync()
    t0 = time.perf_counter()
    blk.step(steps)
    blk.sync()
    dt = (time.perf_counter() - t0) / steps
    blk.close()
    return dt * 1e6


for P in (1, 2):
    for N in (4, 8, 12, 16, 20, 24, 32, 48):
        g, l = run(P, N, "generic"), run(P, N, "lane")
        print("P%d N=%2d cubes %7d  generic %8.1f us/step   lane %8.1f us/step   lane/generic %.2f" % (P, N, N ** 3, g, l, l / g))
        sys.stdout.flush()
