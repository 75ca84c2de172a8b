"""Throughput of the hexahedral path (DQ_1, DQ_2 on the table-driven generic kernel) beside tetrahedra of the same
degree on the same cubes, in bench.py's unit: DoF-updates = (U dofs + S dofs) * steps, 64 algorithmic bytes each
(SURVEY 8d)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from seigen_amd import _lib
from seigen_amd.backend import HipBlock


def run(P, N, diagonal, steps=20):
    h = [1.0 / N] * 3
    blk = HipBlock(3, P, (N, N, N), h, [0.0] * 3, diagonal)
    blk.set_params(1.0, 0.02 / N / P ** 2, 0.5, 0.25)
    r = np.random.default_rng(0)
    blk.set_field(_lib.FIELD_U, r.uniform(-1, 1, blk.field_shape(_lib.FIELD_U)))
    s = r.uniform(-1, 1, blk.field_shape(_lib.FIELD_S))
    blk.set_field(_lib.FIELD_S, 0.5 * (s + np.swapaxes(s, -1, -2)))
    blk.step(3)
    blk.sync()
    t0 = time.perf_counter()
    blk.step(steps)
    blk.sync()
    dt = (time.perf_counter() - t0) / steps
    upd = blk.ncells * blk.nd * 12.0          # 3 velocity + 9 stress values per node
    out = (P, N, diagonal, blk.ncells, blk.nd, dt * 1e3, upd / dt / 1e9, 64.0 * upd / dt / 1e12)
    blk.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) >= 4:          # one case: P N diagonal
        print("P%d N=%d %-13s cells %9d nd %2d  %8.3f ms/step  %6.2f G DoF-updates/s  (%.2f TB/s algorithmic)"
              % run(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]))
        sys.exit(0)
    for P, N in ((3, 48), (4, 40)):          # DQ_3, DQ_4: the sum-factorised generic kernel (no lane kernel)
        print("P%d N=%d %-13s cells %9d nd %2d  %8.3f ms/step  %6.2f G DoF-updates/s  (%.2f TB/s algorithmic)" % run(P, N, "quadrilateral"))
        sys.stdout.flush()
    for P, N in ((1, 96), (2, 64), (2, 96)):
        for diagonal in ("quadrilateral", "left"):
            print("P%d N=%d %-13s cells %9d nd %2d  %8.3f ms/step  %6.2f G DoF-updates/s  (%.2f TB/s algorithmic)" % run(P, N, diagonal))
            sys.stdout.flush()
