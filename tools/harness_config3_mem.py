#!/usr/bin/env python
"""The reference-style harness at BASELINE config 3 (Eigenmode3DLF4(64, 4, dt): Expression
interpolation of the initial fields through the solver class, eigenmode_3d.py:30-40), 20 steps:
wall time of set-up + run and the peak host memory of the process."""
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.eigenmode as he
    helpers.log = seigen_amd.elastic.log = he.log = lambda s: None
    N, P = 64, 4
    dt = 0.5 * (1.0 / N) / 2 ** (P - 1)
    t0 = time.perf_counter()
    em = he.Eigenmode3DLF4(N, P, dt, output=False)
    u1, s1 = em.eigenmode3d(T=20 * dt * (1 + 1e-9))
    em.elastic.block.sync()
    t1 = time.perf_counter()
    peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    print("Eigenmode3DLF4(64, 4): set-up + initial conditions + 20 steps in %.1f s, peak host memory %.2f GB" % (t1 - t0, peak))
