#!/usr/bin/env python
"""Host <-> device field transfer through the C-ABI (sg_set_field / sg_get_field: pageable host
memory -> staging buffer -> layout kernel) on config 3, and what it does to a whole job:
upload u0, s0; N steps; download u1, s1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seigen_amd import _lib
from seigen_amd.backend import HipBlock

if __name__ == "__main__":
    n, P = 64, 4
    blk = HipBlock(3, P, (n, n, n), (1.0 / n,) * 3, (0.0,) * 3)
    blk.set_params(1.0, 0.5 / n / 8, 0.5, 0.25)
    layer = n * n * 6
    rng = np.random.default_rng(0)
    u = rng.uniform(-1, 1, size=(layer * 8,) + blk.field_shape(_lib.FIELD_U)[1:])
    s = rng.uniform(-1, 1, size=(layer * 8,) + blk.field_shape(_lib.FIELD_S)[1:])
    s = 0.5 * (s + np.swapaxes(s, -1, -2))
    t0 = time.perf_counter()
    for k in range(0, n, 8):
        blk.set_field_range(_lib.FIELD_U, k * layer, u)
        blk.set_field_range(_lib.FIELD_S, k * layer, s)
    blk.sync()
    t_up = time.perf_counter() - t0
    nbytes = (u.nbytes + s.nbytes) * (n // 8)
    blk.step(2)
    blk.sync()
    t0 = time.perf_counter()
    blk.step(20)
    blk.sync()
    t_step = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for k in range(0, n, 8):
        blk.get_field_range(_lib.FIELD_U, k * layer, layer * 8)
        blk.get_field_range(_lib.FIELD_S, k * layer, layer * 8)
    t_dn = time.perf_counter() - t0
    # the same into arrays that exist already (a fresh numpy array costs its page faults on first touch)
    t0 = time.perf_counter()
    for k in range(0, n, 8):
        blk.get_field_range(_lib.FIELD_U, k * layer, layer * 8, out=u)
        blk.get_field_range(_lib.FIELD_S, k * layer, layer * 8, out=s)
    t_dn2 = time.perf_counter() - t0
    print("download into existing arrays: %.2f s = %.1f GB/s" % (t_dn2, nbytes / 1e9 / t_dn2))
    dofs = blk.u_dofs + blk.s_dofs
    print("upload %.2f GB in %.2f s = %.1f GB/s; download %.2f s = %.1f GB/s; step %.2f ms" %
          (nbytes / 1e9, t_up, nbytes / 1e9 / t_up, t_dn, nbytes / 1e9 / t_dn, t_step * 1e3))
    for nsteps in (100, 640, 5120):
        total = t_up + t_dn + nsteps * t_step
        print("job of %4d steps incl. upload and download: %.1f G DoF-updates/s (resident: %.1f)" %
              (nsteps, dofs * nsteps / total / 1e9, dofs / t_step / 1e9))
