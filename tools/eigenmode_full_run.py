#!/usr/bin/env python
"""Config 3 to the reference's end time: the 3-D eigenmode of tests/eigenmode/eigenmode_3d.py on
64^3 cubes x 6 tets, P4, dt = 0.5/64/8, T = 5 (5120 steps), then the nodal error against the
analytic solution sampled on whole z-layers of cubes."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import seigen_amd
from seigen_amd import ElasticLF4, BoxMesh, _lib
from seigen_amd.functionspace import block_config
import seigen_amd.helpers as helpers

if __name__ == "__main__":
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None
    n, P, T = 64, 4, float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    mesh = BoxMesh(n, n, n, 1.0, 1.0, 1.0)
    el = ElasticLF4.create(mesh, "DG", P, dimension=3, solver="explicit", output=False)
    el.density, el.mu, el.l = 1.0, 0.25, 0.5
    el.dt = dt = 0.5 * (1.0 / n) / 2 ** (P - 1)
    bench.fill_initial_condition(el, dt)
    el.setup()
    blk = el.block
    blk.set_source([], None)
    nsteps = len(el.step_times(T))
    t0 = time.perf_counter()
    blk.step(nsteps)
    blk.sync()
    wall = time.perf_counter() - t0
    lib = _lib.load()
    layer = n * n * 6
    eu = es = 0.0
    for k in (0, 9, 23, 32, 47, 63):
        cfg = block_config(mesh, P)
        cfg.n[2] = 1
        cfg.origin[2] = k * mesh.h[2]
        X = np.empty((layer, blk.nd, 3))
        _lib.check(lib.sg_block_node_coords(C.byref(cfg), P, X.ctypes.data, X.nbytes))
        ue, se = bench.eigenmode3d_fields(X, nsteps * dt, nsteps * dt + dt / 2)
        eu = max(eu, np.abs(blk.get_field_range(_lib.FIELD_U, k * layer, layer) - ue).max())
        es = max(es, np.abs(blk.get_field_range(_lib.FIELD_S, k * layer, layer) - se).max())
    dofs = blk.u_dofs + blk.s_dofs
    print("T = %g: %d steps in %.1f s (%.1f G DoF-updates/s); max nodal error u %.3e, s %.3e (fields are O(1))"
          % (T, nsteps, wall, dofs * nsteps / wall / 1e9, eu, es))
