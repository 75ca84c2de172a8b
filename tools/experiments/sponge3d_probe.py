#!/usr/bin/env python
"""3-D explosive-source-like block (config 4's share set-up: cubes x 6 tets, P4, box-Ricker source) WITH the sponge of the
reference's 2-D script carried over to 3-D - sigma = 1000 in strips 8 cells wide on five faces, none on the free surface
(explosive_source_lf4.py:42-45) - against the same block without it: device time per step and stage."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import seigen_amd
from seigen_amd import Expression, Function, FunctionSpace
from seigen_amd.harness import baseline_configs as bc

seigen_amd.elastic.log = lambda s: None


def run(n, sponge, steps=20, hexa=0, ramp=False):
    h = 2.5
    L = n * h
    if sponge or hexa:
        # config4_share builds and sets up; the sponge has to be in place before setup(): rebuild by hand
        from seigen_amd import BoxMesh, ElasticLF4, Vp, cfl_dt
        mesh = BoxMesh(n, n, n, L, L, L, hexahedral=bool(hexa))
        el = ElasticLF4.create(mesh, "DQ" if hexa else "DG", hexa or 4, dimension=3, solver="explicit", output=False)
        el.density, el.mu, el.l = 1.0, 3600.0, 3599.3664
        el.dt = cfl_dt(h, Vp(el.mu, el.l, el.density), 0.05) / 8
        if sponge:
            el.absorption_function = Function(FunctionSpace(mesh, "DQ" if hexa else "DG", 4))
            w = 8 * h
            if ramp:      # sigma varies inside every sponge cell: a linear ramp over the strips
                el.absorption = Expression("fmax(fmax(fmax(%r - x[0], x[0] - %r), fmax(%r - x[1], x[1] - %r)), fmax(%r - x[2], 0.0)) * 50.0"
                                           % (w, L - w, w, L - w, w))
            else:
                el.absorption = Expression("x[0] <= %r || x[0] >= %r || x[1] <= %r || x[1] >= %r || x[2] <= %r ? 1000 : 0"
                                           % (w, L - w, w, L - w, w))
        el.setup()
        el.block.set_source([], None)
        rng = np.random.default_rng(0)
    else:
        el, _ = bc.config4_share(2 * steps + 10, n=n)
    blk = el.block
    blk.set_field(0, np.random.default_rng(1).uniform(-1, 1, blk.field_shape(0)))
    blk.step(3)
    blk.sync()
    blk.step(steps)
    blk.sync()
    ms = blk.last_step_ms() / steps
    blk.enable_timing(True)
    c0 = blk.counters()
    blk.step(steps)
    blk.sync()
    c1 = blk.counters()
    st = [round((c1["kernel_ms"][i] - c0["kernel_ms"][i]) / steps, 3) for i in range(6)]
    dofs = blk.u_dofs + blk.s_dofs
    print("%s n %d sponge %s: %.1f G DoF-updates/s, %.3f ms/step, stages %s" % ("hexahedra DQ_%d" % hexa if hexa else "tetrahedra P4", n, ("ramp" if ramp else sponge), dofs / ms / 1e6, ms, st), flush=True)
    blk.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    hexa = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 1 .. 4: hexahedra of that degree
    run(n, False, hexa=hexa)
    run(n, True, hexa=hexa)
    if len(sys.argv) > 3:
        run(n, True, hexa=hexa, ramp=True)
