#!/usr/bin/env python
"""Config 2 (2-D explosive source, 512^2, P2) with a sponge whose sigma varies inside every sponge cell (a linear ramp over the
20 m strips instead of the reference's constant 1000): every sponge cell then has a matrix of its own kind."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import seigen_amd
from seigen_amd import Expression
from seigen_amd.harness.explosive_source import ExplosiveSourceLF4
import seigen_amd.harness.explosive_source as _hx

seigen_amd.elastic.log = _hx.log = lambda s: None


def run(label, ramp, degree=2, n=512, steps=300):
    h = 2.5
    es = ExplosiveSourceLF4()
    el = es.setup(Lx=n * h, Ly=n * h, h=h, degree=degree, courant_number=0.05)
    if ramp:
        Lx = n * h
        el.absorption = Expression("x[0] <= 20 ? 50*(20 - x[0]) : (x[0] >= %r ? 50*(x[0] - %r) : (x[1] <= 20 ? 50*(20 - x[1]) : 0))"
                                   % (Lx - 20.0, Lx - 20.0))
    el.setup()
    el.upload_source([el.dt * (k + 1) for k in range(2 * steps + 10)])
    blk = el.block
    blk.step(5)
    blk.sync()
    blk.step(steps)
    blk.sync()
    ms = blk.last_step_ms() / steps
    blk.enable_timing(True)
    c0 = blk.counters()
    blk.step(steps)
    blk.sync()
    c1 = blk.counters()
    st = [round((c1["kernel_ms"][i] - c0["kernel_ms"][i]) / steps * 1e3, 1) for i in range(6)]
    dofs = blk.u_dofs + blk.s_dofs
    print("%-28s P%d %6.1f G  %7.1f us/step  %s" % (label, degree, dofs / ms / 1e6, ms * 1e3, st), flush=True)
    blk.close()


if __name__ == "__main__":
    for degree in (2, 4):
        run("constant strips (reference)", False, degree)
        run("linear ramp", True, degree)
