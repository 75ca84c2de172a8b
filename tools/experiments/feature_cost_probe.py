#!/usr/bin/env python
"""What the optional features cost on config 3's block (64^3 cubes x 6 tets, P4, FP64; device time per step and stage):
per-cell lambda / mu, per-cell density (physical rule), a stress that is not symmetric (the 9-line kernels), a sparse source."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from seigen_amd import _lib
from seigen_amd.backend import HipBlock


def run(label, n=64, P=4, per_cell=False, density=False, asym=False, source=0, steps=20):
    blk = HipBlock(3, P, (n, n, n), (1.0 / n,) * 3, (0.0,) * 3)
    r = np.random.default_rng(3)
    dt = 0.5 / n / 8
    if per_cell:
        blk.set_params(1.0, dt, r.uniform(0.4, 0.6, blk.ncells), r.uniform(0.2, 0.3, blk.ncells))
    else:
        blk.set_params(1.0, dt, 0.5, 0.25)
    if density:
        blk.set_density(r.uniform(0.8, 1.2, blk.ncells), physical=True)
    layer = n * n * 6
    u = r.uniform(-1, 1, size=(layer * 8,) + blk.field_shape(_lib.FIELD_U)[1:])
    s = r.uniform(-1, 1, size=(layer * 8,) + blk.field_shape(_lib.FIELD_S)[1:])
    if not asym:
        s = 0.5 * (s + np.swapaxes(s, -1, -2))
    for k in range(0, n, 8):
        blk.set_field_range(_lib.FIELD_U, k * layer, u)
        blk.set_field_range(_lib.FIELD_S, k * layer, s)
    if source:
        nodes = np.unique(r.integers(0, blk.ncells * blk.nd, size=source))
        sv = r.uniform(-1, 1, size=(len(nodes), 3, 3))
        blk.set_source_separable(nodes, 0.5 * (sv + np.swapaxes(sv, -1, -2)), np.ones(3 * steps + 10))
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
    print("%-34s %6.1f G  %7.3f ms/step  %s  sym=%s" % (label, dofs / ms / 1e6, ms, st, blk.is_sym()), flush=True)
    blk.close()


if __name__ == "__main__":
    run("plain")
    run("per-cell lambda, mu", per_cell=True)
    run("per-cell density (physical)", density=True)
    run("sparse source, 20 000 nodes", source=20000)
    run("non-symmetric stress", asym=True)
