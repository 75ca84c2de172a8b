"""Repeat the 27-block P3 multi-block case with and without SEIGEN_HIP_GQ and report where results differ from the single
block (one process; a wrong-result flake seen once in the full suite, not a GPU fault)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from seigen_amd import _lib
from seigen_amd.backend import HipBlock
from seigen_amd.mesh import Partition
from tests.test_harness_gpu import _LocalExchange
from tests.util import seeded

def run(gq, trials, degree=3, n=(9, 9, 9), grid=(3, 3, 3)):
    os.environ["SEIGEN_HIP_GQ"] = str(gq)
    dim = 3
    h = [1.0 / n[a] for a in range(dim)]
    dt = 0.02 * min(h) / degree ** 2
    bad = 0
    for t in range(trials):
        single = HipBlock(dim, degree, n, h, [0.0] * dim, "left")
        u0 = seeded(single.field_shape(_lib.FIELD_U), 11)
        s0 = seeded(single.field_shape(_lib.FIELD_S), 12)
        single.set_params(1.0, dt, 0.5, 0.25)
        single.set_field(_lib.FIELD_U, u0); single.set_field(_lib.FIELD_S, s0)
        single.step(3)
        uref, sref = single.get_field(_lib.FIELD_U), single.get_field(_lib.FIELD_S)
        # the single-block run itself, twice: deterministic?
        single2 = HipBlock(dim, degree, n, h, [0.0] * dim, "left")
        single2.set_params(1.0, dt, 0.5, 0.25)
        single2.set_field(_lib.FIELD_U, u0); single2.set_field(_lib.FIELD_S, s0)
        single2.step(3)
        if not np.array_equal(single2.get_field(_lib.FIELD_U), uref):
            print("gq", gq, "trial", t, "SINGLE-BLOCK runs differ", flush=True)
        world = int(np.prod(grid))
        parts = [Partition(n, r, world, grid) for r in range(world)]
        def cells_of(p):
            ax = [np.arange(p.start[a], p.start[a] + p.n[a]) for a in range(3)]
            cube = (ax[0][None, None, :] + n[0] * (ax[1][None, :, None] + n[1] * ax[2][:, None, None])).reshape(-1)
            return (cube[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)
        blocks = []
        for p in parts:
            b = HipBlock(dim, degree, p.n, h, [p.start[a] * h[a] for a in range(dim)], "left", p.nbr_mask)
            sel = cells_of(p)
            b.set_params(1.0, dt, 0.5, 0.25)
            b.set_field(_lib.FIELD_U, u0[sel]); b.set_field(_lib.FIELD_S, s0[sel])
            blocks.append(b)
        ex = _LocalExchange(blocks, parts)
        ex.step(3, True)
        for r, (b, p) in enumerate(zip(blocks, parts)):
            sel = cells_of(p)
            for name, f, ref in (("u", _lib.FIELD_U, uref), ("s", _lib.FIELD_S, sref)):
                got = b.get_field(f)
                if not np.array_equal(got, ref[sel]):
                    d = np.argwhere(got != ref[sel])
                    cells = np.unique(d[:, 0])
                    bad += 1
                    print("gq", gq, "trial", t, "block", r, p.start, "field", name, "cells differing", len(cells), "of", len(sel),
                          "first", cells[:8], "cubes", np.unique(cells // 6)[:12], "max diff", np.abs(got - ref[sel]).max(), flush=True)
        for b in blocks:
            b.close()
        single.close(); single2.close()
    print("gq", gq, ":", bad, "mismatching block-fields in", trials, "trials", flush=True)

for gq in (1, 0):
    run(gq, int(sys.argv[1]) if len(sys.argv) > 1 else 12)
