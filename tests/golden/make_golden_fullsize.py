#!/usr/bin/env python
"""Oracle-backed goldens at BASELINE's full sizes (tests/fullsize_cases.py) -> tests/golden/fullsize_c{2,3,5}.npz.
Run from the repo root in the build container:  python tests/golden/make_golden_fullsize.py [c3] [c2] [c5] [c2q]

The step is the oracle's plain-C restatement (oracle/c/seigen_oracle.c through oracle/cport.py: so_step_ex with
sponge, source, per-cell material and density), itself validated against the numpy oracle on small meshes
(tests/test_oracle_cport.py).  c3 needs about 20 GB and a few minutes on 8 cores.

Stored per case: the sampled cells' u, s after the last step and the contents of the two work fields the product
keeps (utemp, sh1 of the last step); sums of every field over each slab of the slowest mesh axis; a digest of the
inputs the two sides must share (material arrays of c5)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import harness, mesh as omesh  # noqa: E402
from oracle.cport import CPort, sponge_blocks  # noqa: E402
from tests import fullsize_cases as fc  # noqa: E402


def save(name, cp, u, s, nlayers, forced, extra=None):
    uh, sh = cp.work[0], cp.work[1]
    cells = fc.sample_cells(u.shape[0], forced)
    out = dict(cells=cells, u=u[cells], s=s[cells], uh=uh[cells], sh=sh[cells],
               u_layers=fc.layer_sums(u, nlayers), s_layers=fc.layer_sums(s, nlayers),
               uh_layers=fc.layer_sums(uh, nlayers), sh_layers=fc.layer_sums(sh, nlayers))
    out.update(extra or {})
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "written:", len(cells), "cells", flush=True)


def c3():
    c = fc.C3
    t0 = time.time()
    em = None
    m = omesh.UnitCubeMesh(c["n"], c["n"], c["n"])
    print("c3 mesh", time.time() - t0, flush=True)
    cp = CPort(m, c["P"])
    print("c3 tables", time.time() - t0, flush=True)
    # eigenmode_3d.py:25-39 through the oracle harness's formulas
    em = harness.Eigenmode3D.__new__(harness.Eigenmode3D)
    em.A = np.sqrt(2 * c["rho"] * c["mu"])
    em.O = np.pi * np.sqrt(2 * c["mu"] / c["rho"])
    X = m.node_coords(c["P"])
    u = em.u_exact(X, 0.0)
    s = em.s_exact(X, c["dt"] / 2.0)
    del X
    u, s = cp.step_ex(u, s, c["rho"], c["dt"], c["lam"], c["mu"], c["steps"], inplace=True)
    print("c3 steps", time.time() - t0, flush=True)
    save("fullsize_c3.npz", cp, u, s, c["n"], ())


def box_source(m, P, lo, hi, dt, steps):
    """nodal interpolation of the box indicator (explosive_source_lf4.py:36-40) x Ricker centred at step 10"""
    X = m.node_coords(P)
    inb = (X[..., 0] >= lo[0]) & (X[..., 0] <= hi[0]) & (X[..., 1] >= lo[1]) & (X[..., 1] <= hi[1])
    nodes = np.nonzero(inb.reshape(-1))[0]
    vals = np.zeros((steps, len(nodes), 2, 2))
    for k in range(steps):
        w = fc.ricker((k + 1) * dt, 10 * dt)
        vals[k, :, 0, 0] = vals[k, :, 1, 1] = w
    return nodes, vals


def c2(quadrilateral=False):
    c = fc.C2
    n, h, P = c["n"], c["h"], c["P"]
    L = n * h
    m = omesh.RectangleMesh(n, n, L, L, quadrilateral=quadrilateral)
    cp = CPort(m, P)
    Xs = m.node_coords(c["sigma_degree"])
    sig = np.where((Xs[..., 0] <= c["sponge"]) | (Xs[..., 0] >= L - c["sponge"]) | (Xs[..., 1] <= c["sponge"]), c["sigma"], 0.0)
    sx, sy, hw = c["src"][0], c["src"][1], c["src_half"]
    nodes, vals = box_source(m, P, (sx - hw, sy - hw), (sx + hw, sy + hw), c["dt"], c["steps"])
    assert len(nodes) > 0
    cp.set_extra(sponge=sponge_blocks(m, P, sig, c["sigma_degree"]), src_nodes=nodes, src_values=vals)
    u, s = fc.smooth_state(m.node_coords(P), c["k"], c["s_scale"])
    u, s = cp.step_ex(u, s, c["rho"], c["dt"], c["lam"], c["mu"], c["steps"], inplace=True)
    nd = cp.nd
    save("fullsize_c2q.npz" if quadrilateral else "fullsize_c2.npz", cp, u, s, n, np.unique(nodes // nd), dict(src_nodes=nodes))


def c2q():
    """config 2's set-up on quadrilateral cells (DQ_2, 512 x 512 squares): build-defined row"""
    c2(quadrilateral=True)


def c5():
    from seigen_amd import FunctionSpace, RectangleMesh
    from seigen_amd.marmousi import NX, NY, H, cell_material, gardner_density
    c = fc.C5
    nx, ny, P = NX - 1, NY - 1, c["P"]
    m = omesh.RectangleMesh(nx, ny, nx * H, ny * H)
    # the material arrays are INPUT data of the case (Vp table of seigen/marmousi.py at the cell centroids);
    # the test recomputes them and checks the digest stored here
    lam, mu, vp = cell_material(FunctionSpace(RectangleMesh(nx, ny, nx * H, ny * H), "DG", P), density=gardner_density)
    rho = gardner_density(vp)
    dt = c["courant"] * H / float(vp.max())
    sx, sy, hw = 0.5 * nx * H, ny * H - 24.0, c["src_half"]
    nodes, vals = box_source(m, P, (sx - hw, sy - hw), (sx + hw, sy + hw), dt, c["steps"])
    assert len(nodes) > 0
    cp = CPort(m, P)
    cp.set_extra(lam=lam, mu=mu, rho=rho, rho_physical=True, src_nodes=nodes, src_values=vals)
    u, s = fc.smooth_state(m.node_coords(P), c["k"], c["s_scale"])
    u, s = cp.step_ex(u, s, 1.0, dt, 0.0, 0.0, c["steps"], inplace=True)
    save("fullsize_c5.npz", cp, u, s, ny, np.unique(nodes // cp.nd),
         dict(src_nodes=nodes, dt=np.array(dt), material_digest=np.array(fc.digest(lam, mu, rho))))


if __name__ == "__main__":
    what = sys.argv[1:] or ["c5", "c2", "c3"]
    for w in what:
        {"c2": c2, "c3": c3, "c5": c5, "c2q": c2q}[w]()
