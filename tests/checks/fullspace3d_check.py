#!/usr/bin/env python
"""3-D: source normalisation and P-wave propagation of the MFMA path (the ingredients of BASELINE config 4) against the
exact full-space solution of an explosive point source (oracle/analytic.py explosive_point_source_3d).  A 120 m cube of
48^3 cubes x 6 tets, P3, the explosive test's material, a Ricker stress source in the 2 x 2 x 2 cubes around the centre
(the indicator of exactly those 48 cells, assigned cell by cell - a nodal interpolation of the box would drop or add the
nodes on its faces: volume 125 m^3), receivers 25 m away, before the first reflection
from the (free) outer boundary.  Needs a GPU.  (Kept under tests/: it uses the oracle's exact solutions as the checker.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(n=48, P=3, h=2.5, T=0.8, every=10, recv=None):
    import seigen_amd
    from seigen_amd import BoxMesh, ElasticLF4, Expression, Function, Vp, cfl_dt
    from seigen_amd.functionspace import evaluate_at, locate
    seigen_amd.elastic.log = lambda s: None
    L = n * h
    mesh = BoxMesh(n, n, n, L, L, L)
    el = ElasticLF4.create(mesh, "DG", P, dimension=3, solver="explicit", output=False)
    el.density, el.mu, el.l = 1.0, 3600.0, 3599.3664
    vp = Vp(el.mu, el.l, el.density)
    el.dt = cfl_dt(h, vp, 0.05) / 2 ** (P - 1)
    import math
    c, a = 0.5 * L, 159.42
    cen = el.S.node_coords().mean(axis=1)                       # cell centroids
    inbox = (np.abs(cen - c) < h).all(axis=1)
    assert int(inbox.sum()) == 48
    pat = np.zeros((el.S.ncells, el.S.nd, 3, 3))
    for i in range(3):
        pat[inbox, :, i, i] = 1.0
    el.source_function = Function(el.S)
    el.source_function.assign(pat)
    el.source_time_function = lambda t: (-1.0 + 2 * a * (t - 0.3) ** 2) * math.exp(-a * (t - 0.3) ** 2)
    el.setup()
    times = el.step_times(T)
    el.upload_source(times)
    recv = recv or ((c + 25.3, c + 0.4, c + 0.7), (c + 14.3, c - 17.2, c + 11.1), (c - 0.6, c + 0.3, c - 24.8))
    locs = [locate(el.U, r) for r in recv]
    out_t, out_v, done = [], [], 0
    while done + every <= len(times):
        el._advance(every)
        done += every
        out_t.append(times[done - 1])
        out_v.append([evaluate_at(el.u1, r, loc) for r, loc in zip(recv, locs)])
    return np.array(out_t), np.array(out_v), np.array(recv) - c, vp, (2 * h) ** 3


def main():
    from oracle.analytic import explosive_point_source_3d
    for n, P in ((48, 3), (48, 4)):
        t, tr, rel, vp, vol = run(n=n, P=P)
        print("%d^3 cubes, P%d, alpha %.3f, source volume %.1f m^3" % (n, P, vp, vol))
        gx, gw = np.polynomial.legendre.leggauss(4)
        a = vol ** (1.0 / 3.0)                       # edge of the source box
        for i in range(len(rel)):
            r = float(np.linalg.norm(rel[i]))
            # the exact field of the BOX = the point-source solution integrated over the source positions
            vbox = np.zeros((len(t), 3))
            for p0, w0 in zip(gx, gw):
                for p1, w1 in zip(gx, gw):
                    for p2, w2 in zip(gx, gw):
                        dvec = rel[i] - 0.5 * a * np.array([p0, p1, p2])
                        rr = float(np.linalg.norm(dvec))
                        vbox += (w0 * w1 * w2 / 8.0) * explosive_point_source_3d(rr, t, vp, volume=vol)[:, None] * (dvec / rr)[None, :]
            vr = vbox @ (rel[i] / r)
            ours = tr[:, i, :] @ (rel[i] / r)
            tang = np.linalg.norm(tr[:, i, :] - ours[:, None] * (rel[i] / r)[None, :], axis=1)
            print("   r = %.2f m: amplitude ratio %.4f  rel. L2 misfit %.4f  corr %.5f  transverse/radial %.1e"
                  % (r, np.dot(ours, vr) / np.dot(vr, vr), np.linalg.norm(ours - vr) / np.linalg.norm(vr), np.corrcoef(ours, vr)[0, 1],
                     tang.max() / np.abs(vr).max()))


if __name__ == "__main__":
    main()
