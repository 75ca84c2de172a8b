#!/usr/bin/env python
"""HIP path against the exact half-space solution of a buried explosive line source (oracle/analytic.py
explosive_box_source_halfspace: the point-source solution integrated over the 1 m source box): the reference's explosive-source set-up with the unit-moment projected source, receivers
inside cells at several depths and two distances; before the reflections from the sponge edges arrive.  Needs a GPU.  (Kept under tests/: it uses the oracle's exact solutions as the checker.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.explosive_source as hes
    from oracle.analytic import explosive_box_source_halfspace
    helpers.log = seigen_amd.elastic.log = hes.log = lambda s: None
    # the reference's domain puts the source 25 m from the left sponge, whose abrupt onset (sigma 0 -> 1000) reflects:
    # P waves come back to the receivers inside uy.py's windows.  SRC_X / LX move the source away from it.
    LX, SRC_X = float(os.environ.get("HS_LX", "300")), float(os.environ.get("HS_SRC_X", "45"))
    depths = (0.3, 1.0, 2.3)
    xs = (SRC_X + 45.3, SRC_X + 95.3)
    recv = [(x, 150.0 - z) for x in xs for z in depths]
    out = {}
    for (h, P, dt) in ((1.25, 3, 0.0005), (0.625, 4, 0.00025)):
        ex = hes.ExplosiveSourceLF4()
        ex.setup(Lx=LX, h=h, degree=P, dt=dt, source_mode="project", source_x=SRC_X)
        times, tr = ex.record_receivers(2.5, receivers=recv, every=int(round(0.005 / dt)))
        out["h%g_P%d" % (h, P)] = tr
        print("h %.3f P%d" % (h, P))
        for i, (x, y) in enumerate(recv):
            z = 150.0 - y
            vx, vz = explosive_box_source_halfspace(x - SRC_X, z, 1.0, times, ex.Vp, ex.Vs, period=2000.0)
            t1 = (x - SRC_X) / ex.Vs * 0.9194 ** -1 + 0.75          # end of the Rayleigh wave train
            w = (times > 0.3) & (times < t1)
            res = []
            for ours, exact in ((tr[:, i, 0], vx), (-tr[:, i, 1], vz)):
                res.append((np.dot(ours[w], exact[w]) / np.dot(exact[w], exact[w]), np.corrcoef(ours[w], exact[w])[0, 1],
                            np.linalg.norm(ours[w] - exact[w]) / np.linalg.norm(exact[w])))
            print("   x %.1f depth %.1f:  vx ratio %.5f corr %.6f misfit %.5f    vz ratio %.5f corr %.6f misfit %.5f"
                  % (x, z, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2]))
        sys.stdout.flush()
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "halfspace_traces_lx%g.npz" % LX), times=times, recv=np.array(recv), **out)


if __name__ == "__main__":
    main()
