#!/usr/bin/env python
"""Source normalisation and P-wave propagation of the HIP path against the exact 2-D full-space solution
of an explosive line source (oracle/analytic.py): the explosive-source set-up with the source moved to the
middle of the domain, unit-moment projected source, receivers 30-45 m away in four directions; compared before
the first reflection (free surface / sponge edge) can arrive.  Needs a GPU.  (Kept under tests/: it uses the oracle's exact solutions as the checker.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.explosive_source as hes
    from oracle.analytic import explosive_line_source_2d
    helpers.log = seigen_amd.elastic.log = hes.log = lambda s: None
    src = (150.0, 75.0)
    recv = ((195.0, 75.0), (150.0, 110.0), (180.0, 105.0), (118.0, 51.0))
    for (h, P, dt) in ((2.5, 2, 0.001), (2.5, 4, 0.001), (1.25, 3, 0.0005), (0.625, 4, 0.00025)):
        ex = hes.ExplosiveSourceLF4()
        el = ex.setup(h=h, degree=P, dt=dt, source_mode="project", source_x=src[0], source_y=src[1])
        every = int(round(0.005 / dt))
        times, tr = ex.record_receivers(1.1, receivers=recv, every=every)
        print("h %.3f P%d dt %.5f (alpha = %.4f m/s)" % (h, P, dt, ex.Vp))
        for i, (x, y) in enumerate(recv):
            dx, dy = x - src[0], y - src[1]
            r = float(np.hypot(dx, dy))
            vr = explosive_line_source_2d(r, times, ex.Vp)
            ours_r = tr[:, i, 0] * dx / r + tr[:, i, 1] * dy / r
            ours_t = -tr[:, i, 0] * dy / r + tr[:, i, 1] * dx / r
            a = np.dot(ours_r, vr) / np.dot(vr, vr)
            err = np.linalg.norm(ours_r - vr) / np.linalg.norm(vr)
            print("   receiver (%.0f, %.0f) r = %.2f m: amplitude ratio %.5f  rel. L2 misfit %.5f  corr %.6f  max |v_r| %.3e  "
                  "transverse/radial %.1e" % (x, y, r, a, err, np.corrcoef(ours_r, vr)[0, 1], np.abs(vr).max(),
                                              np.abs(ours_t).max() / np.abs(vr).max()))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
