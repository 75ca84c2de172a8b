#!/usr/bin/env python
"""Which source / receiver depths would REF-C2, REF-C3 correspond to?  Unit-moment projected source at several
depths below the free surface, receivers at x = 90 and 140 at several depths; both components against REF-C
(columns ux, uy).  h = 1.25, P3 (mesh-converged to 0.1 % in the far field: tools/refc_convergence.py).  Needs a GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
WIN = {90.0: (0.5, 1.5), 140.0: (1.0, 2.5)}


def main():
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.explosive_source as hes
    helpers.log = seigen_amd.elastic.log = hes.log = lambda s: None
    refs = {90.0: np.loadtxt(os.path.join(GOLD, "ref_c2.txt")), 140.0: np.loadtxt(os.path.join(GOLD, "ref_c3.txt"))}
    rdepth = (0.0, 0.25, 0.5, 1.0, 1.5, 2.0)
    recv = [(x, 150.0 - z) for x in (90.0, 140.0) for z in rdepth]
    for sdepth in [float(v) for v in os.environ.get("SCAN_SRC", "0.5,1.0,1.5,2.0,3.0").split(",")]:
        ex = hes.ExplosiveSourceLF4()
        ex.setup(h=1.25, degree=3, dt=0.0005, source_mode="project", source_y=150.0 - sdepth)
        times, tr = ex.record_receivers(2.5, receivers=recv, every=10)
        print("source depth %.2f m" % sdepth)
        for i, (x, y) in enumerate(recv):
            ref = refs[x]
            w = (times > WIN[x][0]) & (times < WIN[x][1])
            res = []
            for comp in (0, 1):
                o, r = tr[w, i, comp], ref[w, 1 + comp]
                res.append((np.dot(o, r) / np.dot(r, r), np.corrcoef(o, r)[0, 1], np.linalg.norm(np.abs(o) - 0) and
                            np.linalg.norm(o - (np.dot(o, r) / np.dot(r, r)) * r) / np.linalg.norm(o)))
            print("   x %5.0f depth %.2f:  ux ratio %+.4f corr %+.4f shape-misfit %.3f   uy ratio %+.4f corr %+.4f shape-misfit %.3f"
                  % (x, 150.0 - y, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
