#!/usr/bin/env python
"""2-D kernel families over block sizes: ms per LF4 step (device time, best of 3 x `steps`) for the generic,
lane-per-cell and MFMA tile kernels.  Chooses SG_TILE2D_MIN_CELLS in api.cpp.
usage: path_sweep2d.py [--paths generic,lane,tile] [--degrees 1,2,3,4] [--sizes 40x40,128x128,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from seigen_amd.backend import HipBlock


def ms_per_step(path, degree, n, steps=60):
    os.environ["SEIGEN_HIP_PATH"] = path
    blk = HipBlock(2, degree, n, tuple(1.0 / x for x in n), (0.0, 0.0))
    blk.set_params(1.0, 1e-5, 0.5, 0.25)
    blk.step(10)
    best = 1e30
    for _ in range(3):
        blk.step(steps)
        best = min(best, blk.last_step_ms() / steps)
    cells = blk.ncells
    blk.close()
    return best, cells


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--paths", default="generic,lane,tile")
    ap.add_argument("--degrees", default="1,2,3,4")
    ap.add_argument("--sizes", default="40x40,64x64,90x90,128x128,192x192,383x121,256x256,384x384,512x512,1024x1024")
    a = ap.parse_args()
    paths = a.paths.split(",")
    for p in [int(x) for x in a.degrees.split(",")]:
        for sz in a.sizes.split(","):
            n = tuple(int(x) for x in sz.split("x"))
            r = {}
            cells = 0
            for path in paths:
                r[path], cells = ms_per_step(path, p, n)
            nd = (p + 1) * (p + 2) // 2
            best = min(r, key=r.get)
            frac = cells * nd * 6 * 64 / (r[best] * 1e-3) / 8e12
            print("dim 2 P%d n=%-9s cells %8d: " % (p, sz, cells) + "  ".join("%s %.4f" % (k, v) for k, v in r.items()) +
                  " ms  -> %s (%.0f %% of the 64 B/DoF HBM roofline)" % (best, 100 * frac), flush=True)
