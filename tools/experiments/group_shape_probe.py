"""How much of the F stages' trace re-fetch depends on the SHAPE of a 16-cube cell group?  The device layout groups 16
consecutive cubes of the x-fastest order; on a block only 4 cubes wide a group is a 4 x 4 patch of a z-layer instead of a
16 x 1 run, so three quarters of its y faces join cubes of the same group (rows the same wave team reads anyway).  Same
number of cubes, P4 tetrahedra, random fields; run under rocprofv3 (--stats, --pmc FETCH_SIZE) and compare per-launch numbers.
usage: group_shape_probe.py nx ny nz"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from seigen_amd import _lib
from seigen_amd.backend import HipBlock

n = tuple(int(a) for a in sys.argv[1:4])
blk = HipBlock(3, 4, n, [1.0 / 64] * 3, [0.0] * 3)
blk.set_params(1.0, 1e-4, 0.5, 0.25)
r = np.random.default_rng(0)
for f in (_lib.FIELD_U, _lib.FIELD_UH):
    blk.set_field(f, r.uniform(-1, 1, blk.field_shape(f)).astype(np.float64))
s = r.uniform(-1, 1, blk.field_shape(_lib.FIELD_S))
blk.set_field(_lib.FIELD_S, 0.5 * (s + np.swapaxes(s, -1, -2)))
blk.step(2)
blk.sync()
t0 = time.perf_counter()
blk.step(10)
blk.sync()
print("n = %s: %.3f ms/step" % (n, (time.perf_counter() - t0) / 10 * 1e3))
