"""What the per-launch hipEvent pairs of sg_enable_timing cost: config 3, `steps` LF4 steps through sg_step with timing
off and on, alternating, wall clock around a synchronised region."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from seigen_amd import _lib
from seigen_amd.backend import HipBlock
n, h = (64, 64, 64), [1.0 / 64] * 3
blk = HipBlock(3, 4, n, h, [0.0] * 3, "left")
blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
rng = np.random.default_rng(0)
layer = 64 * 64 * 6
u = rng.uniform(-1, 1, (layer,) + blk.field_shape(_lib.FIELD_U)[1:]) * 1e-3
for k in range(64):
    blk.set_field_range(_lib.FIELD_U, k * layer, u)
blk.step(5); blk.sync()
for rep in range(3):
    for timing in (False, True):
        blk.enable_timing(timing)
        blk.step(3); blk.sync()
        t0 = time.perf_counter()
        blk.step(100); blk.sync()
        print("timing %-5s  %.4f ms/step" % (timing, (time.perf_counter() - t0) / 100 * 1e3), flush=True)
blk.enable_timing(False)
