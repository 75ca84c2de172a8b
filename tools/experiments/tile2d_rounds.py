#!/usr/bin/env python
"""Stage times of 383 x ny P3 blocks (config 5 is ny = 121) against the number of items per launch: where the rounds of
waves show.  One item = 16 squares x one class = one wave's unit of work; a P3 plain stage keeps 4 waves per SIMD
(4096 slots), a fused one 3 (3072)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from seigen_amd.backend import HipBlock  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for ny in (8, 16, 32, 48, 64, 80, 85, 96, 112, 121, 128, 144, 160, 171, 192, 256):
    blk = HipBlock(2, P, (383, ny), [24.0, 24.0], [0.0, 0.0], "left", 0)
    blk.set_params(1.0, 1e-3, 0.5, 0.25)
    blk.enable_timing(True)
    blk.step(5)
    blk.sync()
    c0 = blk.counters()
    blk.step(100)
    blk.sync()
    c1 = blk.counters()
    ms = (np.asarray(c1["kernel_ms"][:6]) - np.asarray(c0["kernel_ms"][:6])) / 100
    items = (383 * ny + 15) // 16 * 2
    print("ny %4d items %6d  (/4096 = %.2f, /3072 = %.2f)  stage us %s  sum %.1f" % (
        ny, items, items / 4096, items / 3072, np.round(ms * 1e3, 1), ms.sum() * 1e3), flush=True)
    blk.close()
