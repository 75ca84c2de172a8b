#!/usr/bin/env python
"""Fixed cost of one stage launch: stage times of 64 x 64 x nz P4 blocks against nz (T = c + nz * t per kernel)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from seigen_amd.backend import HipBlock  # noqa: E402

rows = []
for nz in (2, 4, 8, 16, 32, 64):
    blk = HipBlock(3, 4, (64, 64, nz), [1.0 / 64] * 3, [0.0] * 3, "left", 0)
    blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
    blk.enable_timing(True)
    blk.step(3)
    blk.sync()
    c0 = blk.counters()
    blk.step(20)
    blk.sync()
    c1 = blk.counters()
    ms = (np.asarray(c1["kernel_ms"][:6]) - np.asarray(c0["kernel_ms"][:6])) / 20
    rows.append((nz, ms))
    print(nz, np.round(ms, 4), flush=True)
    blk.close()
nzs = np.array([r[0] for r in rows], float)
M = np.array([r[1] for r in rows])
A = np.stack([np.ones_like(nzs[2:]), nzs[2:]], 1)
for k in range(M.shape[1]):
    c, t = np.linalg.lstsq(A, M[2:, k], rcond=None)[0]
    print("stage %d: fixed %.4f ms + %.5f ms per z-layer (fit over nz >= 8)" % (k, c, t))
