import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from seigen_amd.backend import HipBlock
"""DQ_1..4 on 512 x 512 (DQ_4: 256 x 256) quadrilaterals and P1..4 on the same squares cut into triangles: MFMA tile
kernels in double and float, generic kernels in double (no sponge, no source)."""
for P in (1, 2, 3, 4):
    n = (256, 256) if P == 4 else (512, 512)
    for diag in ("quadrilateral", "left"):
        row = []
        for path, dtype in (("generic", "f64"), ("tile", "f64"), ("tile", "f32")):
            os.environ["SEIGEN_HIP_PATH"] = path
            blk = HipBlock(2, P, n, [2.5, 2.5], [0.0, 0.0], diag, dtype=dtype)
            blk.set_params(1.0, 1e-4, 0.5, 0.25)
            blk.step(5); blk.sync()
            t0 = time.perf_counter(); blk.step(100); blk.sync(); dt = (time.perf_counter() - t0) / 100
            row.append("%s %s %.1f G" % (path, dtype, (blk.u_dofs + blk.s_dofs) / dt / 1e9))
            blk.close()
        print("P%d %s %dx%d: %s" % (P, "quadrilaterals" if diag == "quadrilateral" else "triangles", n[0], n[1], ", ".join(row)), flush=True)
