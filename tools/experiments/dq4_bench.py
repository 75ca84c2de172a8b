import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from seigen_amd.backend import HipBlock
for path in ("generic", "tile"):
    for dtype in (("f64",) if path == "generic" else ("f64", "f32")):
        os.environ["SEIGEN_HIP_PATH"] = path
        blk = HipBlock(2, 4, (256, 256), [2.5, 2.5], [0.0, 0.0], "quadrilateral", dtype=dtype)
        blk.set_params(1.0, 1e-4, 0.5, 0.25)
        blk.step(5); blk.sync()
        t0 = time.perf_counter(); blk.step(100); blk.sync(); dt = (time.perf_counter() - t0) / 100
        dofs = blk.u_dofs + blk.s_dofs
        print("DQ_4 256x256 %s %s: %.1f us/step, %.1f G DoF-updates/s" % (path, dtype, dt * 1e6, dofs / dt / 1e9), flush=True)
        blk.close()
