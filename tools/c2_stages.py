import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from seigen_amd import _lib
from seigen_amd.backend import HipBlock
for path in ("lane", "generic"):
    os.environ["SEIGEN_HIP_PATH"] = path
    n = (512, 512)
    blk = HipBlock(2, 2, n, (2.5, 2.5), (0.0, 0.0))
    blk.set_params(1.0, 1e-4, 3599.3664, 3600.0)
    rng = np.random.default_rng(0)
    blk.set_field(_lib.FIELD_U, rng.uniform(-1, 1, blk.field_shape(_lib.FIELD_U)))
    blk.step(5); blk.sync()
    blk.enable_timing(True)
    c0 = blk.counters()
    blk.step(50); blk.sync()
    c1 = blk.counters()
    ms = [(c1["kernel_ms"][i] - c0["kernel_ms"][i]) / 50 for i in range(6)]
    print(path, "stage us:", [round(x * 1e3, 1) for x in ms], "sum %.3f ms" % sum(ms))
    blk.enable_timing(False)
    t0 = time.perf_counter(); blk.step(200); blk.sync(); print("   untimed step: %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
    blk.close()
