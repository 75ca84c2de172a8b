"""One box, one process, wall clock per LF4 step of a 64^3 P4 block (timing off):
  single/sg_step   no neighbours, sg_step
  single/stages    no neighbours, one sg_run_stage(REGION_ALL) per stage from Python
  ghost/all        z-/z+ ghost buffers attached, one REGION_ALL launch per stage + pack (no transport)
  ghost/split      FIRST + pack + SECOND from Python (no transport)
  native/split     csrc/comm.cpp, rank is its own z-/z+ neighbour over RCCL, FIRST + SECOND
  (native/ordered: the ordered schedule of tools/experiments/ordered_schedule.patch, when that patch is applied)
(the native modes without the RCCL calls: a library built with `SRC=comm tools/build_variant.sh dry -DSG_COMM_DRY -x hip`
 and named in SEIGEN_HIP_LIB - wrong results by design, which is why it is a compile-time define of experiment builds and
 no longer an environment switch of the shipped library.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from seigen_amd import _lib
from seigen_amd.backend import HipBlock, comm_unique_id
from seigen_amd.parallel import STAGE_OUTPUT

STEPS = int(os.environ.get("STEPS", "40"))

def fill(blk):
    rng = np.random.default_rng(0)
    layer = 64 * 64 * 6
    u = rng.uniform(-1, 1, (layer,) + blk.field_shape(_lib.FIELD_U)[1:]) * 1e-3
    for k in range(64):
        blk.set_field_range(_lib.FIELD_U, k * layer, u)

def timeit(label, blk, step):
    step(3); blk.sync()
    t0 = time.perf_counter()
    step(STEPS); blk.sync()
    print("%-16s grid %-4s %.3f ms/step" % (label, os.environ.get("SEIGEN_HIP_GRID_BLOCKS", "dflt"), (time.perf_counter() - t0) / STEPS * 1e3), flush=True)

def run(mode):
    n, h = (64, 64, 64), [1.0 / 64] * 3
    mask = 0 if mode.startswith("single") else 0b110000
    if mode.startswith("native"):
        os.environ["SEIGEN_HALO_ORDERED"] = "1" if mode.endswith("ordered") else "0"
    blk = HipBlock(3, 4, n, h, [0.0] * 3, "left", mask)
    blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
    fill(blk)
    if mode == "single/sg_step":
        timeit(mode, blk, lambda k: blk.step(k))
    elif mode.startswith("native"):
        blk.comm_init(comm_unique_id(), 0, 1, [None, None, None, None, 0, 0])
        timeit(mode, blk, lambda k: blk.step(k))
    else:
        sides = [s for s in range(6) if mask >> s & 1]
        bufs = {}
        for kind, field in (("u", _lib.FIELD_U), ("s", _lib.FIELD_S)):
            for s in sides:
                nb = blk.halo_bytes(field, s) // 8
                bufs[(kind, s)] = (torch.zeros(nb, dtype=torch.float64, device="cuda"), torch.zeros(nb, dtype=torch.float64, device="cuda"))
        for field in range(4):
            kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
            for s in sides:
                blk.halo_attach(field, s, bufs[(kind, s)][1].data_ptr())
        torch.cuda.synchronize()
        def step(k):
            for _ in range(k):
                for stage in range(6):
                    fo = STAGE_OUTPUT[stage]
                    ko = "s" if fo in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
                    if mode == "ghost/split":
                        blk.run_stage(stage, _lib.REGION_FIRST)
                        blk.halo_pack_sides(fo, {s: bufs[(ko, s)][0].data_ptr() for s in sides})
                        blk.run_stage(stage, _lib.REGION_SECOND)
                    else:
                        blk.run_stage(stage, _lib.REGION_ALL)
                        if sides:
                            blk.halo_pack_sides(fo, {s: bufs[(ko, s)][0].data_ptr() for s in sides})
                blk.end_step()
        timeit(mode, blk, step)
    blk.close()

for gb in ("512", "480"):
    os.environ["SEIGEN_HIP_GRID_BLOCKS"] = gb
    for mode in ("single/sg_step", "single/stages", "ghost/all", "ghost/split", "native/split", "native/ordered", "single/sg_step"):
        run(mode)
