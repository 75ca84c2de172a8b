"""Where a block WITH neighbours loses time against the single block, on one device without any transport: per-stage
device times (hipEvent pairs of sg_run_stage) of a 64^3 P4 block that has z- / z+ ghost buffers attached, run as
  all      one REGION_ALL launch per stage (ghost-reading kernels, natural item order)
  split    FIRST + pack + SECOND
against the block without neighbours."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from seigen_amd import _lib
from seigen_amd.backend import HipBlock
from seigen_amd.parallel import STAGE_OUTPUT

def run(mode, mask, steps=20):
    n, h = (64, 64, 64), [1.0 / 64] * 3
    blk = HipBlock(3, 4, n, h, [0.0] * 3, "left", mask)
    blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
    rng = np.random.default_rng(0)
    layer = 64 * 64 * 6
    u = rng.uniform(-1, 1, (layer,) + blk.field_shape(_lib.FIELD_U)[1:]) * 1e-3
    for k in range(64):
        blk.set_field_range(_lib.FIELD_U, k * layer, u)
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
    def step():
        for stage in range(6):
            fo = STAGE_OUTPUT[stage]
            ko = "s" if fo in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
            if mode == "split":
                blk.run_stage(stage, _lib.REGION_FIRST)
                blk.halo_pack_sides(fo, {s: bufs[(ko, s)][0].data_ptr() for s in sides})
                blk.run_stage(stage, _lib.REGION_SECOND)
            else:
                blk.run_stage(stage, _lib.REGION_ALL)
                if sides:
                    blk.halo_pack_sides(fo, {s: bufs[(ko, s)][0].data_ptr() for s in sides})
        blk.end_step()
    for _ in range(3):
        step()
    blk.sync()
    blk.enable_timing(True)
    c0 = blk.counters()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    blk.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    c1 = blk.counters()
    ms = [(c1["kernel_ms"][i] - c0["kernel_ms"][i]) / steps for i in range(6)]
    print("%-22s %-8s grid %-4s  %.3f ms/step wall   stages %s  sum %.3f" % (
        "z-/z+ neighbours" if mask else "no neighbours", mode, os.environ.get("SEIGEN_HIP_GRID_BLOCKS", "dflt"), dt,
        [round(x, 3) for x in ms], sum(ms)), flush=True)
    blk.close()

for gb in ("512", "480"):
    os.environ["SEIGEN_HIP_GRID_BLOCKS"] = gb
    run("all", 0)
    for mode in ("all", "split"):
        run(mode, 0b110000)
