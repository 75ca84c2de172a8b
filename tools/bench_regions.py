#!/usr/bin/env python
"""Cost of the multi-GPU launch structure on ONE device (no transport): a 64^3 P4 block that
pretends to have neighbours on some sides runs every stage as FIRST + pack + SECOND launches (the
pipelined exchange of seigen_amd/parallel.py; --unpipelined: pack + INTERIOR + BOUNDARY), compared
with the single REGION_ALL launch.  Tells how much of the scaling loss is launch
structure rather than RCCL."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seigen_amd import _lib  # noqa: E402
from seigen_amd.backend import HipBlock  # noqa: E402
from seigen_amd.parallel import STAGE_INPUT, STAGE_OUTPUT  # noqa: E402

PIPELINED = "--unpipelined" not in sys.argv
WHOLE = "--whole" in sys.argv        # experiment: neighbours present (ghost-reading kernels), but one launch per stage
NOPACK = "--nopack" in sys.argv      # experiment: how much of the overhead the pack launches are


def run(mask, n=(64, 64, 64), degree=4, steps=10):
    h = [1.0 / 64] * 3
    blk = HipBlock(3, degree, n, h, [0.0] * 3, "left", mask)
    blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
    rng = np.random.default_rng(0)
    sides = [s for s in range(6) if mask >> s & 1]
    bufs = {}
    for kind, field in (("u", _lib.FIELD_U), ("s", _lib.FIELD_S)):
        for s in sides:
            nb = blk.halo_bytes(field, s) // 8
            bufs[(kind, s)] = (torch.zeros(nb, dtype=torch.float64, device="cuda"),
                               torch.zeros(nb, dtype=torch.float64, device="cuda"))
    for field in range(4):
        kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
        for s in sides:
            blk.halo_attach(field, s, bufs[(kind, s)][1].data_ptr())

    def step():
        for stage in range(6):
            if not sides or WHOLE:
                blk.run_stage(stage, _lib.REGION_ALL)
                continue
            if PIPELINED:      # HaloExchanger.step
                field = STAGE_OUTPUT[stage]
                kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
                blk.run_stage(stage, _lib.REGION_FIRST)
                if not NOPACK:
                    blk.halo_pack_sides(field, {s: bufs[(kind, s)][0].data_ptr() for s in sides})
                blk.run_stage(stage, _lib.REGION_SECOND)
                continue
            field = STAGE_INPUT[stage]
            kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
            for s in sides:
                blk.halo_pack(field, s, bufs[(kind, s)][0].data_ptr())
            blk.run_stage(stage, _lib.REGION_INTERIOR)
            blk.run_stage(stage, _lib.REGION_BOUNDARY)
        blk.end_step()

    for _ in range(2):
        step()
    blk.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    blk.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def breakdown(mask, n=(64, 64, 64), degree=4, steps=5):
    """ms per step spent in the packs, the interior launches and the shell launches (events on the launch stream)."""
    h = [1.0 / 64] * 3
    blk = HipBlock(3, degree, n, h, [0.0] * 3, "left", mask)
    blk.set_params(1.0, 0.5 / 64 / 8, 0.5, 0.25)
    stream = torch.cuda.ExternalStream(blk.stream_ptr())
    sides = [s for s in range(6) if mask >> s & 1]
    bufs = {}
    for kind, field in (("u", _lib.FIELD_U), ("s", _lib.FIELD_S)):
        for s in sides:
            nb = blk.halo_bytes(field, s) // 8
            bufs[(kind, s)] = (torch.zeros(nb, dtype=torch.float64, device="cuda"),
                               torch.zeros(nb, dtype=torch.float64, device="cuda"))
    for field in range(4):
        kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
        for s in sides:
            blk.halo_attach(field, s, bufs[(kind, s)][1].data_ptr())
    tot = {"pack": 0.0, "interior": 0.0, "shell": 0.0}
    for it in range(steps + 1):
        marks = []
        for stage in range(6):
            field = STAGE_INPUT[stage]
            kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(stream)
            if PIPELINED:     # "interior" = FIRST, "shell" = SECOND in the printed line
                fo = STAGE_OUTPUT[stage]
                ko = "s" if fo in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
                blk.run_stage(stage, _lib.REGION_FIRST)
                ev[1].record(stream)
                blk.halo_pack_sides(fo, {s: bufs[(ko, s)][0].data_ptr() for s in sides})
                ev[2].record(stream)
                blk.run_stage(stage, _lib.REGION_SECOND)
                ev[3].record(stream)
                marks.append([ev[1], ev[2], ev[0], ev[1], ev[2], ev[3]])
                continue
            for s in sides:
                blk.halo_pack(field, s, bufs[(kind, s)][0].data_ptr())
            ev[1].record(stream)
            blk.run_stage(stage, _lib.REGION_INTERIOR)
            ev[2].record(stream)
            blk.run_stage(stage, _lib.REGION_BOUNDARY)
            ev[3].record(stream)
            marks.append([ev[0], ev[1], ev[1], ev[2], ev[2], ev[3]])
        blk.end_step()
        blk.sync()
        if it:
            for ev in marks:
                tot["pack"] += ev[0].elapsed_time(ev[1])
                tot["interior"] += ev[2].elapsed_time(ev[3])
                tot["shell"] += ev[4].elapsed_time(ev[5])
    return {k: v / steps for k, v in tot.items()}


if __name__ == "__main__":
    if "--breakdown" in sys.argv:
        for name, mask in (("z- and z+", 0b110000), ("y+, z-, z+", 0b111000)):
            b = breakdown(mask)
            print("%-28s packs %.3f  %s %.3f  %s %.3f ms/step" % (name, b["pack"], "first" if PIPELINED else "interior",
                                                                  b["interior"], "second" if PIPELINED else "shell", b["shell"]))
        sys.exit(0)
    if "--mask" in sys.argv:      # one case only (for rocprofv3 --kernel-trace --stats: sums per kernel name)
        mask = int(sys.argv[sys.argv.index("--mask") + 1], 0)
        print("mask %s  %.3f ms/step" % (bin(mask), run(mask)))
        sys.exit(0)
    for name, mask in (("no neighbours (REGION_ALL)", 0), ("z- and z+", 0b110000), ("y+, z-, z+", 0b111000),
                       ("x+, y+, z+", 0b101010)):
        print("%-28s %.3f ms/step" % (name, run(mask)))
