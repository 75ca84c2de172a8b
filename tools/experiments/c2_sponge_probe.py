import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import seigen_amd
from seigen_amd.harness import baseline_configs as bc
seigen_amd.elastic.log = lambda s: None
import seigen_amd.harness.explosive_source as _hx
_hx.log = lambda s: None

def run(label, sponge, steps=300):
    el, _ = bc.config2(steps * 2 + 10)
    blk = el.block
    if not sponge:
        blk.set_absorption(None, 0)
    blk.step(5); blk.sync()
    blk.step(steps); blk.sync()
    ms = blk.last_step_ms() / steps
    blk.enable_timing(True)
    c0 = blk.counters(); blk.step(steps); blk.sync(); c1 = blk.counters()
    st = [round((c1["kernel_ms"][i] - c0["kernel_ms"][i]) / steps * 1e3, 1) for i in range(6)]
    dofs = blk.u_dofs + blk.s_dofs
    print(label, round(dofs / ms / 1e6), round(ms * 1e3, 1), st, flush=True)
    blk.close()

for tg in (None, "2008"):
    if tg: os.environ["SEIGEN_HIP_TILE_GRID"] = tg
    else: os.environ.pop("SEIGEN_HIP_TILE_GRID", None)
    run("grid %s sponge" % tg, True)
    run("grid %s no sponge" % tg, False)
