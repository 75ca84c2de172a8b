#!/usr/bin/env python
"""Secondary measurements for the other BASELINE configs (not the contract bench):
  c2: 2-D explosive source, 512x512 squares (524 288 triangles), P2, DG4 sponge + box-Ricker source
  c5: Marmousi 383x121 squares, P3, per-cell (lambda, mu) from seigen_amd/data/marmhard.dat
  c1: 2-D eigenmode 40x40, P1 (launch-overhead bound)
  c4s: one rank's share of config 4 (3-D explosive source 256^3 on 8 GPUs): a 128^3-cube block, P4,
       box-Ricker stress source, zero initial state - 85 GB resident on one device
  ref2d: the 2-D eigenmode N = 256, P4 of the reference's benchmark protocol, per step and stage
  c3h1..c3h4: config 3's eigenmode on HEXAHEDRA, DQ_1 / DQ_2 (96^3 cubes), DQ_3 (48^3), DQ_4 (40^3)
Prints one JSON line per config: M DoF-updates/s and ms/step (device time, hipEvents)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import seigen_amd  # noqa: E402
from seigen_amd.harness import baseline_configs as bc  # noqa: E402  (the set-ups, shared with bench.py)

seigen_amd.elastic.log = lambda s: None
import seigen_amd.harness.eigenmode as _he  # noqa: E402
import seigen_amd.harness.explosive_source as _hx  # noqa: E402
_he.log = _hx.log = lambda s: None


STAGES = False


def timed(build, steps, warmup):
    # the --stages pass steps another `steps` times: the source must still be active
    el, label = build(warmup + steps * (2 if STAGES else 1))
    blk = el.block
    blk.step(warmup)
    blk.sync()
    t0 = time.perf_counter()
    blk.step(steps)
    blk.sync()
    wall = time.perf_counter() - t0
    dev_ms = blk.last_step_ms()
    dofs = blk.u_dofs + blk.s_dofs
    u = blk.get_field_range(0, 0, 4)
    stage_us = None
    if STAGES:     # a second pass with an event pair around every stage (slower: no graph replay, events between launches)
        blk.enable_timing(True)
        c0 = blk.counters()
        blk.step(steps)
        blk.sync()
        c1 = blk.counters()
        blk.enable_timing(False)
        stage_us = [round((c1["kernel_ms"][i] - c0["kernel_ms"][i]) / steps * 1e3, 2) for i in range(6)]
    return dict(config=label, stage_us=stage_us, dofs=dofs, cells=blk.ncells, ms_per_step=dev_ms / steps,
                wall_ms_per_step=wall / steps * 1e3, value=dofs * steps / (dev_ms * 1e-3) / 1e6, finite=bool(np.isfinite(u).all()),
                kernels=sorted({blk.stage_kernel_name(st) for st in range(6)}))


def config2(steps, warmup, n=512, quadrilateral=False, dtype="f64"):
    r = timed(lambda ns: bc.config2(ns, n=n, quadrilateral=quadrilateral, dtype=dtype), steps, warmup)
    if dtype == "f32":
        r["config"] += " [FP32 second mode: 30.7 B per DoF-update (46 words of 4 B per node and step)]"
        r["bytes_per_dof_update"] = 0.5 * 184.0 / 3.0
    return r


def config2_quad(steps, warmup):
    """config 2's set-up on quadrilateral cells (DQ_2, nine nodes per square): MFMA tile kernels (SEIGEN_HIP_PATH=generic:
    the table-driven generic kernels)"""
    r = config2(steps, warmup, quadrilateral=True)
    r["config"] = r["config"].replace("c2:", "c2q (quadrilaterals, DQ_2):")
    return r


def config2_large(steps, warmup):
    """config 2's physics on a 2048x2048 mesh (8.4 M triangles, 302 M DoF): beyond every cache"""
    r = config2(steps, warmup, n=2048)
    r["config"] = r["config"].replace("c2:", "c2-large:")
    return r


def config5(steps, warmup):
    return timed(bc.config5, steps, warmup)


def config4_share(steps, warmup, n=128):
    return timed(lambda ns: bc.config4_share(ns, n=n), steps, warmup)


HEX_N = None


def config3_hex(steps, warmup, P):
    return timed(lambda ns: bc.config3_hex(ns, P, HEX_N), steps, warmup)


def config1(steps, warmup):
    return timed(bc.config1, steps, warmup)


REF2D_DEGREE = 4


def ref2d(steps, warmup, N=None, degree=None):
    """the mesh and element of the reference's own benchmark protocol (2-D eigenmode N = 256, P4: bench.py times the whole
    run(T = 2.0) of it; here: device time per step and per stage)"""
    N = N or HEX_N or 256       # --hex-n overrides the squares per axis here too
    degree = degree or REF2D_DEGREE
    def build(ns):
        em = _he.Eigenmode2DLF4(N, degree, 0.5 * (1.0 / N) / 2.0 ** (degree - 1), output=False)
        el = em.elastic
        el.u0.assign(seigen_amd.Function(el.U).interpolate(em._u(0)))
        el.s0.assign(seigen_amd.Function(el.S).interpolate(em._s(el.dt / 2)))
        return bc.ready(el, ns), "ref2d: 2D eigenmode %dx%d squares x 2 triangles, P%d" % (N, N, degree)
    return timed(build, steps, warmup)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c2", "c5", "c1"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--hex-n", type=int, default=None, help="c3h*: cubes per axis instead of the default 96 / 96 / 48 / 40; ref2d: squares per axis instead of 256")
    ap.add_argument("--degree", type=int, default=4, help="ref2d: polynomial degree instead of 4")
    ap.add_argument("--stages", action="store_true", help="also report device microseconds per stage (UH1 STEMP U1 SH1 UTEMP S1)")
    args = ap.parse_args()
    STAGES = args.stages
    HEX_N = args.hex_n
    REF2D_DEGREE = args.degree
    for c in args.configs:
        r = {"c1": config1, "c2": config2, "c5": config5, "c2l": config2_large, "c4s": config4_share,
             "c2q": config2_quad, "ref2d": ref2d,
             "c3h1": lambda st, w: config3_hex(st, w, 1), "c3h2": lambda st, w: config3_hex(st, w, 2),
             "c3h3": lambda st, w: config3_hex(st, w, 3), "c3h4": lambda st, w: config3_hex(st, w, 4),
             "c2f32": lambda st, w: config2(st, w, dtype="f32"),
             "c2qf32": lambda st, w: config2(st, w, quadrilateral=True, dtype="f32")}[c](args.steps, args.warmup)
        # bytes per DoF-update of this step (bench.py stage_accounting): 60 in 3-D, 61.3 in 2-D (FP32: half)
        bpu = r.get("bytes_per_dof_update") or (60.0 if c.startswith(("c3h", "c4")) else 184.0 / 3.0)
        r["bytes_per_dof_update"] = bpu
        r["algorithmic_GBps"] = r["value"] * 1e6 * bpu / 1e9
        r["hbm_frac"] = r["algorithmic_GBps"] / 8000.0
        print(json.dumps(r))
