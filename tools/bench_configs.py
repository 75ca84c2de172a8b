#!/usr/bin/env python
"""Secondary measurements for the other BASELINE configs (not the contract bench):
  c2: 2-D explosive source, 512x512 squares (524 288 triangles), P2, DG4 sponge + box-Ricker source
  c5: Marmousi 383x121 squares, P3, per-cell (lambda, mu) from seigen_amd/data/marmhard.dat
  c1: 2-D eigenmode 40x40, P1 (launch-overhead bound)
  c4s: one rank's share of config 4 (3-D explosive source 256^3 on 8 GPUs): a 128^3-cube block, P4,
       box-Ricker stress source, zero initial state - 85 GB resident on one device
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
from seigen_amd import (ElasticLF4, Expression, Function, FunctionSpace, RectangleMesh, Vp, cfl_dt)  # noqa: E402
from seigen_amd.harness.eigenmode import Eigenmode2DLF4  # noqa: E402
from seigen_amd.harness.explosive_source import ExplosiveSourceLF4  # noqa: E402

seigen_amd.elastic.log = lambda s: None
import seigen_amd.harness.eigenmode as _he  # noqa: E402
import seigen_amd.harness.explosive_source as _hx  # noqa: E402
_he.log = _hx.log = lambda s: None


STAGES = False


def timed(elastic, steps, warmup):
    elastic.setup()
    blk = elastic.block
    if elastic.source:
        # the --stages pass steps another `steps` times: its source must still be active
        times = [elastic.dt * (k + 1) for k in range(warmup + steps * (2 if STAGES else 1))]
        nodes, values, static = elastic._source_table(times)
        blk.set_source(nodes, values, static=static)
    else:
        blk.set_source([], None)
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
    return dict(stage_us=stage_us, dofs=dofs, cells=blk.ncells, ms_per_step=dev_ms / steps, wall_ms_per_step=wall / steps * 1e3,
                value=dofs * steps / (dev_ms * 1e-3) / 1e6, finite=bool(np.isfinite(u).all()))


def config2(steps, warmup, n=512, quadrilateral=False, dtype="f64"):
    h = 2.5
    ex = ExplosiveSourceLF4()
    # Courant number 0.05 (default of the reference's tiling harness, tests/tiling/utils.py:51-52):
    # the 0.5 of explosive_source_lf4.py:31 is unstable with the explicit sponge
    el = ex.setup(Lx=n * h, Ly=n * h, h=h, degree=2, courant_number=0.05, quadrilateral=quadrilateral, dtype=dtype)
    r = timed(el, steps, warmup)
    r["config"] = "c2: 2D explosive source %dx%d squares, P2, sponge+source" % (n, n)
    if dtype == "f32":
        r["config"] += " [FP32 second mode: 32 B per DoF-update]"
        r["bytes_per_dof_update"] = 32
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
    from seigen_amd.marmousi import cell_material, NX, NY, H
    mesh = RectangleMesh(NX - 1, NY - 1, (NX - 1) * H, (NY - 1) * H)        # seigen/marmousi.py:18-21
    el = ElasticLF4.create(mesh, "DG", 3, dimension=2, solver="explicit", output=False)
    lam, mu, vp = cell_material(el.U)
    el.density, el.l, el.mu = 1.0, lam, mu
    el.dt = cfl_dt(H, float(vp.max()), 0.05)
    # Ricker source near the surface, zero initial state
    a = 159.42
    sx, sy = 0.5 * (NX - 1) * H, (NY - 1) * H - 24.0
    box = "x[0] >= %r && x[0] <= %r && x[1] >= %r && x[1] <= %r" % (sx - 12.0, sx + 12.0, sy - 12.0, sy + 12.0)
    code = "%s ? (-1.0 + 2*a*pow(t - 0.3, 2))*exp(-a*pow(t - 0.3, 2)) : 0.0" % box
    el.source_expression = Expression(((code, "0.0"), ("0.0", code)), a=a, t=0)
    el.source_function = Function(el.S)
    el.source = el.source_expression
    r = timed(el, steps, warmup)
    r["config"] = "c5: Marmousi %dx%d squares, P3, per-cell lambda/mu" % (NX - 1, NY - 1)
    return r


def config4_share(steps, warmup, n=128):
    from seigen_amd import BoxMesh
    h = 2.5
    mesh = BoxMesh(n, n, n, n * h, n * h, n * h)
    el = ElasticLF4.create(mesh, "DG", 4, dimension=3, solver="explicit", output=False)
    el.density, el.mu, el.l = 1.0, 3600.0, 3599.3664            # explosive_source_lf4.py:21-23
    el.dt = cfl_dt(h, Vp(el.mu, el.l, el.density), 0.05) / 8   # 2^(P-1) as in eigenmode_3d.py's dt rule
    c = 0.5 * n * h
    box = " && ".join("x[%d] >= %r && x[%d] <= %r" % (a, c - 2 * h, a, c + 2 * h) for a in range(3))
    code = "%s ? (-1.0 + 2*a*pow(t - 0.3, 2))*exp(-a*pow(t - 0.3, 2)) : 0.0" % box
    z = "0.0"
    el.source_expression = Expression(((code, z, z), (z, code, z), (z, z, code)), a=159.42, t=0)
    el.source_function = Function(el.S)
    el.source = el.source_expression
    r = timed(el, steps, warmup)
    r["config"] = "c4s: one rank's share of config 4: %d^3 cubes x 6 tets, P4, box-Ricker source" % n
    return r


def config3_hex(steps, warmup, P):
    """tests/eigenmode/eigenmode_3d.py on UnitCubeMesh(N, N, N, hexahedral=True): the analytic mode as initial state"""
    from seigen_amd.harness.eigenmode import Eigenmode3DLF4
    N = {1: 96, 2: 96, 3: 48, 4: 40}[P]
    em = Eigenmode3DLF4(N, P, 0.5 * (1.0 / N) / 2.0 ** (P - 1), output=False, hexahedral=True)
    el = em.elastic
    el.u0.assign(Function(el.U).interpolate(em._u(0)))
    el.s0.assign(Function(el.S).interpolate(em._s(el.dt / 2)))
    r = timed(el, steps, warmup)
    r["config"] = "c3h%d: 3D eigenmode on %d^3 hexahedra, DQ_%d" % (P, N, P)
    return r


def config1(steps, warmup):
    em = Eigenmode2DLF4(40, 1, 0.0125, output=False)
    el = em.elastic
    el.u0.assign(Function(el.U).interpolate(em._u(0)))
    el.s0.assign(Function(el.S).interpolate(em._s(el.dt / 2)))
    r = timed(el, steps, warmup)
    r["config"] = "c1: 2D eigenmode 40x40 squares, P1"
    return r


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c2", "c5", "c1"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stages", action="store_true", help="also report device microseconds per stage (UH1 STEMP U1 SH1 UTEMP S1)")
    args = ap.parse_args()
    STAGES = args.stages
    for c in args.configs:
        r = {"c1": config1, "c2": config2, "c5": config5, "c2l": config2_large, "c4s": config4_share,
             "c2q": config2_quad,
             "c3h1": lambda st, w: config3_hex(st, w, 1), "c3h2": lambda st, w: config3_hex(st, w, 2),
             "c3h3": lambda st, w: config3_hex(st, w, 3), "c3h4": lambda st, w: config3_hex(st, w, 4),
             "c2f32": lambda st, w: config2(st, w, dtype="f32"),
             "c2qf32": lambda st, w: config2(st, w, quadrilateral=True, dtype="f32")}[c](args.steps, args.warmup)
        r["algorithmic_GBps"] = r["value"] * 1e6 * r.get("bytes_per_dof_update", 64) / 1e9
        r["hbm_frac"] = r["algorithmic_GBps"] / 8000.0
        print(json.dumps(r))
