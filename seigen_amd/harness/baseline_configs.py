"""The workloads BASELINE.json lists besides its headline configuration, set up through the solver class exactly as
the reference's scripts set up theirs - shared by `bench.py` (the driver-timed `"configs"` object) and
`tools/bench_configs.py` (secondary measurements).  Every function returns a solver that is ready to step
(`setup()` done, source of the next `nsteps` steps on the device) and a one-line description.

  c1   tests/eigenmode 2-D, 40 x 40 squares, P1                         (tests/eigenmode/eigenmode_2d.py:7-36)
  c2   2-D explosive source, 512 x 512 squares, P2, DG4 sponge + source  (tests/explosive_source/explosive_source_lf4.py:7-56)
  c5   Marmousi 383 x 121 squares, P3, per-cell lambda / mu              (seigen/marmousi.py:4-24)
  c4s  one rank's 128^3-cube share of config 4 (3-D explosive source 256^3 on 8 GPUs), P4, no neighbours; from a smooth
       NON-ZERO state (on an all-zero block the package clocks 5-8 % higher: a flattering figure), and - `sponge=True` - with
       the sponge of the reference's 2-D problem carried over to 3-D: sigma = 1000 in strips 8 cells wide on five faces,
       none on the free surface (tests/explosive_source/explosive_source_lf4.py:42-45)
  ref  the reference's own benchmark protocol: 2-D eigenmode N = 256, P = 4, T = 2.0, explicit
       (tests/eigenmode/README.md:7-13, eigenmode_bench.py:19-41) - `reference_strong_2d`, a whole `run(T)`
"""
import time

from seigen_amd import BoxMesh, ElasticLF4, Expression, Function, RectangleMesh, Vp, cfl_dt
from seigen_amd.harness.eigenmode import Eigenmode2DLF4, Eigenmode3DLF4
from seigen_amd.harness.explosive_source import ExplosiveSourceLF4

A_RICKER = 159.42       # explosive_source_lf4.py:35


def ready(el, nsteps):
    """setup() + the source of the next `nsteps` steps (what `run` does before its loop, elastic.py:244-255, :285-288)"""
    el.setup()
    if el.source:
        el.upload_source([el.dt * (k + 1) for k in range(nsteps)])
    else:
        el.block.set_source([], None)
    return el


def config1(nsteps):
    em = Eigenmode2DLF4(40, 1, 0.0125, output=False)
    el = em.elastic
    el.u0.assign(Function(el.U).interpolate(em._u(0)))
    el.s0.assign(Function(el.S).interpolate(em._s(el.dt / 2)))
    return ready(el, nsteps), "c1: 2D eigenmode 40x40 squares x 2 triangles, P1"


def config2(nsteps, n=512, quadrilateral=False, dtype="f64"):
    h = 2.5
    # Courant number 0.05 (default of the reference's tiling harness, tests/tiling/utils.py:51-52):
    # the 0.5 of explosive_source_lf4.py:31 is unstable with the explicit sponge
    el = ExplosiveSourceLF4().setup(Lx=n * h, Ly=n * h, h=h, degree=2, courant_number=0.05,
                                    quadrilateral=quadrilateral, dtype=dtype)
    return ready(el, nsteps), "c2: 2D explosive source %dx%d squares, P2, sponge+source" % (n, n)


def config5(nsteps):
    from seigen_amd.marmousi import cell_material, NX, NY, H
    mesh = RectangleMesh(NX - 1, NY - 1, (NX - 1) * H, (NY - 1) * H)        # seigen/marmousi.py:18-21
    el = ElasticLF4.create(mesh, "DG", 3, dimension=2, solver="explicit", output=False)
    lam, mu, vp = cell_material(el.U)
    el.density, el.l, el.mu = 1.0, lam, mu
    el.dt = cfl_dt(H, float(vp.max()), 0.05)
    # Ricker source near the surface, zero initial state
    sx, sy = 0.5 * (NX - 1) * H, (NY - 1) * H - 24.0
    box = "x[0] >= %r && x[0] <= %r && x[1] >= %r && x[1] <= %r" % (sx - 12.0, sx + 12.0, sy - 12.0, sy + 12.0)
    code = "%s ? (-1.0 + 2*a*pow(t - 0.3, 2))*exp(-a*pow(t - 0.3, 2)) : 0.0" % box
    el.source_expression = Expression(((code, "0.0"), ("0.0", code)), a=A_RICKER, t=0)
    el.source_function = Function(el.S)
    el.source = el.source_expression
    return ready(el, nsteps), "c5: Marmousi %dx%d squares, P3, per-cell lambda/mu" % (NX - 1, NY - 1)


def layer_periodic_state(el, s_scale=3600.0):
    """A smooth non-zero state for large 3-D blocks at the cost of ONE layer of cubes: u_i = sin(k_i . x), s_ij = s_ji =
    s_scale cos(k_(i+j)%3 . x + i - j) with whole wavelengths across the block in x and y and exactly one wavelength per cube
    layer in z - the nodal values of every z layer are those of the first, which is evaluated once and uploaded n_z times
    (a 128^3-cube P4 block has 440 M nodes: evaluating 12 fields there takes the host half a minute)."""
    import numpy as np
    from seigen_amd import _lib
    mesh, blk = el.mesh, el.block
    n, h = mesh.partition.n, mesh.h
    per_layer = n[0] * n[1] * mesh.cells_per_block
    _, X = next(iter(el.U.node_coords_chunks(per_layer * el.U.nd)))       # the first layer of cubes
    assert X.shape[0] == per_layer
    two_pi = 2.0 * np.pi
    k = np.array([[3 * two_pi / (n[0] * h[0]), 2 * two_pi / (n[1] * h[1]), two_pi / h[2]],
                  [2 * two_pi / (n[0] * h[0]), -5 * two_pi / (n[1] * h[1]), two_pi / h[2]],
                  [-4 * two_pi / (n[0] * h[0]), 3 * two_pi / (n[1] * h[1]), -two_pi / h[2]]])
    u = np.stack([np.sin(X @ k[i]) for i in range(3)], axis=-1)
    s = np.zeros(X.shape[:-1] + (3, 3))
    for i in range(3):
        for j in range(i, 3):
            s[..., i, j] = s[..., j, i] = s_scale * np.cos(X @ k[(i + j) % 3] + i - j)
    for layer in range(n[2]):
        blk.set_field_range(_lib.FIELD_U, layer * per_layer, u)
        blk.set_field_range(_lib.FIELD_S, layer * per_layer, s)


def config4_share(nsteps, n=128, degree=4, sponge=False, state="smooth", el=None):
    """`el`: a solver this function returned before (same n, degree) - re-used with the other sponge setting instead of
    allocating the block's 85 GB again (its fields are overwritten by the smooth state)."""
    assert el is None or state == "smooth"
    h = 2.5
    L = n * h
    if el is None:
        mesh = BoxMesh(n, n, n, L, L, L)
        el = ElasticLF4.create(mesh, "DG", degree, dimension=3, solver="explicit", output=False)
        el.density, el.mu, el.l = 1.0, 3600.0, 3599.3664            # explosive_source_lf4.py:21-23
        el.dt = cfl_dt(h, Vp(el.mu, el.l, el.density), 0.05) / 2 ** (degree - 1)   # 2^(P-1) as in eigenmode_3d.py's dt rule
    mesh = el.mesh
    c = 0.5 * L
    box = " && ".join("x[%d] >= %r && x[%d] <= %r" % (a, c - 2 * h, a, c + 2 * h) for a in range(3))
    # the wavelet of explosive_source_lf4.py:37-38, centred inside the steps that are run
    code = "%s ? (-1.0 + 2*a*pow(t - t0, 2))*exp(-a*pow(t - t0, 2)) : 0.0" % box
    z = "0.0"
    el.source_expression = Expression(((code, z, z), (z, code, z), (z, z, code)), a=A_RICKER, t0=0.5 * nsteps * el.dt, t=0)
    el.source_expression.support_box = ((c - 2 * h,) * 3, (c + 2 * h,) * 3)
    el.source_function = Function(el.S)          # zero; the per-step table is what the kernels see
    if sponge:
        from seigen_amd import FunctionSpace
        w = 8 * h       # explosive_source_lf4.py:45: strips 20 units = 8 cells of 2.5 wide; x[1] there is the depth axis, z here
        el.absorption_function = Function(FunctionSpace(mesh, "DG", 4))
        el.absorption = Expression("x[0] <= %r || x[0] >= %r || x[1] <= %r || x[1] >= %r || x[2] <= %r ? 1000 : 0"
                                   % (w, L - w, w, L - w, w))
    else:
        el.absorption_function = None
    el.setup()
    el.upload_source([el.dt * (k + 1) for k in range(nsteps)])
    if state == "smooth":
        layer_periodic_state(el)        # (state = "zero": the fields of a fresh block, as rounds 1-5 measured it)
    return el, "c4s: one rank's share of config 4: %d^3 cubes x 6 tets, P%d, box-Ricker source, %s initial state%s" % (
        n, degree, "smooth non-zero" if state == "smooth" else "zero",
        ", sigma = 1000 in strips 8 cells wide on five faces (explosive_source_lf4.py:42-45)" if sponge else "")


def config3_hex(nsteps, P, N=None):
    """tests/eigenmode/eigenmode_3d.py on UnitCubeMesh(N, N, N, hexahedral=True): the analytic mode as initial state"""
    N = N or {1: 96, 2: 96, 3: 48, 4: 40}[P]
    em = Eigenmode3DLF4(N, P, 0.5 * (1.0 / N) / 2.0 ** (P - 1), output=False, hexahedral=True)
    el = em.elastic
    el.u0.assign(Function(el.U).interpolate(em._u(0)))
    el.s0.assign(Function(el.S).interpolate(em._s(el.dt / 2)))
    return ready(el, nsteps), "c3h%d: 3D eigenmode on %d^3 hexahedra, DQ_%d" % (P, N, P)


def reference_strong_2d(N=256, degree=4, T=2.0, warm_steps=16):
    """One run of the reference's strong-scaling protocol on one device (tests/eigenmode/README.md:7-13:
    `eigenmode_bench.py -- dim=2 explicit=True opt=4 T=2.0 degree=4 N=256`; pybench runs it with `warmups = 1`,
    eigenmode_bench.py:13): Eigenmode2DLF4(N, degree, dt).eigenmode2d(T) through the solver class, dt from the
    Courant rule of eigenmode_bench.py:29-32.  Returns a record: steps, wall seconds of run(T) (everything
    `run` does: set-up, source hand-over, the time loop, the final synchronisation), the 'timestepping' timer
    of the reference's own instrumentation (elastic.py:276), and the L2 error functionals against the analytic
    mode at the time the run ended (the reference's functional compares with t = 5 whatever T was,
    eigenmode_2d.py:41-46: reported as `*_error_vs_t5` for the record only)."""
    from seigen_amd import projected_abs_error_norm
    from seigen_amd.profiling import get_timers
    dt = 0.5 * (1.0 / N) / (2.0 ** (degree - 1))
    warm = Eigenmode2DLF4(N, degree, dt, solver="explicit", output=False)       # warm-up run: same kernels, few steps
    warm.eigenmode2d(T=warm_steps * dt)
    warm.elastic.block.close()
    get_timers(reset=True)
    em = Eigenmode2DLF4(N, degree, dt, solver="explicit", output=False)
    t0 = time.perf_counter()
    u1, s1 = em.eigenmode2d(T=T)
    wall = time.perf_counter() - t0
    el = em.elastic
    steps = len(el.step_times(T))
    timers = {k: v.total for k, v in get_timers(reset=True).items()}
    t_end = steps * dt
    u_err = projected_abs_error_norm(u1, Function(el.U).interpolate(em._u(t_end)), 6)
    s_err = projected_abs_error_norm(s1, Function(el.S).interpolate(em._s(t_end + dt / 2.0)), 6)
    u5, s5 = em.eigenmode_error(u1, s1)
    dofs = el.block.u_dofs + el.block.s_dofs
    rec = {"N": N, "degree": degree, "T": T, "dt": dt, "steps": steps, "dofs": int(dofs), "cells": int(el.block.ncells),
           "run_wall_s": wall, "timestepping_s": timers.get("timestepping"),
           "u_error": float(u_err), "s_error": float(s_err), "u_error_vs_t5": float(u5), "s_error_vs_t5": float(s5)}
    el.block.close()
    return rec
