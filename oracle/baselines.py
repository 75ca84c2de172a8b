"""CPU baselines of bench.py's `cpu_baseline` legs: the oracle's plain-C / OpenMP restatement of the step
(oracle/c/seigen_oracle.c through oracle/cport.py, kind "port") timed on BASELINE's configurations.
ORACLE / TEST INFRASTRUCTURE - see oracle/__init__.py: imported only by bench.py's cpu_baseline leg and by tests.

SURVEY 8d asks for: C1 the full run, C2 20 steps, C5 50 steps, C3 a bounded sample.  Each function builds the case
the way the reference's script does (mesh, constants, sponge, source, initial state), steps it with the C port and
returns a `cpu_baseline` object; meshes are the full-size ones except for C3 (N = 16 of 64: 64^3 x 6 tetrahedra at P4
take minutes per step on a host).  All timed regions are bounded by `budget_s` as well as by their step count."""
import math
import os
import time

import numpy as np

from . import harness, mesh as omesh
from .cport import CPort, build, sponge_blocks

A_RICKER = 159.42       # explosive_source_lf4.py:35


def _ncpu():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _best_threads(cp, one_step, threads=None):
    """fastest of 1, 2, 4 ... threads for `one_step()` (visible CPUs may exceed what a cgroup lets the job use), or
    `threads` if the caller has probed already; returns (threads, {threads: seconds per call})"""
    if threads:
        cp.set_threads(int(threads))
        one_step()      # first touch of the work arrays
        return int(threads), {}
    best, seen, t = (float("inf"), 1), {}, 1
    ncpu = _ncpu()
    one_step()
    while t <= ncpu:
        cp.set_threads(t)
        t0 = time.perf_counter()
        one_step()
        el = time.perf_counter() - t0
        seen[t] = el
        if el < best[0]:
            best = (el, t)
        if el > 4.0 * best[0]:
            break
        t *= 2
    cp.set_threads(best[1])
    return best[1], seen


def _record(dofs, steps, seconds, threads, sample, **extra):
    out = {"value": dofs * steps / seconds / 1e6, "unit": "M DoF-updates/s", "cores": threads, "kind": "port",
           "steps": steps, "seconds": seconds, "sample": sample}
    out.update(extra)
    return out


def _native_build():
    try:
        build(arch="native", force=True)      # rebuild for this host's ISA
    except Exception:      # noqa: BLE001 - a compiler without -march=native support: the portable build
        build(force=True)


def _timed_steps(step, nsteps, budget_s, chunk=1):
    """step(k) advances k steps; run until nsteps are done or the budget is spent -> (steps done, seconds)"""
    t0, done = time.perf_counter(), 0
    while done < nsteps:
        k = min(chunk, nsteps - done)
        step(k)
        done += k
        if time.perf_counter() - t0 > budget_s:
            break
    return done, time.perf_counter() - t0


def _eigenmode2d(N):
    """mesh and closed-form fields of eigenmode_2d.py:7-36 (rho = 1, mu = 0.25, lambda = 0.5) without the numpy
    oracle's global operators (only the C port steps here)"""
    em = harness.Eigenmode2D.__new__(harness.Eigenmode2D)
    em.mesh = omesh.UnitSquareMesh(N, N)
    em.a = math.sqrt(2) * math.pi * math.sqrt(0.25 / 1.0)
    em.b = 2 * math.pi * 0.25
    return em


def config1(budget_s=10.0, threads=None):
    """tests/eigenmode 2-D, 40 x 40, P1, dt = 0.0125, T = 5: the FULL run (400 steps), eigenmode_2d.py:7-36"""
    em, dt = _eigenmode2d(40), 0.0125
    m = em.mesh
    cp = CPort(m, 1)
    X = m.node_coords(1)
    state = [em.u_exact(X, 0.0), em.s_exact(X, dt / 2.0)]

    def step(k):
        state[0], state[1] = cp.step(state[0], state[1], 1.0, dt, 0.5, 0.25, k)
    threads, _ = _best_threads(cp, lambda: step(1), threads)
    state[:] = [em.u_exact(X, 0.0), em.s_exact(X, dt / 2.0)]
    nsteps = int(round(5.0 / dt))
    done, sec = _timed_steps(step, nsteps, budget_s, chunk=50)
    dofs = m.ncells * cp.nd * 6
    return _record(dofs, done, sec, threads, "oracle/c/seigen_oracle.c (plain C + OpenMP): config 1 in full, 2D eigenmode 40x40 P1, "
                   "%d triangles, %d of %d steps (T = 5) in %.2f s, %d threads" % (m.ncells, done, nsteps, sec, threads))


def _box_source(m, P, lo, hi, dt, steps, t0):
    """nodal interpolation of the box indicator (explosive_source_lf4.py:36-40) x Ricker centred at t0"""
    X = m.node_coords(P)
    inb = (X[..., 0] >= lo[0]) & (X[..., 0] <= hi[0]) & (X[..., 1] >= lo[1]) & (X[..., 1] <= hi[1])
    nodes = np.nonzero(inb.reshape(-1))[0]
    vals = np.zeros((steps, len(nodes), 2, 2))
    for k in range(steps):
        vals[k, :, 0, 0] = vals[k, :, 1, 1] = harness.ricker((k + 1) * dt, A_RICKER, t0)
    return nodes, vals


def config2(nsteps=20, budget_s=14.0, n=512, threads=None):
    """2-D explosive source, n x n squares of 2.5 m, P2, DG4 sponge, box source x Ricker, zero initial state
    (explosive_source_lf4.py:17-52; Courant number 0.05 as tests/tiling/utils.py:51-52): `nsteps` steps"""
    h, P, L = 2.5, 2, n * 2.5
    mu, lam, rho = 3600.0, 3599.3664, 1.0
    dt = 0.05 * h / math.sqrt((lam + 2 * mu) / rho)
    m = omesh.RectangleMesh(n, n, L, L)
    cp = CPort(m, P)
    Xs = m.node_coords(4)
    sig = np.where((Xs[..., 0] <= 20.0) | (Xs[..., 0] >= L - 20.0) | (Xs[..., 1] <= 20.0), 1000.0, 0.0)   # :42-45
    sx, sy = 45.0, L - 1.0
    nodes, vals = _box_source(m, P, (sx - 0.5, sy - 0.5), (sx + 0.5, sy + 0.5), dt, nsteps + 8, 0.3)
    cp.set_extra(sponge=sponge_blocks(m, P, sig, 4), src_nodes=nodes, src_values=vals)
    shape_u, shape_s = (m.ncells, cp.nd, 2), (m.ncells, cp.nd, 2, 2)
    u, s = np.zeros(shape_u), np.zeros(shape_s)
    pos = [0]

    def step(k):
        cp.step_ex(u, s, rho, dt, lam, mu, k, step0=pos[0], inplace=True)
        pos[0] += k
    threads, _ = _best_threads(cp, lambda: step(1), threads)
    pos[0] = 0
    u[:], s[:] = 0.0, 0.0
    done, sec = _timed_steps(step, nsteps, budget_s, chunk=4)
    dofs = m.ncells * cp.nd * 6
    return _record(dofs, done, sec, threads, "oracle/c/seigen_oracle.c (plain C + OpenMP): config 2 at full size, 2D explosive source "
                   "%dx%d squares P2 with sponge and source, %d triangles, %d steps in %.2f s, %d threads" % (n, n, m.ncells, done, sec, threads))


def config5(nsteps=50, budget_s=10.0, threads=None):
    """Marmousi 383 x 121 squares of 24 m, P3, per-cell lambda / mu (seigen/marmousi.py:4-24; the material table is
    input DATA read through seigen_amd.marmousi's lookup), box source near the surface: `nsteps` steps"""
    from seigen_amd import FunctionSpace, RectangleMesh
    from seigen_amd.marmousi import NX, NY, H, cell_material
    nx, ny, P = NX - 1, NY - 1, 3
    m = omesh.RectangleMesh(nx, ny, nx * H, ny * H)
    lam, mu, vp = cell_material(FunctionSpace(RectangleMesh(nx, ny, nx * H, ny * H), "DG", P))
    dt = 0.05 * H / float(vp.max())
    sx, sy = 0.5 * nx * H, ny * H - 24.0
    nodes, vals = _box_source(m, P, (sx - 12.0, sy - 12.0), (sx + 12.0, sy + 12.0), dt, nsteps + 8, 0.3)
    cp = CPort(m, P)
    cp.set_extra(lam=lam, mu=mu, src_nodes=nodes, src_values=vals)
    u, s = np.zeros((m.ncells, cp.nd, 2)), np.zeros((m.ncells, cp.nd, 2, 2))
    pos = [0]

    def step(k):
        cp.step_ex(u, s, 1.0, dt, 0.0, 0.0, k, step0=pos[0], inplace=True)
        pos[0] += k
    threads, _ = _best_threads(cp, lambda: step(1), threads)
    pos[0] = 0
    u[:], s[:] = 0.0, 0.0
    done, sec = _timed_steps(step, nsteps, budget_s, chunk=5)
    dofs = m.ncells * cp.nd * 6
    return _record(dofs, done, sec, threads, "oracle/c/seigen_oracle.c (plain C + OpenMP): config 5 at full size, Marmousi %dx%d squares "
                   "P3 per-cell material, %d triangles, %d steps in %.2f s, %d threads" % (nx, ny, m.ncells, done, sec, threads))


def reference_strong_2d(N=256, degree=4, nsteps=6, budget_s=8.0, threads=None):
    """the reference's own benchmark mesh (tests/eigenmode/README.md:7-13: 2-D eigenmode N = 256, P = 4): a bounded
    sample of `nsteps` of its 8192 steps"""
    dt = 0.5 * (1.0 / N) / (2.0 ** (degree - 1))
    em = _eigenmode2d(N)
    m = em.mesh
    cp = CPort(m, degree)
    X = m.node_coords(degree)
    state = [em.u_exact(X, 0.0), em.s_exact(X, dt / 2.0)]

    def step(k):
        state[0], state[1] = cp.step(state[0], state[1], 1.0, dt, 0.5, 0.25, k)
    threads, _ = _best_threads(cp, lambda: step(1), threads)
    done, sec = _timed_steps(step, nsteps, budget_s)
    dofs = m.ncells * cp.nd * 6
    return _record(dofs, done, sec, threads, "oracle/c/seigen_oracle.c (plain C + OpenMP): 2D eigenmode N=%d P=%d, %d triangles, "
                   "%d of the run's %d steps in %.2f s, %d threads" % (N, degree, m.ncells, done, int(round(2.0 / dt)), sec, threads))


def config3(degree=4, budget_s=12.0, N=16):
    """3-D eigenmode (eigenmode_3d.py:7-40) at N = 16 (the bench runs N = 64), same P, FP64: ~budget_s of stepping with
    the fastest thread count + the one-thread figure of SURVEY 8d (about 3 s)"""
    em = harness.Eigenmode3D.__new__(harness.Eigenmode3D)
    em.A = math.sqrt(2 * 1.0 * 0.25)
    em.O = math.pi * math.sqrt(2 * 0.25 / 1.0)
    m = omesh.UnitCubeMesh(N, N, N)
    cp = CPort(m, degree)
    X = m.node_coords(degree)
    dt = 0.5 / N / 2 ** (degree - 1)
    state = [em.u_exact(X, 0.0), em.s_exact(X, dt / 2.0)]

    def step(k):
        state[0], state[1] = cp.step(state[0], state[1], 1.0, dt, 0.5, 0.25, k)
    step(1)           # warm-up
    dofs = m.ncells * cp.nd * 12
    cp.set_threads(1)
    done1, sec1 = _timed_steps(step, 50, 3.0)
    threads, _ = _best_threads(cp, lambda: step(1))
    done, sec = _timed_steps(step, 2000, budget_s, chunk=2)
    return _record(dofs, done, sec, threads, "oracle/c/seigen_oracle.c (plain C + OpenMP) 3D eigenmode N=%d P=%d, %d tets, %d steps in "
                   "%.1f s, %d threads" % (N, degree, m.ncells, done, sec, threads), value_1core=dofs * done1 / sec1 / 1e6)
