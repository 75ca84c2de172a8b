#!/usr/bin/env python
"""Regenerates the committed fixtures under tests/golden/ (run from the repo root in the
build container: `python tests/golden/make_golden.py`).

  ref_c{1,2,3}.txt        every 5th row of the reference's receiver traces
                          /root/reference/tests/explosive_source/REF-C{1,2,3}
                          (data files of the reference's own test; `t ux uy`, uy.py:7-17 reads them)
  explosive_oracle.npz    the oracle's receiver traces for the explosive_source_lf4.py set-up
                          with dt = 0.001 (uy.py:25), T = 2.5
  explosive_oracle_project.npz   the same with the unit-moment source (L2 projection of the source box,
                          oracle.harness.ExplosiveSource(source_mode="project")): the h = 2.5, P2 row of the REF-C
                          convergence study (tools/refc_convergence.py)
  fullspace_oracle.npz    oracle traces of the explosive source moved into the interior (160 m x 100 m, h = 1.25, P3,
                          dt = 0.0005, projected source), compared with the exact 2-D full-space solution
                          (oracle/analytic.py) in tests/test_oracle_pins.py
  halfspace_oracle.npz    oracle (C port) traces of the explosive source 180 m from the nearest sponge, for the comparison
                          with the exact half-space solution (Garvin's problem with buried receivers)
  refc_convergence_hip.npz  NOT made here: receiver traces of the HIP path for the REF-C convergence study, written
                          by tools/refc_convergence.py on a GPU box (gpurun_out/refc_convergence.npz, 'project' rows
                          and the reference's own 'interpolate' h = 2.5 P2 row)
  stage_vectors.npz       F / G / full-step outputs of the oracle for seeded inputs on tiny meshes
                          (numpy.random.default_rng(seed), uniform [-1, 1))
  eigenmode_errors.json   oracle error functionals of the eigenmode sweeps
                          (eigenmode_2d.py:68-84, eigenmode_3d.py:72-88: all rows) and of config 1
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import harness, mesh as omesh  # noqa: E402
from oracle.forms import ElasticOperators  # noqa: E402
from oracle.lf4 import OracleLF4  # noqa: E402

REF = "/root/reference/tests/explosive_source"


def ref_traces():
    for i in (1, 2, 3):
        rows = np.loadtxt(os.path.join(REF, "REF-C%d" % i))
        np.savetxt(os.path.join(HERE, "ref_c%d.txt" % i), rows[4::5], fmt="%.9e")


def explosive():
    ex = harness.ExplosiveSource()
    ex.elastic.dt = 0.001
    times, tr = ex.run(2.5)
    np.savez_compressed(os.path.join(HERE, "explosive_oracle.npz"), times=times[4::5], traces=tr[4::5])


def explosive_project():
    ex = harness.ExplosiveSource(source_mode="project")
    ex.elastic.dt = 0.001
    times, tr = ex.run(2.5)
    np.savez_compressed(os.path.join(HERE, "explosive_oracle_project.npz"), times=times[4::5], traces=tr[4::5])


FULLSPACE = dict(Lx=160.0, Ly=100.0, h=1.25, degree=3, src=(80.0, 55.0), dt=0.0005, T=0.75,
                 receivers=((105.0, 55.0), (80.0, 30.0), (98.0, 73.0)))


def fullspace():
    c = FULLSPACE
    ex = harness.ExplosiveSource(Lx=c["Lx"], Ly=c["Ly"], h=c["h"], degree=c["degree"], src=c["src"], source_mode="project")
    ex.elastic.dt = c["dt"]
    times, tr = ex.run(c["T"], receivers=c["receivers"])
    np.savez_compressed(os.path.join(HERE, "fullspace_oracle.npz"), times=times[9::10], traces=tr[9::10],
                        receivers=np.array(c["receivers"]), src=np.array(c["src"]), Vp=ex.Vp)


HALFSPACE = dict(Lx=500.0, Ly=150.0, h=1.25, degree=3, src=(200.0, 149.0), dt=0.0005, T=2.5, every=10,
                 receivers=((245.3, 149.0), (245.3, 149.7), (295.3, 149.0), (295.3, 147.7)))


def halfspace():
    """The explosive-source set-up with the source 180 m from the nearest sponge (whose abrupt onset reflects), unit-moment
    projected source, stepped by the oracle's C port (sponge + source: oracle/cport.py, validated against the numpy
    oracle in tests/test_oracle_cport.py); receiver traces every 10th step for the comparison with the exact
    half-space solution (oracle/analytic.py explosive_line_source_halfspace) in tests/test_oracle_pins.py."""
    from oracle.cport import CPort, sponge_blocks
    c = HALFSPACE
    ex = harness.ExplosiveSource(Lx=c["Lx"], Ly=c["Ly"], h=c["h"], degree=c["degree"], src=c["src"], source_mode="project")
    el = ex.elastic
    m, P = ex.mesh, c["degree"]
    cp = CPort(m, P)
    nsteps = int(round(c["T"] / c["dt"]))
    nodes = np.nonzero(np.abs(ex.pattern).reshape(m.ncells * cp.nd, -1).max(axis=1) > 0)[0]
    pat = ex.pattern.reshape(m.ncells * cp.nd, 2, 2)[nodes]
    w = np.array([harness.ricker((k + 1) * c["dt"]) for k in range(nsteps)])
    cp.set_extra(sponge=sponge_blocks(m, P, ex.sigma, 4), src_nodes=nodes, src_values=w[:, None, None, None] * pat[None])
    ev = [ex.point_evaluator(x, y) for (x, y) in c["receivers"]]
    u = np.zeros((m.ncells, cp.nd, 2))
    s = np.zeros((m.ncells, cp.nd, 2, 2))
    times, traces = [], []
    for k0 in range(0, nsteps, c["every"]):
        u, s = cp.step_ex(u, s, el.density, c["dt"], el.l, el.mu, c["every"], step0=k0, inplace=True)
        times.append((k0 + c["every"]) * c["dt"])
        traces.append([[float(phi @ u[cc, :, k]) for k in range(2)] for (cc, phi) in ev])
        if k0 % 1000 == 0:
            print("halfspace step", k0, flush=True)
    np.savez_compressed(os.path.join(HERE, "halfspace_oracle.npz"), times=np.array(times), traces=np.array(traces),
                        receivers=np.array(c["receivers"]), src=np.array(c["src"]), Vp=ex.Vp, Vs=ex.Vs)


STAGE_CASES = [
    (1, 2, (6,), (1.5,), "left"),
    (2, 1, (4, 4), (1.0, 1.0), "left"),
    (2, 2, (4, 4), (1.0, 1.0), "left"),
    (2, 4, (4, 4), (1.0, 1.0), "right"),
    (3, 1, (2, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 3, (2, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 4, (2, 2, 2), (1.0, 1.0, 1.0), "left"),
]


def stage_vectors():
    out = {}
    for ci, (dim, P, n, L, diag) in enumerate(STAGE_CASES):
        m = omesh.structured(dim, n, L, diag)
        E = ElasticOperators(m, P)
        nd = E.nd
        rng = np.random.default_rng(ci)
        u = rng.uniform(-1, 1, (m.ncells, nd, dim))
        T = rng.uniform(-1, 1, (m.ncells, nd, dim, dim))
        key = "c%d" % ci
        out[key + "_meta"] = np.array([dim, P] + list(n))
        out[key + "_L"] = np.array(L)
        out[key + "_u"] = u
        out[key + "_T"] = T
        out[key + "_F"] = E.apply_F(T, u)
        out[key + "_G"] = E.apply_G(u, 0.7, 0.3)
        orc = OracleLF4(m, P)
        orc.u0, orc.s0 = u.copy(), T.copy()
        orc.dt = 0.05 * min(L[a] / n[a] for a in range(dim)) / P ** 2
        orc.l, orc.mu, orc.density = 0.5, 0.25, 1.0
        for k in range(10):
            orc.step((k + 1) * orc.dt)
            if k == 0:
                out[key + "_u_step1"], out[key + "_s_step1"] = orc.u1.copy(), orc.s1.copy()
        out[key + "_dt"] = np.array(orc.dt)
        out[key + "_u_step10"], out[key + "_s_step10"] = orc.u1.copy(), orc.s1.copy()
    np.savez_compressed(os.path.join(HERE, "stage_vectors.npz"), **out)


def eigenmode_errors():
    """The reference's full sweeps: 2-D P1..4 x N in {4, 8, 16, 32} (eigenmode_2d.py:68-84), 3-D P1..3 x
    N in {2, 4, 8} (eigenmode_3d.py:72-88), T = 5, and config 1.  Rows already in the file are kept
    (the numpy oracle is deterministic; this only saves the minutes they take)."""
    path = os.path.join(HERE, "eigenmode_errors.json")
    res = json.load(open(path)) if os.path.exists(path) else {"2d": [], "3d": []}
    have = {(k, r["P"], r["N"]) for k in ("2d", "3d") for r in res[k]}
    for P in (1, 2, 3, 4):
        for N in (4, 8, 16, 32):
            if ("2d", P, N) in have:
                continue
            dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
            em = harness.Eigenmode2D(N, P, dt)
            u1, s1 = em.run()
            e = em.errors(u1, s1)
            res["2d"].append(dict(P=P, N=N, dt=dt, **e))
            print("2d", P, N, e, flush=True)
    if "config1" not in res:
        em = harness.Eigenmode2D(40, 1, 0.0125)
        u1, s1 = em.run()
        res["config1"] = dict(P=1, N=40, dt=0.0125, **em.errors(u1, s1))
    for P in (1, 2, 3):
        for N in (2, 4, 8):
            if ("3d", P, N) in have:
                continue
            dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
            em = harness.Eigenmode3D(N, P, dt)
            u1, s1 = em.run()
            e = em.errors(u1, s1)
            res["3d"].append(dict(P=P, N=N, dt=dt, **e))
            print("3d", P, N, e, flush=True)
    # the 2-D sweep on UnitSquareMesh(N, N, quadrilateral=True) (tensor-product element; build-defined row)
    res.setdefault("2d_quadrilateral", [])
    haveq = {(r["P"], r["N"]) for r in res["2d_quadrilateral"]}
    for P in (1, 2, 3, 4):
        for N in (4, 8, 16):
            if (P, N) in haveq:
                continue
            dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
            em = harness.Eigenmode2D(N, P, dt, quadrilateral=True)
            u1, s1 = em.run()
            e = em.errors(u1, s1)
            res["2d_quadrilateral"].append(dict(P=P, N=N, dt=dt, **e))
            print("2d quadrilateral", P, N, e, flush=True)
    for k in ("2d", "3d", "2d_quadrilateral"):
        res[k].sort(key=lambda r: (r["P"], r["N"]))
    json.dump(res, open(path, "w"), indent=1)


if __name__ == "__main__":
    what = sys.argv[1:] or ["ref", "stage", "eigen", "explosive"]
    if "ref" in what:
        ref_traces()
    if "stage" in what:
        stage_vectors()
    if "eigen" in what:
        eigenmode_errors()
    if "explosive" in what:
        explosive()
    if "explosive_project" in what:
        explosive_project()
    if "fullspace" in what:
        fullspace()
    if "halfspace" in what:
        halfspace()
