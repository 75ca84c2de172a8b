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
    for k in ("2d", "3d"):
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
