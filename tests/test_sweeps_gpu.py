"""The reference's own verification sweeps on the HIP path, and its solver-string switch.

* ``convergence_analysis`` of tests/eigenmode/eigenmode_2d.py:68-84 (P1..4 x N in {4, 8, 16, 32}) and
  eigenmode_3d.py:72-88 (P1..3 x N in {2, 4, 8}): dt = 0.5 (1/N) / 2^(P-1), T = 5, error functional
  ||Pi_DG6 |e||| (2-D) / ||Pi_DG3 |e||| (3-D) - through ``Eigenmode2DLF4`` / ``Eigenmode3DLF4`` of
  seigen_amd.harness, compared with the oracle's committed values
  (tests/golden/eigenmode_errors.json, tests/golden/make_golden.py) to 1e-9 absolute
  (north star: within 1e-6 of the reference), plus the observed orders of convergence.
* ``ElasticLF4.create(..., solver=...)`` (seigen/elastic.py:50-64): 'parloop' / 'fusion' /
  'tiling' run the same arithmetic as 'explicit' (bitwise here), 'implicit' runs it with the implicit
  forms' density convention, an unknown string raises ValueError.
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eigenmode_errors.json")


def _quiet():
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.eigenmode as he
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None
    he.log = lambda s: None


def _check_orders(rows, key, P, expect):
    """observed order between the two finest meshes of degree P"""
    r = sorted([x for x in rows if x[0] == P], key=lambda x: -x[1])      # decreasing h
    e1, e2 = r[-2][key], r[-1][key]
    order = math.log(e1 / e2) / math.log(r[-2][1] / r[-1][1])
    assert order > expect, (P, key, order)
    return order


def test_convergence_analysis_2d_full_sweep(gpu):
    _quiet()
    from seigen_amd.harness.eigenmode import convergence_analysis_2d
    gold = {(r["P"], r["N"]): r for r in json.load(open(GOLD))["2d"]}
    assert len(gold) == 16, "tests/golden/eigenmode_errors.json must hold the full 2-D sweep"
    rows = convergence_analysis_2d()                 # degrees 1..4, N = 4, 8, 16, 32 as the reference
    assert len(rows) == 16
    for (d, h, dt, u_error, s_error) in rows:
        g = gold[(d, int(round(1.0 / h)))]
        assert dt == g["dt"]
        assert abs(u_error - g["u_error"]) < 1e-9 and abs(s_error - g["s_error"]) < 1e-9, (d, h, u_error, s_error, g)
    # velocity converges at about order P+1 (P = 4 at N = 32 is already at the 1e-9 time-stepping floor), stress at P
    for P in (1, 2, 3):
        _check_orders(rows, 3, P, P + 0.6)
        _check_orders(rows, 4, P, P - 0.3)


def test_convergence_analysis_3d_full_sweep(gpu):
    _quiet()
    from seigen_amd.harness.eigenmode import convergence_analysis_3d
    gold = {(r["P"], r["N"]): r for r in json.load(open(GOLD))["3d"]}
    assert len(gold) == 9, "tests/golden/eigenmode_errors.json must hold the full 3-D sweep"
    rows = convergence_analysis_3d()                 # degrees 1..3, N = 2, 4, 8
    assert len(rows) == 9
    for (d, h, dt, u_error, s_error) in rows:
        g = gold[(d, int(round(1.0 / h)))]
        assert abs(u_error - g["u_error"]) < 1e-9 and abs(s_error - g["s_error"]) < 1e-9, (d, h, u_error, s_error, g)
    for P in (1, 2, 3):
        _check_orders(rows, 4, P, P - 0.5)


@pytest.mark.parametrize("solver", ["parloop", "fusion", "tiling", "hip"])
def test_solver_strings_share_the_explicit_path(gpu, solver):
    _quiet()
    from seigen_amd import ExplicitElasticLF4, TilingElasticLF4
    from seigen_amd.harness.eigenmode import Eigenmode2DLF4
    N, P = 8, 2
    dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
    base = Eigenmode2DLF4(N, P, dt, solver="explicit", output=False)
    assert type(base.elastic) is ExplicitElasticLF4
    ub, sb = base.eigenmode2d(T=1.0)
    em = Eigenmode2DLF4(N, P, dt, solver=solver, output=False)
    if solver != "hip":
        assert isinstance(em.elastic, TilingElasticLF4)
        assert em.elastic.tiling_mode == {"parloop": None, "fusion": "hard", "tiling": "tile"}[solver]
        assert em.elastic.calculate_sdepth(8, em.elastic.num_unroll, 0) == 1          # one rank
    u1, s1 = em.eigenmode2d(T=1.0)
    np.testing.assert_array_equal(u1.dat.data, ub.dat.data)
    np.testing.assert_array_equal(s1.dat.data, sb.dat.data)


def test_implicit_solver_is_the_explicit_path_with_the_implicit_density_convention(gpu):
    """seigen/elastic.py:318-332: every KSP system is a block-diagonal DG mass matrix, so the implicit class runs the
    explicit arithmetic; only form_u1 differs (:175-178 against :341-345).  rho = 1: bitwise the explicit result;
    rho = 2: the oracle's LF4 with the density on the left-hand side, and NOT the explicit class's result."""
    _quiet()
    from oracle import mesh as omesh
    from oracle.lf4 import OracleLF4
    from seigen_amd import ElasticLF4, Function, ImplicitElasticLF4, UnitSquareMesh
    from seigen_amd.harness.eigenmode import Eigenmode2DLF4
    N, P = 8, 2
    dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
    base = Eigenmode2DLF4(N, P, dt, solver="explicit", output=False)
    ub, sb = base.eigenmode2d(T=1.0)
    em = Eigenmode2DLF4(N, P, dt, solver="implicit", output=False)
    assert isinstance(em.elastic, ImplicitElasticLF4) and em.elastic.density_physical
    u1, s1 = em.eigenmode2d(T=1.0)
    np.testing.assert_array_equal(u1.dat.data, ub.dat.data)
    np.testing.assert_array_equal(s1.dat.data, sb.dat.data)

    rng = np.random.default_rng(5)
    res = {}
    for solver in ("implicit", "explicit"):
        el = ElasticLF4.create(UnitSquareMesh(N, N), "DG", P, dimension=2, solver=solver, output=False)
        el.density, el.l, el.mu, el.dt = 2.0, 0.5, 0.25, dt
        if solver == "implicit":
            u0 = rng.uniform(-1, 1, (el.U.ncells, el.U.nd, 2))
            s0 = rng.uniform(-1, 1, (el.S.ncells, el.S.nd, 2, 2))
            s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        el.u0.assign(Function(el.U).assign(u0))
        el.s0.assign(Function(el.S).assign(s0))
        u, s = el.run(10 * dt * (1 + 1e-9))
        res[solver] = (u.dat.data_cells.copy(), s.dat.data_cells.copy())
    orc = OracleLF4(omesh.structured(2, (N, N), (1.0, 1.0)), P)
    orc.dt, orc.l, orc.mu, orc.density, orc.density_physical = dt, 0.5, 0.25, 2.0, True
    orc.u0, orc.s0 = u0.copy(), s0.copy()
    for k in range(10):
        orc.step((k + 1) * dt)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(res["implicit"][0], orc.u1) < 1e-12 and rel(res["implicit"][1], orc.s1) < 1e-12
    assert rel(res["explicit"][0], orc.u1) > 1e-2
