"""Pins of the CPU oracle (oracle/): everything the reference itself offers to check the path
against - analytic eigenmodes (tests/eigenmode/eigenmode_2d.py:30-47, eigenmode_3d.py:30-51),
the receiver traces tests/explosive_source/REF-C1..3 - plus implementation-independent
known-answer properties of the weak form seigen/elastic.py:204-219.

The Firedrake stack cannot be installed here and the reference stores no output of its
own, so bit-level parity with Firedrake is "parity unpinned" (oracle/__init__.py)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import harness, mesh as omesh, refelem
from oracle.forms import ElasticOperators
from oracle.lf4 import OracleLF4, count_steps

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------------------ reference element
@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_lagrange_basis_is_nodal_and_partition_of_unity(dim, P):
    xi = refelem.node_ref_coords(dim, P)
    phi, dphi = refelem.tabulate(dim, P, xi)
    np.testing.assert_allclose(phi, np.eye(len(xi)), atol=1e-13)
    rng = np.random.default_rng(0)
    pts = rng.dirichlet(np.ones(dim + 1), size=10)[:, 1:]
    phi, dphi = refelem.tabulate(dim, P, pts)
    np.testing.assert_allclose(phi.sum(axis=1), 1.0, atol=1e-13)
    np.testing.assert_allclose(dphi.sum(axis=1), 0.0, atol=1e-12)
    # reproduces x^P: sum_a x_a^P phi_a(x) = x^P
    np.testing.assert_allclose(phi @ (xi[:, 0] ** P), pts[:, 0] ** P, atol=1e-12)


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_quadrature_exactness(dim):
    for deg in range(0, 9):
        x, w = refelem.simplex_quadrature(dim, deg)
        # int x_1^deg over the unit simplex = deg! / (deg + dim)!
        exact = math.factorial(deg) / math.factorial(deg + dim)
        assert abs(np.dot(w, x[:, 0] ** deg) - exact) < 1e-14


# ------------------------------------------------------------------------------ weak form KATs
@pytest.mark.parametrize("dim,n,P", [(1, (5,), 2), (2, (3, 3), 1), (2, (3, 2), 3), (3, (2, 2, 2), 2)])
def test_polynomial_reproduction(dim, n, P):
    """For a globally polynomial field of degree <= P the central flux is consistent, so
    Minv f / Minv g return the exact derivatives in every cell that has no boundary facet, and
    g (whose ds terms use the own trace, elastic.py:214-216) returns them in every cell."""
    m = omesh.structured(dim, n, tuple(1.0 + 0.5 * a for a in range(dim)))
    E = ElasticOperators(m, P)
    X = m.node_coords(P)
    rng = np.random.default_rng(3)
    # u_i = c_i . x^P-type polynomial: use (a_i . x)^P
    A = rng.uniform(-1, 1, (dim, dim))
    u = np.stack([(X @ A[i]) ** P for i in range(dim)], axis=-1)
    grad = np.stack([np.stack([P * (X @ A[i]) ** (P - 1) * A[i, k] for k in range(dim)], axis=-1)
                     for i in range(dim)], axis=-2)          # [c, a, i, k]
    lam, mu = 0.7, 0.3
    sh = E.apply_G(u, lam, mu)
    tr = np.einsum('cakk->ca', grad)
    exp = mu * (grad + np.swapaxes(grad, -1, -2))
    for i in range(dim):
        exp[..., i, i] += lam * tr
    np.testing.assert_allclose(sh, exp, atol=1e-9)
    # f: divergence of a polynomial tensor, interior cells only (T.n = 0 is imposed on the boundary)
    T = np.zeros(X.shape[:-1] + (dim, dim))
    B = rng.uniform(-1, 1, (dim, dim, dim))
    div = np.zeros(X.shape[:-1] + (dim,))
    for i in range(dim):
        for j in range(dim):
            T[..., i, j] = (X @ B[i, j]) ** P
            div[..., i] += P * (X @ B[i, j]) ** (P - 1) * B[i, j, j]
    uh = E.apply_F(T)
    boundary_cells = set(m.exterior_facets[:, 0].tolist())
    interior = [c for c in range(m.ncells) if c not in boundary_cells]
    if interior:
        np.testing.assert_allclose(uh[interior], div[interior], atol=1e-9)


def test_operators_are_negative_adjoint():
    """Central flux + traction-free boundary: <F T, u>_M = - <T, (G u)/(lam,mu -> strain)>_M, i.e. the
    semi-discrete system conserves 1/2 (rho |u|^2 + compliance energy)."""
    m = omesh.UnitSquareMesh(3, 3)
    P = 2
    E = ElasticOperators(m, P)
    rng = np.random.default_rng(5)
    u = rng.uniform(-1, 1, (m.ncells, E.nd, 2))
    T = rng.uniform(-1, 1, (m.ncells, E.nd, 2, 2))
    T = T + np.swapaxes(T, -1, -2)
    M = E.ops.M
    Fu = E.apply_F(T).reshape(E.ops.N, 2)
    lhs = sum(u.reshape(E.ops.N, 2)[:, i] @ (M @ Fu[:, i]) for i in range(2))
    # strain rate of u through g with lam=0, mu=1/2: eps = sym grad u
    eps = E.apply_G(u, 0.0, 0.5).reshape(E.ops.N, 2, 2)
    Tf = T.reshape(E.ops.N, 2, 2)
    rhs = sum(Tf[:, i, j] @ (M @ eps[:, i, j]) for i in range(2) for j in range(2))
    assert abs(lhs + rhs) < 1e-11 * max(1.0, abs(lhs))


def test_energy_is_conserved_by_lf4():
    em = harness.Eigenmode2D(8, 2, 0.25 / 8)
    el = em.elastic
    X = el.node_coords()
    el.u0 = em.u_exact(X, 0.0)
    el.s0 = em.s_exact(X, el.dt / 2)
    M = el.E.ops.M
    lam, mu = el.l, el.mu

    def energy():
        u = el.u0.reshape(-1, 2)
        s = el.s0.reshape(-1, 2, 2)
        ek = 0.5 * sum(u[:, i] @ (M @ u[:, i]) for i in range(2))
        tr = s[:, 0, 0] + s[:, 1, 1]
        # compliance: eps = (s - lam/(2(lam+mu)) tr I) / (2 mu)   (2-D plane strain)
        es = 0.0
        for i in range(2):
            for j in range(2):
                eij = (s[:, i, j] - (lam / (2 * (lam + mu)) * tr if i == j else 0.0)) / (2 * mu)
                es += 0.5 * (s[:, i, j] @ (M @ eij))
        return ek + es

    # u lives at t and s at t + dt/2, so this staggered energy oscillates with the mode
    # (period 2 pi / a = 90.5 steps) by O(dt); it must not drift from period to period
    e = []
    for k in range(362):
        el.step((k + 1) * el.dt)
        e.append(energy())
    e = np.array(e)
    per = [e[i * 181:(i + 1) * 181].mean() for i in range(2)]
    assert abs(per[1] - per[0]) / per[0] < 1e-4
    assert e.std() / e.mean() < 5e-2


# ------------------------------------------------------------------------------ eigenmode pins
def test_eigenmode_2d_convergence_orders():
    """Analytic solution of eigenmode_2d.py:30-47: observed orders ~P+1 for u (super-convergent at
    P1: ~2) and ~P for the stress, central flux."""
    g = json.load(open(os.path.join(GOLD, "eigenmode_errors.json")))
    rows = {(r["P"], r["N"]): r for r in g["2d"]}
    for P in (1, 2, 3):
        ou = math.log2(rows[(P, 8)]["u_l2"] / rows[(P, 16)]["u_l2"])
        os_ = math.log2(rows[(P, 8)]["s_l2"] / rows[(P, 16)]["s_l2"])
        assert ou > P + 0.5 - (0.1 if P > 1 else 0.7), (P, ou)
        assert os_ > P - 0.25, (P, os_)
    # a live re-run of one point reproduces the committed number
    em = harness.Eigenmode2D(8, 1, 0.5 / 8)
    u1, s1 = em.run()
    e = em.errors(u1, s1)
    assert em.elastic.nsteps == 80
    assert abs(e["u_l2"] - rows[(1, 8)]["u_l2"]) < 1e-12
    assert abs(e["u_error"] - rows[(1, 8)]["u_error"]) < 1e-12
    # values of the survey's throw-away restatement, independent code
    assert abs(e["u_l2"] - 1.3748e-1) < 1e-5 and abs(e["s_l2"] - 2.0775e-1) < 1e-5


def test_eigenmode_diagonal_direction_does_not_matter():
    """[upstream] the diagonal of Firedrake's split is an assumption; the eigenmode is
    symmetric under x -> 1-x so both choices give the same errors."""
    res = []
    for diag in ("left", "right"):
        em = harness.Eigenmode2D(8, 1, 0.5 / 8, diag)
        u1, s1 = em.run()
        res.append(em.errors(u1, s1))
    assert abs(res[0]["u_l2"] - res[1]["u_l2"]) < 1e-10
    assert abs(res[0]["s_l2"] - res[1]["s_l2"]) < 1e-10
    # the reference's functional || Pi_DG6 |e| || integrates the non-polynomial |e| with a
    # quadrature rule that is not symmetric under the mirror map, so it moves in the 4th digit
    assert abs(res[0]["s_error"] - res[1]["s_error"]) < 2e-3 * res[0]["s_error"]
    assert abs(res[0]["u_error"] - res[1]["u_error"]) < 2e-3 * res[0]["u_error"]


def test_eigenmode_3d_converges():
    g = json.load(open(os.path.join(GOLD, "eigenmode_errors.json")))
    rows = {(r["P"], r["N"]): r for r in g["3d"]}
    for P in (2, 3):
        assert math.log2(rows[(P, 2)]["u_l2"] / rows[(P, 4)]["u_l2"]) > P
    em = harness.Eigenmode3D(2, 1, 0.25)
    u1, s1 = em.run()
    e = em.errors(u1, s1)
    assert abs(e["u_error"] - rows[(1, 2)]["u_error"]) < 1e-12


def test_step_count():
    assert count_steps(0.0125, 5.0) == 400          # BASELINE config 1
    assert count_steps(0.001, 2.5) == 2500


# ------------------------------------------------------------------------------ REF-C pins
def _load_traces():
    d = np.load(os.path.join(GOLD, "explosive_oracle.npz"))
    refs = [np.loadtxt(os.path.join(GOLD, "ref_c%d.txt" % i)) for i in (1, 2, 3)]
    return d["times"], d["traces"], refs


def _lsq_ratio(ours, ref, w):
    return float(np.dot(ours[w], ref[w]) / np.dot(ref[w], ref[w]))


UY_WINDOWS = ((0.0, 1.0), (0.5, 1.5), (1.0, 2.5))       # the plot ranges of uy.py:52,65,78


def _refc_metrics(tr, refs, times, comp=1, sign=-1.0):
    """(least-squares amplitude ratio, correlation) per receiver of sign*u_comp against REF-C's column comp + 1"""
    out = []
    for i in range(3):
        w = (times > UY_WINDOWS[i][0]) & (times < UY_WINDOWS[i][1] - 1e-9)
        o, r = sign * tr[w, i, comp], refs[i][w, 1 + comp]
        if not np.any(r):                       # REF-C1 holds no x-motion
            out.append((float("nan"), float("nan")))
            continue
        out.append((float(np.dot(o, r) / np.dot(r, r)), float(np.corrcoef(o, r)[0, 1])))
    return out


def test_ref_c_with_the_reference_set_up_is_characterised_not_matched():
    """tests/explosive_source/REF-C1..3 (external code, compared by eye in uy.py:45-80) against the oracle's
    run of explosive_source_lf4.py as the reference sets it up (nodal interpolation of the source box, h = 2.5,
    P2; dt = 0.001 of uy.py:25).  These numbers CHARACTERISE that run; they are not a match:

      * wave forms and arrival times agree (correlation 0.97-0.99),
      * amplitudes are C1 1.05, C2 2.24, C3 2.18 times REF-C - and all three scale with the integral of the
        nodally interpolated source (2.083 m^2 on this mesh instead of the box's 1 m^2; 0.52 m^2 and ratios
        0.27 / 0.79 / 0.78 on the h = 1.25 mesh): the near-1 value at C1 is an accident of this mesh (round 2
        asserted it as a pin; it is none).

    The mesh-independent comparison is test_ref_c_unit_moment_convergence below."""
    times, tr, refs = _load_traces()
    for r in refs:
        np.testing.assert_allclose(r[:, 0], times, atol=1e-9)
    m = _refc_metrics(tr, refs, times)
    for (ratio, corr), want in zip(m, (1.046, 2.235, 2.175)):
        assert abs(ratio / want - 1.0) < 0.01 and corr > 0.97, m
    # integral of the interpolated source per unit amplitude, from the oracle's own mass matrix
    ex = harness.ExplosiveSource()
    assert abs(ex.source_integral - 2.0 * 2.5 ** 2 / 2.0 / 3.0) < 1e-12 and int(ex.src_mask.sum()) == 2
    # the same run on the h = 1.25 mesh (HIP path, tools/refc_convergence.py): the source integral drops to
    # 0.52 m^2 and every receiver follows it - refinement makes the "match" at C1 four times worse
    h = np.load(os.path.join(GOLD, "refc_convergence_hip.npz"))
    m2 = _refc_metrics(h["interpolate_h1.25_P2"], refs, times)
    assert abs(m2[0][0] / 0.266 - 1) < 0.02 and abs(m2[1][0] / 0.786 - 1) < 0.02 and abs(m2[2][0] / 0.779 - 1) < 0.02
    # C1 records no x-motion in the reference (source x-position); ours is small
    w = times < 0.6
    assert np.abs(tr[w, 0, 0]).max() < 0.02 * np.abs(tr[:, 0, 1]).max()
    # quiet before the first arrival, as in the reference
    assert np.abs(tr[times < 0.9, 2, 1]).max() < 1e-3 * np.abs(tr[:, 2, 1]).max()


def test_ref_c_unit_moment_convergence():
    """REF-C against a source of UNIT MOMENT on every mesh (the L2 projection of the source box; build-defined
    `source_mode='project'`), h in {2.5, 1.25, 0.625} x P in {2, 3, 4} (tools/refc_convergence.py on the HIP
    path, profiles/r03/refc_convergence.txt; the h = 2.5 / P2 row is also run by the oracle and must agree).

    Result: REF-C is NOT reproduced in amplitude.  The far field is mesh-converged to 0.1 % and sits at
    1.271 x REF-C2 and 1.260 x REF-C3 in uy (correlation 0.993), 0.50 x / 0.56 x in ux (correlation 0.75 / 0.85);
    C1, inside the source box, stays at 0.46-0.49.  The same code reproduces the exact full-space solution to
    0.4 % (test_fullspace_analytic_pin), so the factor belongs to REF-C's generator, whose set-up the reference
    does not record (REF-C's horizontal-to-vertical ratio is that of receivers AT the free
    surface, not 1 m below it)."""
    refs = [np.loadtxt(os.path.join(GOLD, "ref_c%d.txt" % i)) for i in (1, 2, 3)]
    times = refs[0][:, 0]
    h = np.load(os.path.join(GOLD, "refc_convergence_hip.npz"))
    # the oracle's own run of the coarsest row = the HIP path's
    o = np.load(os.path.join(GOLD, "explosive_oracle_project.npz"))
    np.testing.assert_allclose(o["times"], times, atol=1e-9)
    assert np.abs(o["traces"] - h["project_h2.5_P2"]).max() < 1e-9 * np.abs(o["traces"]).max()
    rows = {}
    for hh in (2.5, 1.25, 0.625):
        for P in (2, 3, 4):
            rows[(hh, P)] = _refc_metrics(h["project_h%g_P%d" % (hh, P)], refs, times)
    fine = [rows[k] for k in ((1.25, 3), (1.25, 4), (0.625, 2), (0.625, 3), (0.625, 4))]
    for i, want in ((1, 1.2707), (2, 1.2596)):
        vals = np.array([m[i][0] for m in fine])
        assert vals.max() / vals.min() - 1.0 < 2e-3, vals            # mesh-converged
        assert abs(vals.mean() / want - 1.0) < 0.05, vals            # ... to this, not to 1
        assert min(m[i][1] for m in fine) > 0.99
    # even the coarsest mesh is within 6 % of the converged far field once the source has the right moment
    assert abs(rows[(2.5, 2)][1][0] / 1.2707 - 1.0) < 0.06 and abs(rows[(2.5, 2)][2][0] / 1.2596 - 1.0) < 0.08
    # C1 sits inside the 1 m source box: it sees the projected indicator itself, which converges slowly
    c1 = np.array([m[0][0] for m in fine])
    assert 0.40 < c1.min() and c1.max() < 0.52
    # horizontal component: half of REF-C's, poorly correlated - REF-C's receivers are not where uy.py probes
    ux = _refc_metrics(h["project_h0.625_P4"], refs, times, comp=0, sign=1.0)
    assert abs(ux[1][0] - 0.50) < 0.03 and abs(ux[2][0] - 0.56) < 0.03 and ux[1][1] < 0.8


def test_fullspace_analytic_pin():
    """Source normalisation and P-wave propagation against theory: the explosive source moved into the interior
    of a 160 m x 100 m domain (h = 1.25, P3, projected unit-moment source), oracle traces (fixture
    fullspace_oracle.npz, make_golden.py) against the exact 2-D full-space solution of an explosive line source
    (oracle/analytic.py) before any reflection arrives.  Amplitude within 0.5 %, misfit 0.5 % at the receiver
    inside a cell; a receiver ON a mesh line (where DG fields are two-valued and least accurate) 3 % / 7 %."""
    from oracle.analytic import explosive_line_source_2d
    d = np.load(os.path.join(GOLD, "fullspace_oracle.npz"))
    t, tr, src, Vp = d["times"], d["traces"], d["src"], float(d["Vp"])
    res = []
    for i in (0, 2):
        x, y = d["receivers"][i]
        dx, dy = x - src[0], y - src[1]
        r = float(np.hypot(dx, dy))
        vr = explosive_line_source_2d(r, t, Vp)
        ours = tr[:, i, 0] * dx / r + tr[:, i, 1] * dy / r
        res.append((np.dot(ours, vr) / np.dot(vr, vr), np.linalg.norm(ours - vr) / np.linalg.norm(vr)))
    assert abs(res[1][0] - 1.0) < 0.005 and res[1][1] < 0.006, res
    assert abs(res[0][0] - 1.0) < 0.035 and res[0][1] < 0.08, res
    # the fixture is current: first 20 steps live
    from tests.golden.make_golden import FULLSPACE as c
    ex = harness.ExplosiveSource(Lx=c["Lx"], Ly=c["Ly"], h=c["h"], degree=c["degree"], src=c["src"], source_mode="project")
    ex.elastic.dt = c["dt"]
    tl, live = ex.run(20 * c["dt"], receivers=c["receivers"])
    np.testing.assert_allclose(tl[9::10], t[:2], atol=1e-12)
    assert np.abs(live[9::10] - tr[:2]).max() <= 1e-9 * np.abs(tr[:2]).max() + 1e-300


def test_halfspace_analytic_pin():
    """The exact solution of the reference's explosive-source problem - an explosive line source 1 m below the free
    surface of a half space, receivers below the surface (oracle/analytic.py explosive_line_source_halfspace) - against
    the oracle (C port with sponge and source; fixture halfspace_oracle.npz, make_golden.py: h = 1.25, P3, unit-moment
    source, the source 180 m from the nearest sponge): direct P, reflected P, converted SV and the Rayleigh wave, both
    components, within 0.1 % in amplitude and 0.2 % in relative L2 misfit at the receiver 45 m away (0.4 % / 0.7 % at
    95 m, 2.3 m deep: mesh resolution; the HIP path at h = 0.625 / P4 reaches 1e-4 there too) once the point-source
    solution is integrated over the 1 m source box.

    And what that says about REF-C1..3 (the reference's only stored numbers): at the positions uy.py probes (1 m below
    the surface) REF-C's uy is 0.78 of the exact solution and its ux 1.25-1.5 of it; REF-C fits the exact solution
    best for receivers AT the surface above a source 1 m deep (the receiver over the source within 3 %, the far field a
    consistent 0.81-0.84 in both components at 45 and 95 m; 12 % misfit in shape) - it was not made for the set-up
    explosive_source_lf4.py / uy.py describe."""
    from oracle.analytic import explosive_box_source_halfspace, explosive_line_source_halfspace
    d = np.load(os.path.join(GOLD, "halfspace_oracle.npz"))
    t, tr, src, Vp, Vs = d["times"], d["traces"], d["src"], float(d["Vp"]), float(d["Vs"])
    for i, tol_a, tol_m in ((0, 1e-3, 2e-3), (3, 4e-3, 7e-3)):     # 45 m / 1 m deep; 95 m / 2.3 m deep (coarser per wavelength)
        x, y = d["receivers"][i]
        vx, vz = explosive_box_source_halfspace(x - src[0], 150.0 - y, 150.0 - src[1], t, Vp, Vs, period=2000.0)
        w = (t > 0.3) & (t < (x - src[0]) / (0.9194 * Vs) + 0.75)
        for ours, exact in ((tr[:, i, 0], vx), (-tr[:, i, 1], vz)):
            a = np.dot(ours[w], exact[w]) / np.dot(exact[w], exact[w])
            m = np.linalg.norm(ours[w] - exact[w]) / np.linalg.norm(exact[w])
            assert abs(a - 1.0) < tol_a and m < tol_m, (x, y, a, m)
    # REF-C2 / C3 against the exact solution at the probe positions of uy.py (x - 45 = 45, 95; 1 m deep)
    refs = [np.loadtxt(os.path.join(GOLD, "ref_c%d.txt" % i)) for i in (1, 2, 3)]
    tt = refs[0][:, 0]
    for i, xr, want_uy in ((1, 45.0, 0.778), (2, 95.0, 0.776)):
        vx, vz = explosive_line_source_halfspace(xr, 1.0, 1.0, tt, Vp, Vs, period=2000.0)
        w = (tt > UY_WINDOWS[i][0]) & (tt < UY_WINDOWS[i][1])
        a = np.dot(refs[i][w, 2], vz[w]) / np.dot(vz[w], vz[w])
        assert abs(a / want_uy - 1.0) < 0.02 and np.corrcoef(refs[i][w, 2], vz[w])[0, 1] > 0.99, (i, a)
        ax = np.dot(refs[i][w, 1], vx[w]) / np.dot(vx[w], vx[w])
        assert ax > 1.2
    # ... and at the SURFACE above the same source: the receiver over the source (C1) within 5 %, the far field a
    # consistent 0.81-0.84 in both components at both distances
    got = []
    for i, xr in ((0, 1e-6), (1, 45.0), (2, 95.0)):
        vx, vz = explosive_line_source_halfspace(xr, 0.0, 1.0, tt, Vp, Vs, period=2000.0)
        w = (tt > UY_WINDOWS[i][0]) & (tt < UY_WINDOWS[i][1])
        got.append(np.dot(refs[i][w, 2], vz[w]) / np.dot(vz[w], vz[w]))
        if i:
            got.append(np.dot(refs[i][w, 1], vx[w]) / np.dot(vx[w], vx[w]))
    assert abs(got[0] - 1.03) < 0.05 and all(0.78 < v < 0.87 for v in got[1:]), got


def test_source_box_projection():
    """The two independent implementations of the projected source (oracle: box clipped by the triangle's edges;
    product host code: triangle clipped by the box's edges) agree, integrate to the box area on every mesh, and
    reproduce the indicator where the box is a union of cells."""
    from seigen_amd import FunctionSpace, RectangleMesh
    from seigen_amd.functionspace import Function, integral, project_box_indicator
    for hh, P in ((2.5, 2), (1.25, 3), (2.5, 4)):
        nx, ny = int(300 / hh), int(150 / hh)
        V = FunctionSpace(RectangleMesh(nx, ny, 300.0, 150.0), "DG", P)
        a = project_box_indicator(V, (44.5, 148.5), (45.5, 149.5))
        b = harness.project_box_indicator(omesh.RectangleMesh(nx, ny, 300.0, 150.0), P, (44.5, 148.5), (45.5, 149.5))
        assert np.abs(a - b).max() < 1e-12
        f = Function(V)
        f.assign(a)
        assert abs(float(integral(f)) - 1.0) < 1e-12
    V = FunctionSpace(RectangleMesh(4, 4, 4.0, 4.0), "DG", 2)
    v = project_box_indicator(V, (1.0, 1.0), (3.0, 2.0))
    X = V.node_coords().mean(axis=1)
    inside = (X[:, 0] > 1) & (X[:, 0] < 3) & (X[:, 1] > 1) & (X[:, 1] < 2)
    assert np.abs(v - inside[:, None]).max() < 1e-12


def test_explosive_oracle_rerun_matches_fixture():
    """The committed oracle trace is reproducible: first 120 steps live."""
    times, tr, _ = _load_traces()
    ex = harness.ExplosiveSource()
    ex.elastic.dt = 0.001
    t, live = ex.run(0.12)
    sel = slice(4, None, 5)
    np.testing.assert_allclose(t[sel], times[:24], atol=1e-12)
    scale = np.abs(tr[:24]).max()
    assert np.abs(live[sel] - tr[:24]).max() < 1e-9 * scale


def test_reference_timestep_is_unstable_with_the_explicit_sponge():
    """explosive_source_lf4.py:30-32 sets dt = cfl_dt(2.5, Vp, 0.5) = 0.012 s, for which the
    explicit sponge term has sigma*dt = 12: the scheme blows up (it was 'previously hard-coded to
    0.001 s', :32).  Documented so nobody benchmarks NaNs."""
    ex = harness.ExplosiveSource(Lx=100.0, Ly=50.0)
    assert abs(ex.elastic.dt - 0.012028483448806774) < 1e-15
    with np.errstate(all="ignore"):
        ex.elastic.run(60 * ex.elastic.dt)
    assert not np.isfinite(ex.elastic.u1).all() or np.abs(ex.elastic.u1).max() > 1e6


def test_stage_vectors_fixture_is_current():
    """tests/golden/stage_vectors.npz was produced by tests/golden/make_golden.py from this oracle."""
    d = np.load(os.path.join(GOLD, "stage_vectors.npz"))
    dim, P = int(d["c2_meta"][0]), int(d["c2_meta"][1])
    n = tuple(int(x) for x in d["c2_meta"][2:])
    m = omesh.structured(dim, n, tuple(d["c2_L"]), "left")
    E = ElasticOperators(m, P)
    np.testing.assert_allclose(E.apply_F(d["c2_T"], d["c2_u"]), d["c2_F"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(E.apply_G(d["c2_u"], 0.7, 0.3), d["c2_G"], rtol=0, atol=1e-13)


# ------------------------------------------------------------------------------ heterogeneous material
@pytest.mark.parametrize("case", ["stiffer", "denser"])
def test_two_layer_reflection_and_transmission_1d(case):
    """The build-defined heterogeneous extension (each cell scales its own g by its own lambda, mu; per-cell density in
    the physical update - DESIGN.md section 2; the reference defines no semantics for discontinuous coefficients)
    against theory: a pulse in a 1-D bar meets an interface between two media and splits into a reflected and a
    transmitted pulse with the velocity amplitudes (Z1 - Z2) / (Z1 + Z2) and 2 Z1 / (Z1 + Z2), Z = rho c.  Oracle,
    400 cells, P3: the field after the interaction equals the exact one to 1e-3 of the incident amplitude."""
    n, L, P = 400, 4.0, 3
    m = omesh.IntervalMesh(n, L)
    orc = OracleLF4(m, P)
    X = m.node_coords(P)[..., 0]
    xc = X.mean(axis=1)
    xi = 2.0                                           # the interface, on a cell boundary
    right = xc > xi
    rho1, M1 = 1.0, 1.0                                # lambda + 2 mu = 0.5 + 2 * 0.25
    if case == "stiffer":
        rho2, M2 = 1.0, 4.0                            # c2 = 2, Z2 = 2
    else:
        rho2, M2 = 4.0, 1.0                            # c2 = 1/2, Z2 = 2
    c1, c2 = math.sqrt(M1 / rho1), math.sqrt(M2 / rho2)
    Z1, Z2 = rho1 * c1, rho2 * c2
    orc.l = np.where(right, M2 / 2.0, M1 / 2.0)         # lambda = M / 2, mu = M / 4: lambda + 2 mu = M
    orc.mu = np.where(right, M2 / 4.0, M1 / 4.0)
    orc.density = np.where(right, rho2, rho1)
    orc.density_physical = True
    g = lambda x: np.exp(-50.0 * (x - 1.0) ** 2)
    orc.u0 = g(X)[..., None].copy()
    orc.s0 = (-Z1 * g(X))[..., None, None].copy()       # right-going in medium 1
    orc.dt = 0.0005
    T = 1.5                                             # the pulse (at x = 1) crosses x = 2 at t = 1
    orc.run(T)
    R, Tc = (Z1 - Z2) / (Z1 + Z2), 2 * Z1 / (Z1 + Z2)
    exact = np.where(X < xi, g(X - c1 * T) + R * g(2 * xi - X - c1 * T), Tc * g(xi + (X - xi) * c1 / c2 - c1 * T))
    err = np.abs(orc.u1[..., 0] - exact).max()
    assert abs(R + 1.0 / 3.0) < 1e-15 and abs(Tc - 2.0 / 3.0) < 1e-15
    assert err < 1e-3, (case, err)
    # and the stress is continuous across the interface, the transmitted one - Z2 * u
    sx = np.where(X < xi, -Z1 * g(X - c1 * T) + Z1 * R * g(2 * xi - X - c1 * T), -Z2 * Tc * g(xi + (X - xi) * c1 / c2 - c1 * T))
    assert np.abs(orc.s1[..., 0, 0] - sx).max() < 3e-3 * Z2
