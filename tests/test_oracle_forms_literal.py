"""The oracle's weak-form decomposition against the TEXT of the reference's forms.

``oracle/forms.py`` assembles ``f`` and ``g`` from scalar matrices K_j, Fc_j, Bc_j - a hand
derivation of ``seigen/elastic.py:204-219``.  Here the two forms are written out exactly as
the reference writes them (``:206`` and ``:213-216``: same operators, same index notation,
same restrictions and measures) on top of a small exact evaluator of UFL semantics
(tests/ufl_literal.py), applied to every (test basis function, trial basis function) pair of
small patches - two triangles with four boundary edges, two tetrahedra, two intervals, with
rational vertices and both orientations - and compared entry by entry with what the oracle's
matrices predict:

    f(w = phi_a^c e_i ; s0 = phi_b^c' e_m (x) e_n)      = delta_im AF_n[(c,a),(c',b)]
    g(v = phi_a^c e_i (x) e_j ; u1 = phi_b^c' e_k)      = l delta_ij AG_k[..] + mu (delta_ik AG_j[..] + delta_jk AG_i[..])

So the K/Fc/Bc split, the signs, the 1/2 of avg(), which side's normal each dS term sees and
the own-trace ds terms are checked against the form text, not against the builder's reading of
it.  (What remains [upstream] is UFL's definition of the operators themselves, listed in
tests/ufl_literal.py and SURVEY Appendix A U5.)
"""
import itertools

import numpy as np
import pytest
import sympy as sp

from oracle import mesh as omesh
from oracle.forms import ScalarOperators
from tests.ufl_literal import (Patch, Coefficient, Const, FacetNormal, Index, assemble, avg, div, dot, dS, ds, dx,
                               grad, inner, jump)

i, j, k = Index(), Index(), Index()


# ---- the two forms, as written at seigen/elastic.py:206 and :213-216 ---------------------------
def f(w, s0, n):
    """The RHS of the velocity equation (without the optional absorption term)."""
    f = -inner(grad(w), s0)*dx + inner(avg(s0)*n('+'), w('+'))*dS + inner(avg(s0)*n('-'), w('-'))*dS
    return f


def g(v, u1, I, n, l, mu):
    """The RHS of the stress equation (without the optional source term)."""
    g = - l*(v[i, j]*I[i, j]).dx(k)*u1[k]*dx + l*(jump(v[i, j], n[k])*I[i, j]*avg(u1[k]))*dS \
        + l*(v[i, j]*I[i, j]*u1[k]*n[k])*ds - mu*inner(div(v), u1)*dx + mu*inner(avg(u1), jump(v, n))*dS \
        - mu*inner(div(v.T), u1)*dx + mu*inner(avg(u1), jump(v.T, n))*dS \
        + mu*inner(u1, dot(v, n))*ds + mu*inner(u1, dot(v.T, n))*ds
    return g


# ---- patches -------------------------------------------------------------------------------------
PATCHES = {
    # two intervals of different length
    "1d": ([(0,), ("3/4",), (2,)], [(0, 1), (1, 2)]),
    # a skewed quadrilateral cut in two: 1 interior + 4 exterior edges, opposite orientations
    "2d_skew": ([(0, 0), (2, 0), ("1/2", "3/2"), ("5/2", 2)], [(0, 1, 2), (3, 2, 1)]),
    # the unit square cut the way the structured meshes cut it ("left" diagonal, oracle/mesh.py)
    "2d_unit": ([(0, 0), (1, 0), (0, 1), (1, 1)], [(0, 1, 2), (3, 2, 1)]),
    # two tetrahedra sharing a face, the second one negatively oriented
    "3d": ([(0, 0, 0), (1, 0, 0), (0, "3/2", 0), (0, 0, 2), (1, 1, 1)], [(0, 1, 2, 3), (4, 1, 2, 3)]),
}


def _unit(shape, idx, phi):
    a = np.empty(shape, dtype=object)
    a[...] = sp.Integer(0)
    a[idx] = phi
    return a


def _setup(name, P):
    verts, cells = PATCHES[name]
    patch = Patch(verts, cells)
    mesh = omesh.Mesh(np.array([[float(sp.Rational(c)) for c in v] for v in verts]), cells)
    ops = ScalarOperators(mesh, P)
    basis = [patch.lagrange_basis(c, P) for c in range(len(cells))]
    return patch, ops, basis


def _pairs(nc, nd, d, sample, seed):
    """(test cell, test node, trial cell, trial node) combinations: all, or a seeded sample."""
    allp = list(itertools.product(range(nc), range(nd), range(nc), range(nd)))
    if sample is None or sample >= len(allp):
        return allp
    rng = np.random.default_rng(seed)
    return [allp[q] for q in rng.choice(len(allp), size=sample, replace=False)]


# (patch, degree, number of sampled (test, trial) node pairs or None for all of them); every pair
# is evaluated for all component combinations.  3-D degree 1 in full takes two minutes (run once:
# all 64 pairs agree to 1e-15), so the suite keeps a seeded sample.
CASES = [("1d", 2, None), ("2d_skew", 1, None), ("2d_unit", 1, None), ("2d_skew", 2, 24), ("2d_unit", 3, 10),
         ("3d", 1, 16), ("3d", 2, 6)]


@pytest.mark.parametrize("name,P,sample", CASES)
def test_f_as_written_matches_oracle_AF(name, P, sample):
    patch, ops, basis = _setup(name, P)
    d, nd, nc = patch.dim, ops.nd, len(patch.cells)
    AF = [A.toarray() for A in ops.AF]
    n = FacetNormal(d)
    worst = 0.0
    for (c, a, c2, b) in _pairs(nc, nd, d, sample, 1):
        for ti in range(d):
            w = Coefficient((d,), {c: _unit((d,), (ti,), basis[c][a])})
            for (m, nn) in itertools.product(range(d), repeat=2):
                s0 = Coefficient((d, d), {c2: _unit((d, d), (m, nn), basis[c2][b])})
                val = float(assemble(f(w, s0, n), patch))
                want = AF[nn][c * nd + a, c2 * nd + b] if ti == m else 0.0
                worst = max(worst, abs(val - want))
    assert worst < 1e-13, worst


@pytest.mark.parametrize("name,P,sample", CASES)
def test_g_as_written_matches_oracle_AG(name, P, sample):
    patch, ops, basis = _setup(name, P)
    d, nd, nc = patch.dim, ops.nd, len(patch.cells)
    AG = [A.toarray() for A in ops.AG]
    n = FacetNormal(d)
    I = Const(np.eye(d, dtype=int))
    l, mu = sp.Rational(3, 7), sp.Rational(5, 11)
    worst = 0.0
    for (c, a, c2, b) in _pairs(nc, nd, d, sample, 2):
        for (ti, tj) in itertools.product(range(d), repeat=2):
            v = Coefficient((d, d), {c: _unit((d, d), (ti, tj), basis[c][a])})
            for tk in range(d):
                u1 = Coefficient((d,), {c2: _unit((d,), (tk,), basis[c2][b])})
                val = float(assemble(g(v, u1, I, n, l, mu), patch))
                r, s = c * nd + a, c2 * nd + b
                want = float(l) * (ti == tj) * AG[tk][r, s] + float(mu) * ((ti == tk) * AG[tj][r, s] + (tj == tk) * AG[ti][r, s])
                worst = max(worst, abs(val - want))
    assert worst < 1e-13, worst


def test_evaluator_knows_the_divergence_theorem():
    """Self-check of the evaluator: for a continuous polynomial vector field q on the patch,
    int div(q) dx = sum over exterior facets of q.n - and the interior-facet jump vanishes."""
    for name in ("2d_skew", "3d"):
        verts, cells = PATCHES[name]
        patch = Patch(verts, cells)
        d = patch.dim
        x = patch.x
        q = [x[0] ** 2 + 3 * x[d - 1], x[0] * x[1] - 2] + ([x[2] ** 2 * x[0]] if d == 3 else [])
        qf = Coefficient((d,), {c: _unit((d,), slice(None), q) for c in range(len(cells))})
        n = FacetNormal(d)
        lhs = assemble(div(qf) * dx, patch)
        rhs = assemble(inner(qf, n) * ds, patch)
        assert sp.simplify(lhs - rhs) == 0
        one = Coefficient((), {c: sp.Integer(1) for c in range(len(cells))})
        assert assemble(inner(jump(one, n), avg(qf)) * dS, patch) == 0
