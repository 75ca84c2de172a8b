"""A small exact evaluator of UFL-style weak forms on piecewise-polynomial fields.
TEST INFRASTRUCTURE (used by tests/test_oracle_forms_literal.py only).

Purpose: evaluate the right-hand-side forms of the reference *as they are written*
(``seigen/elastic.py:206`` for ``f``, ``:213-216`` for ``g``) - with ``grad``, ``div``,
``inner``, ``dot``, ``avg``, ``jump``, ``.T``, ``.dx(k)``, index notation with implied
summation, restrictions ``('+')``/``('-')`` and the measures ``dx``/``dS``/``ds`` -
on tiny patches of simplices with rational vertices, in exact rational arithmetic
(sympy), so that the oracle's K/Fc/Bc decomposition (oracle/forms.py) can be checked
against the text of the forms rather than against a hand derivation of them.

Operator semantics follow UFL [upstream, SURVEY Appendix A, U5]:
  grad(w)[i, j] = d w_i / d x_j          div(v)[i] = d v_ij / d x_j
  inner(a, b)   = full contraction       dot(a, b) = contraction of last/first index
  A * n (matrix * vector) = matrix-vector product; scalar * anything = scaling
  avg(f) = (f('+') + f('-')) / 2
  jump(v, n) = v('+') n('+') + v('-') n('-')           for scalar v
             = dot(v('+'), n('+')) + dot(v('-'), n('-'))  for tensor v
  a repeated index inside a product is summed;  f.dx(k) = d f / d x_k
  dx: all cells; dS: all interior facets (both restrictions available); ds: exterior facets
  n = outward unit normal of the cell a quantity is restricted to.
"""
import itertools
from fractions import Fraction
from math import factorial

import numpy as np
import sympy as sp

XS = sp.symbols("x0:3")


# ------------------------------------------------------------------------------------ geometry
class Patch(object):
    """A few simplices with rational vertex coordinates."""

    def __init__(self, vertices, cells):
        self.V = [tuple(sp.Rational(c) for c in v) for v in vertices]
        self.cells = [tuple(c) for c in cells]
        self.dim = d = len(self.V[0])
        self.x = XS[:d]
        table = {}
        for c, cell in enumerate(self.cells):
            for f in range(d + 1):
                key = tuple(sorted(cell[v] for v in range(d + 1) if v != f))
                table.setdefault(key, []).append((c, f))
        self.interior = [(e[0], e[1]) for k, e in sorted(table.items()) if len(e) == 2]
        self.exterior = [e[0] for k, e in sorted(table.items()) if len(e) == 1]

    def cell_vertices(self, c):
        return [sp.Matrix(self.V[v]) for v in self.cells[c]]

    def ref_to_phys(self, c):
        """(v0, J) with x = v0 + J xi."""
        X = self.cell_vertices(c)
        J = sp.Matrix.hstack(*[X[m + 1] - X[0] for m in range(self.dim)])
        return X[0], J

    def facet_vertices(self, c, f):
        return [sp.Matrix(self.V[self.cells[c][v]]) for v in range(self.dim + 1) if v != f]

    def scaled_normal(self, c, f):
        """|F| * (outward unit normal) of local facet f of cell c - a rational vector."""
        d = self.dim
        P = self.facet_vertices(c, f)
        opp = sp.Matrix(self.V[self.cells[c][f]])
        if d == 1:
            N = sp.Matrix([1])
        elif d == 2:
            t = P[1] - P[0]
            N = sp.Matrix([t[1], -t[0]])
        else:
            N = (P[1] - P[0]).cross(P[2] - P[0]) / 2
        if (N.T * (P[0] - opp))[0] < 0:
            N = -N
        return N

    def lagrange_basis(self, c, P):
        """Equispaced Lagrange basis of P_k on cell c as sympy expressions in x, in the
        oracle's node order (lattice, first reference coordinate fastest)."""
        from oracle import refelem
        d = self.dim
        v0, J = self.ref_to_phys(c)
        xi = J.inv() * (sp.Matrix(self.x) - v0)
        lam = [1 - sum(xi)] + list(xi)
        out = []
        for al in refelem.lattice(d, P):
            full = (P - int(al.sum()),) + tuple(int(a) for a in al)
            e = sp.Integer(1)
            for m in range(d + 1):
                for s in range(full[m]):
                    e = e * (P * lam[m] - s) / (full[m] - s)
            out.append(sp.expand(e))
        return out

    # ---- exact integration of a polynomial in x ---------------------------------------------
    def _integrate_affine(self, expr, p0, E, measure):
        """measure * int over the unit simplex (dimension = columns of E) of expr(p0 + E t)."""
        m = E.shape[1]
        if expr == 0:
            return sp.Integer(0)
        t = sp.symbols("t0:%d" % max(m, 1))[:m]
        sub = {self.x[i]: p0[i] + sum(E[i, k] * t[k] for k in range(m)) for i in range(self.dim)}
        e = sp.expand(sp.sympify(expr).subs(sub, simultaneous=True))
        if m == 0:
            return e * measure
        total = sp.Integer(0)
        for mon, coef in sp.Poly(e, *t).terms():
            num = 1
            for a in mon:
                num *= factorial(a)
            total += coef * sp.Rational(num, factorial(sum(mon) + m))
        return total * measure

    def integrate_cell(self, expr, c):
        v0, J = self.ref_to_phys(c)
        return self._integrate_affine(expr, v0, J, abs(J.det()))

    def integrate_facet(self, expr, c, f):
        """int over the facet with respect to dS / |F|  (the caller multiplies unit normals by
        |F| through `scaled_normal`, so all quantities stay rational)."""
        P = self.facet_vertices(c, f)
        m = self.dim - 1
        E = sp.Matrix.hstack(*[P[k + 1] - P[0] for k in range(m)]) if m > 0 else sp.zeros(self.dim, 0)
        return self._integrate_affine(expr, P[0], E, factorial(m))


# ------------------------------------------------------------------------------ expression tree
class Index(object):
    pass


class Ctx(object):
    """Where an integrand is being evaluated: a cell (dx), an exterior facet (ds), or an
    interior facet with its two sides (dS)."""

    def __init__(self, patch, kind, cell=None, facet=None, plus=None, minus=None):
        self.patch, self.kind, self.cell, self.facet, self.plus, self.minus = patch, kind, cell, facet, plus, minus

    def side(self, s):
        assert self.kind == "dS", "restriction outside an interior-facet integral"
        c, f = self.plus if s == "+" else self.minus
        return Ctx(self.patch, "side", cell=c, facet=f)


def _arr(v):
    a = np.empty(np.shape(v), dtype=object)
    a[...] = v
    return a


class Node(object):
    shape = ()
    free = ()

    # algebra --------------------------------------------------------------------------------
    def __add__(self, o):
        return Sum(self, as_node(o))

    __radd__ = __add__

    def __sub__(self, o):
        return Sum(self, Scale(-1, as_node(o)))

    def __rsub__(self, o):
        return Sum(as_node(o), Scale(-1, self))

    def __neg__(self):
        return Scale(-1, self)

    def __mul__(self, o):
        if isinstance(o, Measure):
            return Form([(self, o.kind)])
        return Product(self, as_node(o))

    def __rmul__(self, o):
        return Product(as_node(o), self)

    def __truediv__(self, o):
        return Product(self, as_node(1 / sp.sympify(o)))

    def __call__(self, side):
        return Restricted(self, side)

    def __getitem__(self, idx):
        idx = idx if isinstance(idx, tuple) else (idx,)
        return Indexed(self, idx)

    def dx(self, k):
        return Deriv(self, k)

    @property
    def T(self):
        return Transposed(self)


def as_node(o):
    return o if isinstance(o, Node) else Const(o)


class Const(Node):
    def __init__(self, v):
        self.v = _arr(v) if np.ndim(v) else sp.sympify(v)
        self.shape = np.shape(v)

    def ev(self, ctx, env):
        return self.v


class Coefficient(Node):
    """Piecewise polynomial field: data[cell] = sympy scalar or object array of the field's shape."""

    def __init__(self, shape, data):
        self.shape = tuple(shape)
        self.data = data

    def ev(self, ctx, env):
        assert ctx.kind != "dS", "unrestricted discontinuous coefficient in a dS integral"
        z = _arr(np.zeros(self.shape, dtype=int)) if self.shape else sp.Integer(0)
        return self.data.get(ctx.cell, z)


class FacetNormal(Node):
    """Evaluates to |F| n (rational); facet integrals are taken per unit facet measure."""

    def __init__(self, dim):
        self.shape = (dim,)

    def ev(self, ctx, env):
        assert ctx.kind in ("ds", "side"), "facet normal needs a side"
        return _arr(list(ctx.patch.scaled_normal(ctx.cell, ctx.facet)))


class Restricted(Node):
    def __init__(self, a, side):
        self.a, self.s, self.shape, self.free = a, side, a.shape, a.free

    def ev(self, ctx, env):
        return self.a.ev(ctx.side(self.s), env)


class Sum(Node):
    def __init__(self, a, b):
        assert a.shape == b.shape and set(a.free) == set(b.free)
        self.a, self.b, self.shape, self.free = a, b, a.shape, a.free

    def ev(self, ctx, env):
        return self.a.ev(ctx, env) + self.b.ev(ctx, env)


class Scale(Node):
    def __init__(self, c, a):
        self.c, self.a, self.shape, self.free = c, a, a.shape, a.free

    def ev(self, ctx, env):
        return self.c * self.a.ev(ctx, env)


class Product(Node):
    """scalar*anything, matrix*vector, and products of indexed scalars with implied summation."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        if a.shape == () or b.shape == ():
            self.shape = a.shape or b.shape
        elif len(a.shape) == 2 and len(b.shape) == 1:
            self.shape = (a.shape[0],)
        else:
            raise TypeError("unsupported product %r * %r" % (a.shape, b.shape))
        self.summed = tuple(i for i in a.free if i in b.free)
        self.free = tuple(i for i in a.free + b.free if i not in self.summed)
        self.dim = None

    def ev(self, ctx, env):
        d = ctx.patch.dim
        total = None
        for vals in itertools.product(range(d), repeat=len(self.summed)):
            e = dict(env)
            e.update(zip(self.summed, vals))
            x, y = self.a.ev(ctx, e), self.b.ev(ctx, e)
            if len(self.a.shape) == 2 and len(self.b.shape) == 1:
                t = _arr([sum(x[i, j] * y[j] for j in range(len(y))) for i in range(x.shape[0])])
            else:
                t = x * y
            total = t if total is None else total + t
        return total


class Indexed(Node):
    def __init__(self, a, idx):
        assert len(idx) == len(a.shape)
        self.a, self.idx = a, idx
        self.free = a.free + tuple(i for i in idx if isinstance(i, Index))

    def ev(self, ctx, env):
        v = self.a.ev(ctx, env)
        return v[tuple(env[i] if isinstance(i, Index) else i for i in self.idx)]


class Deriv(Node):
    def __init__(self, a, k):
        assert a.shape == ()
        self.a, self.k = a, k
        self.free = a.free + ((k,) if isinstance(k, Index) and k not in a.free else ())

    def ev(self, ctx, env):
        k = env[self.k] if isinstance(self.k, Index) else self.k
        return sp.diff(self.a.ev(ctx, env), ctx.patch.x[k])


class Transposed(Node):
    def __init__(self, a):
        assert len(a.shape) == 2
        self.a, self.shape = a, a.shape[::-1]

    def ev(self, ctx, env):
        return self.a.ev(ctx, env).T


class Grad(Node):
    def __init__(self, a):
        assert a.shape, "grad of a scalar is not needed by the forms"
        self.a, self.shape = a, a.shape + (a.shape[-1],)

    def ev(self, ctx, env):
        v = self.a.ev(ctx, env)
        x = ctx.patch.x
        out = np.empty(np.shape(v) + (len(x),), dtype=object)
        for idx in itertools.product(*[range(n) for n in np.shape(v)]):
            for j in range(len(x)):
                out[idx + (j,)] = sp.diff(v[idx], x[j])
        return out


def grad(a):
    return Grad(a)


class Div(Node):
    def __init__(self, a):
        assert len(a.shape) >= 1
        self.a, self.shape = a, a.shape[:-1]

    def ev(self, ctx, env):
        v = self.a.ev(ctx, env)
        x = ctx.patch.x
        out = np.empty(v.shape[:-1], dtype=object)
        for idx in itertools.product(*[range(n) for n in v.shape[:-1]]):
            out[idx] = sum(sp.diff(v[idx + (j,)], x[j]) for j in range(len(x)))
        return out if out.shape else out[()]


def div(a):
    return Div(a)


class Inner(Node):
    def __init__(self, a, b):
        assert a.shape == b.shape, (a.shape, b.shape)
        self.a, self.b = a, b

    def ev(self, ctx, env):
        x, y = self.a.ev(ctx, env), self.b.ev(ctx, env)
        return (x * y).sum() if np.ndim(x) else x * y


def inner(a, b):
    return Inner(as_node(a), as_node(b))


class Dot(Node):
    def __init__(self, a, b):
        self.a, self.b = a, b
        self.shape = a.shape[:-1] + b.shape[1:]

    def ev(self, ctx, env):
        r = np.tensordot(self.a.ev(ctx, env), self.b.ev(ctx, env), axes=(-1, 0))
        return r if np.ndim(r) else r[()]


def dot(a, b):
    return Dot(as_node(a), as_node(b))


def avg(a):
    return Scale(sp.Rational(1, 2), Sum(Restricted(a, "+"), Restricted(a, "-")))


def jump(v, n):
    if len(v.shape) == 0:
        return Sum(Product(Restricted(v, "+"), Restricted(n, "+")), Product(Restricted(v, "-"), Restricted(n, "-")))
    return Sum(Dot(Restricted(v, "+"), Restricted(n, "+")), Dot(Restricted(v, "-"), Restricted(n, "-")))


class Measure(object):
    def __init__(self, kind):
        self.kind = kind

    def __rmul__(self, o):
        return Form([(as_node(o), self.kind)])


dx, dS, ds = Measure("dx"), Measure("dS"), Measure("ds")


class Form(object):
    def __init__(self, integrals):
        self.integrals = integrals

    def __add__(self, o):
        return Form(self.integrals + o.integrals)

    def __sub__(self, o):
        return Form(self.integrals + [(Scale(-1, n), k) for n, k in o.integrals])

    def __neg__(self):
        return Form([(Scale(-1, n), k) for n, k in self.integrals])

    def __rmul__(self, c):
        return Form([(Scale(sp.sympify(c), n), k) for n, k in self.integrals])


def assemble(form, patch):
    """Exact value of a form whose arguments have all been replaced by concrete fields."""
    total = sp.Integer(0)
    for node, kind in form.integrals:
        assert node.shape == () and not node.free, "integrand must be a scalar without free indices"
        if kind == "dx":
            for c in range(len(patch.cells)):
                total += patch.integrate_cell(node.ev(Ctx(patch, "dx", cell=c), {}), c)
        elif kind == "ds":
            for (c, f) in patch.exterior:
                total += patch.integrate_facet(node.ev(Ctx(patch, "ds", cell=c, facet=f), {}), c, f)
        else:
            for (p, m) in patch.interior:
                total += patch.integrate_facet(node.ev(Ctx(patch, "dS", plus=p, minus=m), {}), p[0], p[1])
    return total
