"""Problem set-ups of the reference's test scripts, restated for the oracle.
ORACLE (test infrastructure) - see oracle/__init__.py.

  Eigenmode2D  : tests/eigenmode/eigenmode_2d.py:7-65
  Eigenmode3D  : tests/eigenmode/eigenmode_3d.py:7-69
  ExplosiveSource : tests/explosive_source/explosive_source_lf4.py:7-56,
                    receivers of tests/explosive_source/uy.py:36-43
"""
import math
import numpy as np
from . import refelem, mesh as omesh
from .lf4 import OracleLF4


# ----------------------------------------------------------------------------- norms
def l2_norm(mesh, P, e):
    """sqrt(int |e|^2 dx) of a DG_P field e [nc, nd, ...] (exact: mass matrix)."""
    d, kind = mesh.dim, getattr(mesh, "kind", "simplex")
    xq, wq = refelem.el_quadrature(d, 2 * P, kind)
    phi, _ = refelem.el_tabulate(d, P, xq, kind)
    Mref = np.einsum('q,qa,qb->ab', wq, phi, phi)
    e2 = e.reshape(mesh.ncells, e.shape[1], -1)
    return math.sqrt(np.einsum('c,cak,ab,cbk->', np.abs(mesh.detJ), e2, Mref, e2))


def projected_abs_norm(mesh, P, e, Pproj):
    """|| Pi_{DG Pproj} |e| ||_L2 - the error functional of
    eigenmode_2d.py:49-63 (Pproj=6) / eigenmode_3d.py:53-67 (Pproj=3):
    component-wise abs of the DG_P field e, L2-projected into DG_Pproj, then
    norm().  [upstream] quadrature degree = Pproj + P (UFL degree estimation:
    abs() keeps the degree of its operand)."""
    d, kind = mesh.dim, getattr(mesh, "kind", "simplex")
    xq, wq = refelem.el_quadrature(d, Pproj + P, kind)
    phi, _ = refelem.el_tabulate(d, P, xq, kind)
    psi, _ = refelem.el_tabulate(d, Pproj, xq, kind)
    xm, wm = refelem.el_quadrature(d, 2 * Pproj, kind)
    psm, _ = refelem.el_tabulate(d, Pproj, xm, kind)
    Mproj = np.einsum('q,qa,qb->ab', wm, psm, psm)
    Minv = np.linalg.inv(Mproj)
    e2 = e.reshape(mesh.ncells, e.shape[1], -1)
    eq = np.abs(np.einsum('qa,cak->cqk', phi, e2))
    b = np.einsum('q,qa,cqk->cak', wq, psi, eq)            # / |detJ| cancels below
    return math.sqrt(np.einsum('c,cak,ab,cbk->', np.abs(mesh.detJ), b, Minv, b))


# ----------------------------------------------------------------------------- eigenmode
class Eigenmode2D(object):
    def __init__(self, N, degree, dt, diagonal="left", quadrilateral=False):
        self.mesh = omesh.UnitSquareMesh(N, N, diagonal, quadrilateral=quadrilateral)   # eigenmode_2d.py:11
        self.elastic = OracleLF4(self.mesh, degree)
        self.elastic.density = 1.0                                        # :17-20
        self.elastic.dt = dt
        self.elastic.mu = 0.25
        self.elastic.l = 0.5
        vs = math.sqrt(self.elastic.mu / self.elastic.density)
        self.a = math.sqrt(2) * math.pi * vs                              # :25
        self.b = 2 * math.pi * self.elastic.mu                            # :26

    def u_exact(self, X, t):
        a, x, y = self.a, X[..., 0], X[..., 1]
        pi = math.pi
        return np.stack([a * np.cos(pi * x) * np.sin(pi * y) * math.cos(a * t),
                         -a * np.sin(pi * x) * np.cos(pi * y) * math.cos(a * t)], axis=-1)

    def s_exact(self, X, t):
        a, b, x, y = self.a, self.b, X[..., 0], X[..., 1]
        pi = math.pi
        out = np.zeros(X.shape[:-1] + (2, 2))
        out[..., 0, 0] = -b * np.sin(pi * x) * np.sin(pi * y) * math.sin(a * t)
        out[..., 1, 1] = b * np.sin(pi * x) * np.sin(pi * y) * math.sin(a * t)
        return out

    def run(self, T=5.0, max_steps=None):
        el = self.elastic
        X = el.node_coords()
        el.u0 = self.u_exact(X, 0.0)                                      # :30-32
        el.s0 = self.s_exact(X, el.dt / 2.0)                              # :33-36
        return el.run(T, max_steps)

    def errors(self, u1, s1, Pproj=6):
        el = self.elastic
        X = el.node_coords()
        ue = self.u_exact(X, 5.0)                                         # hard-coded t=5, :41-43
        se = self.s_exact(X, 5.0 + el.dt / 2.0)                           # :44-47
        P = el.degree
        return dict(
            u_l2=l2_norm(self.mesh, P, u1 - ue), s_l2=l2_norm(self.mesh, P, s1 - se),
            u_error=projected_abs_norm(self.mesh, P, u1 - ue, Pproj),
            s_error=projected_abs_norm(self.mesh, P, s1 - se, Pproj))


class Eigenmode3D(object):
    def __init__(self, N, degree, dt, hexahedral=False):
        self.mesh = omesh.UnitCubeMesh(N, N, N, hexahedral=hexahedral)    # eigenmode_3d.py:11
        self.elastic = OracleLF4(self.mesh, degree)
        self.elastic.density = 1.0                                        # :17-20
        self.elastic.dt = dt
        self.elastic.mu = 0.25
        self.elastic.l = 0.5
        self.A = math.sqrt(2 * self.elastic.density * self.elastic.mu)    # :25
        self.O = math.pi * math.sqrt(2 * self.elastic.mu / self.elastic.density)  # :26

    def u_exact(self, X, t):
        pi, O = math.pi, self.O
        x, y, z = X[..., 0], X[..., 1], X[..., 2]
        c = math.cos(O * t)
        return np.stack([np.cos(pi * x) * (np.sin(pi * y) - np.sin(pi * z)) * c,
                         np.cos(pi * y) * (np.sin(pi * z) - np.sin(pi * x)) * c,
                         np.cos(pi * z) * (np.sin(pi * x) - np.sin(pi * y)) * c], axis=-1)

    def s_exact(self, X, t):
        pi, O, A = math.pi, self.O, self.A
        x, y, z = X[..., 0], X[..., 1], X[..., 2]
        s = math.sin(O * t)
        out = np.zeros(X.shape[:-1] + (3, 3))
        out[..., 0, 0] = -A * np.sin(pi * x) * (np.sin(pi * y) - np.sin(pi * z)) * s
        out[..., 1, 1] = -A * np.sin(pi * y) * (np.sin(pi * z) - np.sin(pi * x)) * s
        out[..., 2, 2] = -A * np.sin(pi * z) * (np.sin(pi * x) - np.sin(pi * y)) * s
        return out

    def run(self, T=5.0, max_steps=None):
        el = self.elastic
        X = el.node_coords()
        el.u0 = self.u_exact(X, 0.0)                                      # :30-34
        el.s0 = self.s_exact(X, el.dt / 2.0)                              # :35-39
        return el.run(T, max_steps)

    def errors(self, u1, s1, Pproj=3):
        el = self.elastic
        X = el.node_coords()
        ue = self.u_exact(X, 5.0)                                         # :43-47
        se = self.s_exact(X, 5.0 + el.dt / 2.0)                           # :48-52
        P = el.degree
        return dict(
            u_l2=l2_norm(self.mesh, P, u1 - ue), s_l2=l2_norm(self.mesh, P, s1 - se),
            u_error=projected_abs_norm(self.mesh, P, u1 - ue, Pproj),
            s_error=projected_abs_norm(self.mesh, P, s1 - se, Pproj))


# ----------------------------------------------------------------------------- explosive source
def ricker(t, a=159.42, t0=0.3):
    """(-1 + 2 a (t-t0)^2) exp(-a (t-t0)^2)   explosive_source_lf4.py:37-38"""
    return (-1.0 + 2 * a * (t - t0) ** 2) * math.exp(-a * (t - t0) ** 2)


def _clip_halfplane(poly, p0, nrm):
    """part of the convex polygon poly [n, 2] on the side (x - p0).nrm >= 0"""
    out = []
    s = (poly - p0) @ nrm
    for i in range(len(poly)):
        j = (i + 1) % len(poly)
        if s[i] >= 0:
            out.append(poly[i])
        if (s[i] >= 0) != (s[j] >= 0):
            out.append(poly[i] + (s[i] / (s[i] - s[j])) * (poly[j] - poly[i]))
    return np.array(out).reshape(-1, 2)


def project_box_indicator(mesh, P, lo, hi):
    """L2 projection of the indicator of the box [lo, hi] onto scalar DG_P (2-D) [nc, nd]:
    M_K^-1 int_{K n box} phi_a.  The box polygon is clipped by the three edge half-planes of every
    triangle it can touch, fan-triangulated and integrated with a rule exact for P_k.  (Build-defined
    source of the REF-C convergence study, not in the reference - see ExplosiveSource.source_mode.)"""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    if getattr(mesh, "kind", "simplex") == "tensor":
        # axis-parallel rectangles: K n box is a rectangle; Gauss-Legendre product rule on it
        nd = refelem.el_nnodes(2, P, "tensor")
        out = np.zeros((mesh.ncells, nd))
        X = mesh.vertices[mesh.cells]                          # [nc, 4, 2]
        xq, wq = refelem.el_quadrature(2, P, "tensor")
        xm, wm = refelem.el_quadrature(2, 2 * P, "tensor")
        pm, _ = refelem.el_tabulate(2, P, xm, "tensor")
        Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wm, pm, pm))
        for c in range(mesh.ncells):
            a0, a1 = np.maximum(lo, X[c].min(axis=0)), np.minimum(hi, X[c].max(axis=0))
            if (a1 - a0).min() <= 0:
                continue
            xp = a0 + xq * (a1 - a0)
            xi = np.einsum('mi,qi->qm', mesh.Jinv[c], xp - mesh.v0[c])
            phi, _ = refelem.el_tabulate(2, P, xi, "tensor")
            out[c] = Minv @ (np.prod(a1 - a0) * (wq @ phi)) / abs(mesh.detJ[c])
        return out
    nd = refelem.nnodes(2, P)
    out = np.zeros((mesh.ncells, nd))
    X = mesh.vertices[mesh.cells]                              # [nc, 3, 2]
    touch = np.nonzero((X[..., 0].max(1) > lo[0]) & (X[..., 0].min(1) < hi[0]) &
                       (X[..., 1].max(1) > lo[1]) & (X[..., 1].min(1) < hi[1]))[0]
    xq, wq = refelem.simplex_quadrature(2, P)
    xm, wm = refelem.simplex_quadrature(2, 2 * P)
    pm, _ = refelem.tabulate(2, P, xm)
    Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wm, pm, pm))
    box = np.array([[lo[0], lo[1]], [hi[0], lo[1]], [hi[0], hi[1]], [lo[0], hi[1]]])
    for c in touch:
        V = X[c]
        ctr = V.mean(axis=0)
        poly = box
        for e in range(3):
            a, b = V[e], V[(e + 1) % 3]
            nrm = np.array([-(b - a)[1], (b - a)[0]])
            if (ctr - a) @ nrm < 0:
                nrm = -nrm
            poly = _clip_halfplane(poly, a, nrm)
            if len(poly) < 3:
                break
        if len(poly) < 3:
            continue
        b_a = np.zeros(nd)
        for t in range(1, len(poly) - 1):
            T = np.array([poly[0], poly[t], poly[t + 1]])
            Jt = (T[1:] - T[0]).T
            xp = T[0] + xq @ Jt.T
            xi = np.einsum('mi,qi->qm', mesh.Jinv[c], xp - mesh.v0[c])
            phi, _ = refelem.tabulate(2, P, xi)
            b_a += abs(np.linalg.det(Jt)) * (wq @ phi)
        out[c] = Minv @ b_a / abs(mesh.detJ[c])
    return out


class ExplosiveSource(object):
    """explosive_source_lf4.py:12-56 on RectangleMesh(int(Lx/h), int(Ly/h), Lx, Ly).

    source_mode: 'interpolate' = the reference's nodal interpolation of the box indicator (:36-40);
    'unit_integral' = that interpolant scaled to the box's area; 'project' = its L2 projection (area exact on
    every mesh).  The last two are build-defined, for the REF-C convergence study."""

    def __init__(self, Lx=300.0, Ly=150.0, h=2.5, degree=2, diagonal="left",
                 src=(45.0, None), sponge=20.0, sigma_degree=4, source_mode="interpolate", quadrilateral=False):
        nx, ny = int(Lx / h), int(Ly / h)
        self.Lx, self.Ly, self.h = Lx, Ly, h
        self.mesh = omesh.RectangleMesh(nx, ny, Lx, Ly, diagonal, quadrilateral=quadrilateral)   # :9-10
        el = self.elastic = OracleLF4(self.mesh, degree)
        el.density = 1.0                                                  # :21-23
        el.mu = 3600.0
        el.l = 3599.3664
        self.Vp = math.sqrt((el.l + 2 * el.mu) / el.density)              # helpers.py:15-28
        self.Vs = math.sqrt(el.mu / el.density)
        el.dt = (0.5 * h) / self.Vp                                       # cfl_dt, helpers.py:46-54; :30-32
        X = el.node_coords()
        sx = src[0]
        sy = (Ly - 1.0) if src[1] is None else src[1]
        # closed box [sx-0.5, sx+0.5] x [sy-0.5, sy+0.5]                   :37-38
        self.src_mask = ((X[..., 0] >= sx - 0.5) & (X[..., 0] <= sx + 0.5) &
                         (X[..., 1] >= sy - 0.5) & (X[..., 1] <= sy + 0.5))
        d = 2
        scalar = self.src_mask.astype(np.float64)
        ops = el.E.ops
        if source_mode == "project":
            scalar = project_box_indicator(self.mesh, degree, (sx - 0.5, sy - 0.5), (sx + 0.5, sy + 0.5))
        elif source_mode == "unit_integral":
            scalar = scalar / float((ops.M @ scalar.reshape(-1)).sum())
        elif source_mode != "interpolate":
            raise ValueError("source_mode")
        self.source_integral = float((ops.M @ scalar.reshape(-1)).sum())
        pattern = np.zeros(X.shape[:-1] + (d, d))
        pattern[..., 0, 0] = scalar
        pattern[..., 1, 1] = scalar
        self.pattern = pattern
        el.source = lambda t: ricker(t) * pattern                          # :36-40, elastic.py:285-288
        # DG4 sponge                                                       :43-45
        Xs = self.mesh.node_coords(sigma_degree)
        sig = np.where((Xs[..., 0] <= sponge) | (Xs[..., 0] >= Lx - sponge) | (Xs[..., 1] <= sponge),
                       1000.0, 0.0)
        self.sigma = sig
        el.E.set_absorption(sig, sigma_degree)

    def point_evaluator(self, x, y):
        """Returns (cell, phi[nd]) to evaluate a DG field at (x, y)."""
        m = self.mesh
        P = self.elastic.degree
        xi = np.einsum('cmi,ci->cm', m.Jinv, np.array([x, y])[None, :] - m.v0)
        lam0 = (1.0 - xi.max(axis=1)) if m.kind == "tensor" else (1.0 - xi.sum(axis=1))
        inside = (xi.min(axis=1) >= -1e-12) & (lam0 >= -1e-12)
        c = int(np.nonzero(inside)[0][0])
        phi, _ = refelem.el_tabulate(2, P, xi[c][None, :], m.kind)
        return c, phi[0]

    def run(self, T=2.5, receivers=((45.0, 149.0), (90.0, 149.0), (140.0, 149.0))):
        el = self.elastic
        ev = [self.point_evaluator(x, y) for (x, y) in receivers]
        times, traces = [], []

        def probe(n, t, u1, s1):
            times.append(t)
            traces.append([[float(phi @ u1[c, :, k]) for k in range(2)] for (c, phi) in ev])

        el.probes = probe
        el.run(T)
        return np.array(times), np.array(traces)     # [nsteps], [nsteps, nrecv, 2]
