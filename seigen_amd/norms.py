"""Norms and the eigenmode error functional (host side, numpy).

Counterpart of what the reference's harnesses ask Firedrake for:
``norm(f)`` and the projected-abs error of ``tests/eigenmode/eigenmode_2d.py:49-63``
/ ``eigenmode_3d.py:53-67``:  || Pi_{DG q} |u1 - uexact| ||_L2.
Basis tabulation comes from libseigen_hip's device-free ``sg_tabulate``.
"""
import math

import numpy as np
from scipy.special import roots_jacobi

from . import _lib
from .parallel import _dist


def _gj01(n, alpha):
    x, w = roots_jacobi(n, alpha, 0.0)
    return 0.5 * (x + 1.0), w / 2.0 ** (alpha + 1)


def simplex_rule(dim, degree):
    """Collapsed Gauss-Jacobi rule exact to `degree` on the unit simplex
    ([upstream] the family FIAT builds for a requested degree)."""
    m = max(1, (degree + 2) // 2)
    if dim == 1:
        t, w = _gj01(m, 0.0)
        return t[:, None], w
    if dim == 2:
        u, wu = _gj01(m, 1.0)
        s, ws = _gj01(m, 0.0)
        pts = np.array([(a, (1 - a) * b) for a in u for b in s])
        return pts, np.array([x * y for x in wu for y in ws])
    u, wu = _gj01(m, 2.0)
    s, ws = _gj01(m, 1.0)
    r, wr = _gj01(m, 0.0)
    pts = np.array([(a, (1 - a) * b, (1 - a) * (1 - b) * c) for a in u for b in s for c in r])
    return pts, np.array([x * y * z for x in wu for y in ws for z in wr])


def cell_rule(dim, degree, kind=0):
    """Quadrature on the reference cell: the simplex rule, or (kind 1, tensor-product cells) the Gauss-Legendre
    product rule exact to `degree` PER VARIABLE on the unit square / cube."""
    if kind == 0 or dim == 1:
        return simplex_rule(dim, degree)
    t, w = _gj01(max(1, degree // 2 + 1), 0.0)
    if dim == 2:
        pts = np.array([(a, b) for b in t for a in t])
        return pts, np.array([x * y for y in w for x in w])
    pts = np.array([(a, b, c) for c in t for b in t for a in t])
    return pts, np.array([x * y * z for z in w for y in w for x in w])


def tabulate(dim, degree, xi, kind=0):
    xi = np.ascontiguousarray(xi, dtype=np.float64).reshape(-1, dim)
    nd = {1: degree + 1, 2: (degree + 1) * (degree + 2) // 2,
          3: (degree + 1) * (degree + 2) * (degree + 3) // 6}[dim]
    if kind == 1 and dim > 1:
        nd = (degree + 1) ** dim
    phi = np.empty((xi.shape[0], nd))
    _lib.check(_lib.load().sg_tabulate_cell(kind, dim, degree, xi.shape[0], xi.ctypes.data, phi.ctypes.data))
    return phi


def _cell_volume_factor(mesh):
    """|det J| of every cell class (all equal on a structured mesh; the rules' weights sum to the measure of
    the reference cell)."""
    return float(np.prod(mesh.h))


def _global_sqrt(local_sq):
    dist = _dist()
    if dist is not None and dist.get_world_size() > 1:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([local_sq], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        local_sq = t.item()
    return math.sqrt(local_sq)


def norm(f):
    """L2 norm of a DG Function: sqrt(assemble(inner(f, f)*dx)) [upstream]."""
    space = f.function_space()
    dim, P, kind = space.dim, space.degree, space.mesh.cell_kind
    xq, wq = cell_rule(dim, 2 * P, kind)
    phi = tabulate(dim, P, xq, kind)
    M = np.einsum('q,qa,qb->ab', wq, phi, phi)
    v = f.dat.data_cells.reshape(space.ncells, space.nd, -1)
    return _global_sqrt(_cell_volume_factor(space.mesh) * np.einsum('cak,ab,cbk->', v, M, v))


def projected_abs_error_norm(f, exact, proj_degree):
    """|| Pi_{DG proj_degree} |f - exact| || for two Functions of one space
    (eigenmode_2d.py:49-55: proj_degree 6; eigenmode_3d.py:53-59: 3).
    [upstream] quadrature degree = proj_degree + degree (UFL: abs keeps the degree)."""
    space = f.function_space()
    dim, P, kind = space.dim, space.degree, space.mesh.cell_kind
    e = (f.dat.data_cells - exact.dat.data_cells).reshape(space.ncells, space.nd, -1)
    xq, wq = cell_rule(dim, proj_degree + P, kind)
    phi = tabulate(dim, P, xq, kind)
    psi = tabulate(dim, proj_degree, xq, kind)
    xm, wm = cell_rule(dim, 2 * proj_degree, kind)
    pm = tabulate(dim, proj_degree, xm, kind)
    Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wm, pm, pm))
    eq = np.abs(np.einsum('qa,cak->cqk', phi, e))
    b = np.einsum('q,qa,cqk->cak', wq, psi, eq)
    return _global_sqrt(_cell_volume_factor(space.mesh) * np.einsum('cak,ab,cbk->', b, Minv, b))
