"""Reference simplex: equispaced Lagrange P_k basis and collapsed Gauss-Jacobi
quadrature.  ORACLE (test infrastructure) - see oracle/__init__.py.

[upstream] Firedrake ``FunctionSpace(mesh, "DG", k)`` on simplices is FIAT's
DiscontinuousLagrange: the nodal basis of P_k at the principal (equispaced)
lattice.  The reference selects it at ``seigen/elastic.py:81-82``.  The node
ORDER inside a cell is this build's own convention (lattice-lexicographic,
first reference coordinate fastest); the discrete function is independent of it.
"""
import itertools
import numpy as np
from scipy.special import roots_jacobi


def lattice(dim, P):
    """Integer lattice points alpha (|alpha| <= P) of the P_k simplex,
    ordered with alpha_1 fastest.  Node coordinates are alpha / P."""
    pts = []
    if dim == 1:
        pts = [(a,) for a in range(P + 1)]
    elif dim == 2:
        pts = [(a1, a2) for a2 in range(P + 1) for a1 in range(P + 1 - a2)]
    elif dim == 3:
        pts = [(a1, a2, a3) for a3 in range(P + 1) for a2 in range(P + 1 - a3)
               for a1 in range(P + 1 - a3 - a2)]
    else:
        raise ValueError("dim must be 1, 2 or 3")
    return np.array(pts, dtype=np.int64)


def nnodes(dim, P):
    return len(lattice(dim, P))


def _psi(n, P, lam):
    """psi_n(lam) = prod_{s<n} (P lam - s)/(n - s) and its derivative."""
    val = np.ones_like(lam)
    der = np.zeros_like(lam)
    for s in range(n):
        f = (P * lam - s) / (n - s)
        df = P / (n - s)
        der = der * f + val * df
        val = val * f
    return val, der


def tabulate(dim, P, xi):
    """Lagrange basis at reference points xi [npts, dim].

    Closed form on the principal lattice: phi_alpha(lambda) =
    prod_m psi_{alpha_m}(lambda_m) with barycentrics lambda_0 = 1 - sum xi,
    lambda_m = xi_m and alpha_0 = P - |alpha|.
    Returns phi [npts, nd] and dphi [npts, nd, dim] (d/dxi)."""
    xi = np.atleast_2d(np.asarray(xi, dtype=np.float64))
    npts = xi.shape[0]
    lat = lattice(dim, P)
    nd = len(lat)
    lam = np.empty((npts, dim + 1))
    lam[:, 0] = 1.0 - xi.sum(axis=1)
    lam[:, 1:] = xi
    # psi tables: val[m][n], der[m][n]
    val = [[None] * (P + 1) for _ in range(dim + 1)]
    der = [[None] * (P + 1) for _ in range(dim + 1)]
    for m in range(dim + 1):
        for n in range(P + 1):
            val[m][n], der[m][n] = _psi(n, P, lam[:, m])
    phi = np.empty((npts, nd))
    dphi = np.empty((npts, nd, dim))
    for a, al in enumerate(lat):
        full = (P - int(al.sum()),) + tuple(int(x) for x in al)
        v = np.ones(npts)
        for m in range(dim + 1):
            v = v * val[m][full[m]]
        phi[:, a] = v
        # d/dlambda_m
        dl = []
        for m in range(dim + 1):
            t = der[m][full[m]].copy()
            for m2 in range(dim + 1):
                if m2 != m:
                    t = t * val[m2][full[m2]]
            dl.append(t)
        for r in range(dim):
            dphi[:, a, r] = dl[r + 1] - dl[0]
    return phi, dphi


def _gj01(n, alpha):
    """n-point Gauss-Jacobi rule on [0,1] for weight (1-t)^alpha."""
    x, w = roots_jacobi(n, alpha, 0.0)
    return 0.5 * (x + 1.0), w / 2.0 ** (alpha + 1)


def simplex_quadrature(dim, degree):
    """Collapsed (Stroud conical) Gauss-Jacobi rule on the unit simplex exact
    for polynomials of total degree <= ``degree``.  [upstream] FIAT's
    ``create_quadrature`` of that era builds the same family with
    m = ceil((degree+1)/2) points per direction.
    Returns points [nq, dim], weights [nq] (weights sum to 1/dim!)."""
    m = max(1, (degree + 2) // 2)
    if dim == 0:
        return np.zeros((1, 0)), np.ones(1)
    if dim == 1:
        t, w = _gj01(m, 0.0)
        return t[:, None], w
    if dim == 2:
        u, wu = _gj01(m, 1.0)
        s, ws = _gj01(m, 0.0)
        pts = np.array([(ui, (1 - ui) * sj) for ui in u for sj in s])
        wts = np.array([wi * wj for wi in wu for wj in ws])
        return pts, wts
    if dim == 3:
        u, wu = _gj01(m, 2.0)
        s, ws = _gj01(m, 1.0)
        r, wr = _gj01(m, 0.0)
        pts = np.array([(ui, (1 - ui) * sj, (1 - ui) * (1 - sj) * rk)
                        for ui in u for sj in s for rk in r])
        wts = np.array([wi * wj * wk for wi in wu for wj in ws for wk in wr])
        return pts, wts
    raise ValueError(dim)


def face_vertices(dim, f):
    """Local vertex numbers of face f (the facet opposite vertex f)."""
    return [v for v in range(dim + 1) if v != f]


def face_nodes(dim, P, f):
    """Element-node indices lying on face f (barycentric lambda_f == 0), in
    increasing element-node order."""
    lat = lattice(dim, P)
    lam0 = P - lat.sum(axis=1)
    if f == 0:
        mask = lam0 == 0
    else:
        mask = lat[:, f - 1] == 0
    return np.nonzero(mask)[0]


def node_ref_coords(dim, P):
    return lattice(dim, P).astype(np.float64) / P


# ---------------------------------------------------------------------------------------------------------------------
# Tensor-product cells (quadrilaterals, hexahedra).  [upstream] On a quadrilateral or hexahedral mesh
# ``FunctionSpace(mesh, "DG", k)`` - the call of seigen/elastic.py:81-82, which is family-agnostic - is DQ_k: the
# tensor product of interval DiscontinuousLagrange elements, i.e. the Lagrange basis at the (k+1)^dim equispaced
# lattice points of the unit square / cube.  Node order: first reference coordinate fastest (this build's convention,
# as for simplices).  Local vertex v sits at the corner whose coordinate m is bit m of v (2-D: 0 (0,0), 1 (1,0),
# 2 (0,1), 3 (1,1)); face 2m is x_m = 0, face 2m + 1 is x_m = 1; a face's vertices in ascending order.
# The functions below dispatch on `kind` ("simplex" / "tensor"); the simplex ones above are unchanged.
QUAD_FACE_VERTICES = [[0, 2], [1, 3], [0, 1], [2, 3]]


def tensor_face_vertices(dim, f):
    return [v for v in range(1 << dim) if ((v >> (f // 2)) & 1) == (f % 2)]


def el_lattice(dim, P, kind="simplex"):
    if kind == "simplex" or dim == 1:
        return lattice(dim, P)
    if dim == 2:
        return np.array([(a1, a2) for a2 in range(P + 1) for a1 in range(P + 1)], dtype=np.int64)
    return np.array([(a1, a2, a3) for a3 in range(P + 1) for a2 in range(P + 1) for a1 in range(P + 1)], dtype=np.int64)


def el_nnodes(dim, P, kind="simplex"):
    return len(el_lattice(dim, P, kind))


def el_nfaces(dim, kind="simplex"):
    return dim + 1 if kind == "simplex" else 2 * dim


def el_tabulate(dim, P, xi, kind="simplex"):
    if kind == "simplex" or dim == 1:
        return tabulate(dim, P, xi)
    xi = np.atleast_2d(np.asarray(xi, dtype=np.float64))
    lat = el_lattice(dim, P, kind)
    one = [tabulate(1, P, xi[:, m:m + 1]) for m in range(dim)]      # ([npts, P+1], [npts, P+1, 1]) per axis
    phi = np.ones((len(xi), len(lat)))
    dphi = np.ones((len(xi), len(lat), dim))
    for m in range(dim):
        v, dv = one[m][0][:, lat[:, m]], one[m][1][:, lat[:, m], 0]
        phi *= v
        for r in range(dim):
            dphi[:, :, r] *= dv if r == m else v
    return phi, dphi


def el_quadrature(dim, degree, kind="simplex"):
    """Exact for total degree <= `degree` on simplices; for degree <= `degree` PER VARIABLE on tensor cells
    (Gauss-Legendre products).  Weights sum to the reference cell's measure."""
    if kind == "simplex" or dim <= 1:
        return simplex_quadrature(dim, degree)
    t, w = _gj01(max(1, degree // 2 + 1), 0.0)
    if dim == 2:
        pts = np.array([(ti, tj) for tj in t for ti in t])
        wts = np.array([wi * wj for wj in w for wi in w])
    else:
        pts = np.array([(ti, tj, tk) for tk in t for tj in t for ti in t])
        wts = np.array([wi * wj * wk for wk in w for wj in w for wi in w])
    return pts, wts


def el_facet_rule(dim, degree, kind="simplex"):
    """(weights of the facet's vertices at every quadrature point [q, nfv], wf [q], fact): a facet integral is
    sum_q wf[q] * fact * measure(facet) * g(sum_v bary[q, v] X_v).  Simplices: the rule of the unit (dim-1)-simplex
    in barycentric form; tensor cells: the Gauss-Legendre product rule of the unit (dim-1)-cube with the multilinear
    vertex weights (vertices in ascending order, el_face_vertices)."""
    if kind == "simplex" or dim <= 2:
        xf, wf = simplex_quadrature(dim - 1, degree)
        bary = np.concatenate([1.0 - xf.sum(axis=1, keepdims=True), xf], axis=1)
        return bary, wf, float(np.prod(np.arange(1, dim)))
    xf, wf = el_quadrature(dim - 1, degree, kind)
    a, b = xf[:, 0], xf[:, 1]
    bary = np.stack([(1 - a) * (1 - b), a * (1 - b), (1 - a) * b, a * b], axis=1)
    return bary, wf, 1.0


def el_face_vertices(dim, f, kind="simplex"):
    if kind == "simplex":
        return face_vertices(dim, f)
    return tensor_face_vertices(dim, f)


def el_face_nodes(dim, P, f, kind="simplex"):
    if kind == "simplex":
        return face_nodes(dim, P, f)
    lat = el_lattice(dim, P, kind)
    return np.nonzero(lat[:, f // 2] == (P if f % 2 else 0))[0]


def el_node_ref_coords(dim, P, kind="simplex"):
    return el_lattice(dim, P, kind).astype(np.float64) / P
