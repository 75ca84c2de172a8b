"""Quadrilateral cells / tensor-product element DQ_k without a GPU: the oracle's restatement on quadrilaterals
(oracle/refelem.py el_*, oracle/mesh.py kind "tensor") and the library's device-free tables for that cell type
(sg_reference_operator_cell, sg_tabulate_cell, sg_mesh_tables / sg_block_node_coords with diagonal = 2) against
each other.  The reference holds no vectors for such meshes (its tests use triangles and tetrahedra only)."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import mesh as omesh, refelem
from oracle.forms import ElasticOperators
from oracle.harness import Eigenmode2D


def _ops(P):
    """D_r = Mhat^-1 Shat_r and L_f of the unit square by Gauss quadrature (the definitions of csrc/refelem.hpp)."""
    xq, wq = refelem.el_quadrature(2, 2 * P, "tensor")
    phi, dphi = refelem.el_tabulate(2, P, xq, "tensor")
    M = np.einsum('q,qa,qb->ab', wq, phi, phi)
    Minv = np.linalg.inv(M)
    D = np.stack([Minv @ np.einsum('q,qa,qb->ab', wq, dphi[:, :, r], phi) for r in range(2)])
    t, w = refelem.el_quadrature(1, 2 * P, "tensor")
    L, fn = [], []
    for f in range(4):
        axis, at = f // 2, float(f % 2)
        pts = np.zeros((len(t), 2))
        pts[:, axis] = at
        pts[:, 1 - axis] = t[:, 0]
        ph, _ = refelem.el_tabulate(2, P, pts, "tensor")
        nodes = refelem.el_face_nodes(2, P, f, "tensor")
        L.append(Minv @ np.einsum('q,qa,qb->ab', w, ph, ph[:, nodes]))
        fn.append(nodes)
    return M, D, np.stack(L), np.stack(fn)


@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_library_tables_of_the_tensor_product_element(P):
    from seigen_amd import _lib
    lib = _lib.load()
    nd, nf = (P + 1) ** 2, P + 1
    M, D, L, fn = _ops(P)

    def op(which, shape):
        out = np.empty(shape)
        n = lib.sg_reference_operator_cell(1, 2, P, which, 0, out.ctypes.data, out.nbytes)
        assert n == out.size, n
        return out
    scale = lambda a: np.abs(a).max()
    assert np.abs(op(2, (nd, nd)) - M).max() < 1e-14 * scale(M)
    assert np.abs(op(0, (2, nd, nd)) - D).max() < 1e-9 * scale(D)        # the check inverts M in double precision
    assert np.abs(op(1, (4, nd, nf)) - L).max() < 1e-9 * scale(L)
    assert np.array_equal(op(4, (4, nf)).astype(int), fn)
    xi = np.random.default_rng(P).uniform(0, 1, (17, 2))
    phi = np.empty((17, nd))
    assert lib.sg_tabulate_cell(1, 2, P, 17, xi.ctypes.data, phi.ctypes.data) == 0
    np.testing.assert_allclose(phi, refelem.el_tabulate(2, P, xi, "tensor")[0], atol=1e-12)
    # a simplex call with the tensor entry points, and an unsupported cell type
    assert lib.sg_tabulate_cell(0, 2, P, 17, xi.ctypes.data, np.empty((17, (P + 1) * (P + 2) // 2)).ctypes.data) == 0
    assert lib.sg_tabulate_cell(2, 2, P, 17, xi.ctypes.data, phi.ctypes.data) < 0
    assert lib.sg_reference_operator_cell(1, 1, P, 0, 0, None, 0) < 0     # no tensor cells in 1-D


def test_node_coordinates_and_neighbour_tables_of_a_quadrilateral_block():
    from seigen_amd import _lib
    lib = _lib.load()
    P, n, L = 3, (4, 3), (2.0, 1.5)
    cfg = _lib.SgConfig()
    cfg.dim, cfg.degree, cfg.diagonal = 2, P, 2
    for a in range(3):
        cfg.n[a] = n[a] if a < 2 else 1
        cfg.h[a] = L[a] / n[a] if a < 2 else 1.0
        cfg.origin[a] = (0.5, -1.0, 0.0)[a]
    X = np.empty((n[0] * n[1], (P + 1) ** 2, 2))
    assert lib.sg_block_node_coords(C.byref(cfg), P, X.ctypes.data, X.nbytes) == 0
    m = omesh.structured(2, n, L, origin=(0.5, -1.0), quadrilateral=True)
    np.testing.assert_allclose(X, m.node_coords(P), atol=1e-13)
    # facets: the oracle finds them from vertex ids, the library from the structure
    h = np.array([L[0] / n[0], L[1] / n[1], 1.0])
    nb = np.zeros((1, 4, 5), dtype=np.int32)
    nbn = np.zeros((1, 4, P + 1), dtype=np.int32)
    cn = np.zeros((1, 4, 3))
    jinv = np.zeros((1, 3, 3))
    assert lib.sg_mesh_tables(2, P, 2, h.ctypes.data, nb.ctypes.data, nbn.ctypes.data, cn.ctypes.data, jinv.ctypes.data) == 0
    assert m.nfaces == 4 and len(m.interior_facets) == (n[0] - 1) * n[1] + n[0] * (n[1] - 1)
    for (c1, f1, c2, f2) in m.interior_facets:
        for (c, f, co, fo) in ((c1, f1, c2, f2), (c2, f2, c1, f1)):
            axis, d, _, face, _ = nb[0, f]
            assert co == c + d * (1 if axis == 0 else n[0]) and face == fo
            nrm, area = m.facet_geometry(np.array([c]), np.array([f]))
            np.testing.assert_allclose(cn[0, f, :2], nrm[0] * area[0] / abs(m.detJ[c]), atol=1e-14)
            # matching nodes sit at the same place
            mine = refelem.el_face_nodes(2, P, f, "tensor")
            np.testing.assert_allclose(m.node_coords(P)[c, mine], m.node_coords(P)[co, nbn[0, f]], atol=1e-13)


def test_oracle_on_quadrilaterals_reproduces_polynomials_and_converges():
    m = omesh.structured(2, (3, 4), (1.5, 1.0), quadrilateral=True)
    for P in (1, 2, 3):
        E = ElasticOperators(m, P)
        X = m.node_coords(P)
        x, y = X[..., 0], X[..., 1]
        # a field in Q_P: continuous, so F is the strong divergence in the interior ... traction-free boundary aside
        u = np.stack([x ** P * y + 1.0, x - 2.0 * y ** P], axis=-1)
        W = E.apply_G(u, 0.0, 0.5)            # mu (grad u + grad u^T), exterior facets use the own trace: exact
        du = np.zeros(X.shape[:2] + (2, 2))
        du[..., 0, 0] = P * x ** (P - 1) * y
        du[..., 0, 1] = x ** P
        du[..., 1, 0] = 1.0
        du[..., 1, 1] = -2.0 * P * y ** (P - 1)
        np.testing.assert_allclose(W, 0.5 * (du + np.swapaxes(du, -1, -2)), atol=1e-10)
    errs = []
    for N in (4, 8):
        em = Eigenmode2D(N, 2, 0.5 * (1.0 / N) / 2.0, quadrilateral=True)
        u1, s1 = em.run(5.0)
        e = em.errors(u1, s1)
        errs.append((e["u_error"], e["s_error"]))
    assert math.log2(errs[0][0] / errs[1][0]) > 2.5 and math.log2(errs[0][1] / errs[1][1]) > 2.5


def test_oracle_quadrilateral_sweep_matches_committed_goldens():
    """tests/golden/eigenmode_errors.json "2d_quadrilateral" (make_golden.py eigen): the coarse rows, recomputed."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eigenmode_errors.json")))
    rows = {(r["P"], r["N"]): r for r in gold["2d_quadrilateral"]}
    assert len(rows) == 12
    for (P, N) in ((1, 8), (2, 8), (3, 4), (4, 4)):
        g = rows[(P, N)]
        em = Eigenmode2D(N, P, g["dt"], quadrilateral=True)
        u1, s1 = em.run(5.0)
        e = em.errors(u1, s1)
        assert abs(e["u_error"] - g["u_error"]) < 1e-12 and abs(e["s_error"] - g["s_error"]) < 1e-12


# ---- host layer on quadrilateral meshes (device-free entry points of the library only) ----------------------------
def test_box_projection_on_quadrilaterals():
    """source_mode='project' of the explosive-source harness: the L2 projection of the source box has the box's
    area on every mesh, equals the oracle's, and reproduces an indicator that the mesh resolves."""
    from oracle.harness import project_box_indicator as oracle_project
    from seigen_amd import Function, FunctionSpace, RectangleMesh
    from seigen_amd.functionspace import integral, project_box_indicator
    for P in (1, 2, 3):
        mesh = RectangleMesh(7, 5, 3.5, 2.0, quadrilateral=True)
        V = FunctionSpace(mesh, "DG", P)
        lo, hi = (0.8, 0.3), (2.1, 1.45)
        c = project_box_indicator(V, lo, hi)
        f = Function(V).assign(c)
        assert abs(float(integral(f)) - (hi[0] - lo[0]) * (hi[1] - lo[1])) < 1e-13
        m = omesh.structured(2, (7, 5), (3.5, 2.0), quadrilateral=True)
        np.testing.assert_allclose(c, oracle_project(m, P, lo, hi), atol=1e-12)
        c2 = project_box_indicator(V, (0.5, 0.4), (2.0, 1.2))              # cell boundaries: the indicator is in the space
        X = V.node_coords()
        inside = (X[..., 0].min(1) >= 0.5 - 1e-12) & (X[..., 0].max(1) <= 2.0 + 1e-12) & \
                 (X[..., 1].min(1) >= 0.4 - 1e-12) & (X[..., 1].max(1) <= 1.2 + 1e-12)
        np.testing.assert_allclose(c2, np.where(inside[:, None], 1.0, 0.0) * np.ones_like(c2), atol=1e-11)


def test_function_evaluation_and_integral_on_quadrilaterals():
    from seigen_amd import Function, UnitSquareMesh, VectorFunctionSpace
    from seigen_amd.expression import Expression
    from seigen_amd.functionspace import evaluate_at, integral
    mesh = UnitSquareMesh(4, 3, quadrilateral=True)
    U = VectorFunctionSpace(mesh, "DG", 3)
    f = Function(U).interpolate(Expression(("x[0]*x[0]*x[1]", "1 + x[0] - 2*x[1]*x[1]*x[1]")))
    for p in ((0.3, 0.7), (0.99, 0.01), (0.5, 1.0 / 3.0), (1.0, 1.0)):
        v = evaluate_at(f, p)
        np.testing.assert_allclose(v, [p[0] ** 2 * p[1], 1 + p[0] - 2 * p[1] ** 3], atol=1e-13)
    np.testing.assert_allclose(integral(f), [1.0 / 6.0, 1.0], atol=1e-13)


def test_vtu_output_of_a_quadrilateral_mesh(tmp_path):
    """Output path (seigen/elastic.py:221-232) on quadrilateral cells: VTK_QUAD cells with their own four vertices,
    counter-clockwise; the probe of uy.py:36-43 interpolates bilinearly."""
    from seigen_amd import Function, UnitSquareMesh, VectorFunctionSpace
    from seigen_amd.vtu import VtuStream, probe, read_vtu, vertex_nodes
    mesh = UnitSquareMesh(3, 2, quadrilateral=True)
    U = VectorFunctionSpace(mesh, "DG", 3)
    u = Function(U, name="VelocityNew")
    X = U.node_coords()
    vals = np.stack([X[..., 0] + 2 * X[..., 1], X[..., 0] * X[..., 1]], axis=-1)      # linear and bilinear
    u.dat.data = vals.reshape(-1, 2)
    st = VtuStream("velocity", directory=str(tmp_path))
    f = st.write(u, 0.25)
    text = (tmp_path / f).read_text()
    assert text.count(" 9") >= 6 or "9 9 9 9 9 9" in text
    pts, data = read_vtu(str(tmp_path / f))
    vn = vertex_nodes(2, 3, True)
    assert pts.shape == (6 * 4, 3)
    np.testing.assert_allclose(pts[:, :2], X[:, vn, :].reshape(-1, 2), atol=1e-15)
    corners = X[:, vn, :]
    e1, e2 = corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 1]
    cross = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    assert (cross > 0).all()                                                         # counter-clockwise
    xq = np.array([[0.31, 0.47], [0.5, 0.5], [1.0, 1.0], [0.0, 0.9]])
    got = probe(str(tmp_path / f), "VelocityNew", xq)
    np.testing.assert_allclose(got[:, 0], xq[:, 0] + 2 * xq[:, 1], atol=1e-13)
    np.testing.assert_allclose(got[:, 1], xq[:, 0] * xq[:, 1], atol=1e-13)           # bilinear: exact too


@pytest.mark.parametrize("seed", range(12))
def test_host_layer_point_evaluation_and_norms_on_random_meshes(seed):
    """Host-layer fuzz without a GPU: point location and evaluation of an interpolated polynomial at random points
    (cell edges and corners included) on triangles, quadrilaterals and tetrahedra; norm() and the projected-abs error
    functional of eigenmode_2d.py:49-63 against the oracle's on a random field."""
    from oracle.harness import l2_norm, projected_abs_norm
    from seigen_amd import BoxMesh, Function, RectangleMesh, VectorFunctionSpace
    from seigen_amd.expression import Expression
    from seigen_amd.functionspace import evaluate_at
    from seigen_amd.norms import norm, projected_abs_error_norm
    rng = np.random.default_rng(seed)
    dim = 2 if seed % 3 else 3
    quad = dim == 2 and seed % 2 == 0
    P = int(rng.integers(1, 5))
    n = tuple(int(x) for x in rng.integers(1, 6 if dim == 2 else 4, size=dim))
    L = tuple(float(x) for x in rng.uniform(0.5, 2.0, size=dim))
    if dim == 2:
        mesh = RectangleMesh(n[0], n[1], L[0], L[1], diagonal=("left", "right")[seed % 4 // 2], quadrilateral=quad)
        om = omesh.structured(2, n, L, ("left", "right")[seed % 4 // 2], quadrilateral=quad)
    else:
        mesh, om = BoxMesh(n[0], n[1], n[2], L[0], L[1], L[2]), omesh.structured(3, n, L)
    U = VectorFunctionSpace(mesh, "DG", P)
    # a polynomial of total degree <= P in every cell type's space
    c = rng.uniform(-1, 1, (dim, dim + 1))
    poly = lambda X, i: (c[i, 0] + sum(c[i, a + 1] * X[..., a] for a in range(dim))) ** P
    code = tuple("pow(%r + %s, %d)" % (float(c[i, 0]), " + ".join("%r*x[%d]" % (float(c[i, a + 1]), a) for a in range(dim)), P)
                 for i in range(dim))
    f = Function(U).interpolate(Expression(code))
    pts = rng.uniform(0, 1, (12, dim)) * np.asarray(L)
    grid = np.stack([rng.integers(0, n[a] + 1, 6) * (L[a] / n[a]) for a in range(dim)], axis=1)     # mesh vertices
    mixed = pts[:6].copy()
    mixed[:, 0] = grid[:, 0]                                                                        # points on grid lines
    for p in np.concatenate([pts, grid, mixed]):
        v = evaluate_at(f, p)
        assert v is not None, p
        np.testing.assert_allclose(v, [poly(p, i) for i in range(dim)], atol=1e-11)
    g = Function(U).assign(rng.uniform(-1, 1, (U.ncells, U.nd, dim)))
    z = Function(U).assign(np.zeros((U.ncells, U.nd, dim)))
    assert abs(norm(g) - l2_norm(om, P, g.dat.data_cells)) < 1e-12 * max(1.0, norm(g))
    q = 6 if dim == 2 else 3
    assert abs(projected_abs_error_norm(g, z, q) - projected_abs_norm(om, P, g.dat.data_cells, q)) < 1e-11
