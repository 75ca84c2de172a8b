"""Hexahedral cells / tensor-product element DQ_k in 3-D without a GPU: the oracle's restatement on cubes
(oracle/refelem.py el_*, oracle/mesh.py kind "tensor") and the library's device-free tables for that cell type
(sg_reference_operator_cell, sg_tabulate_cell, sg_mesh_tables / sg_block_node_coords with diagonal = 2, dim = 3)
against each other, and the host layer (function evaluation, norms, .vtu output).  The reference holds no vectors
for such meshes (its tests use triangles and tetrahedra only); `FunctionSpace(mesh, "DG", k)` of
seigen/elastic.py:81-82 on a hexahedral mesh is this element."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import mesh as omesh, refelem
from oracle.forms import ElasticOperators
from oracle.harness import Eigenmode3D, l2_norm


def _ops(P):
    """D_r = Mhat^-1 Shat_r and L_f of the unit cube by Gauss quadrature (the definitions of csrc/refelem.hpp)."""
    xq, wq = refelem.el_quadrature(3, 2 * P, "tensor")
    phi, dphi = refelem.el_tabulate(3, P, xq, "tensor")
    M = np.einsum('q,qa,qb->ab', wq, phi, phi)
    Minv = np.linalg.inv(M)
    D = np.stack([Minv @ np.einsum('q,qa,qb->ab', wq, dphi[:, :, r], phi) for r in range(3)])
    t, w = refelem.el_quadrature(2, 2 * P, "tensor")
    L, fn = [], []
    for f in range(6):
        axis, at = f // 2, float(f % 2)
        others = [a for a in range(3) if a != axis]
        pts = np.zeros((len(t), 3))
        pts[:, axis] = at
        pts[:, others[0]] = t[:, 0]
        pts[:, others[1]] = t[:, 1]
        ph, _ = refelem.el_tabulate(3, P, pts, "tensor")
        nodes = refelem.el_face_nodes(3, P, f, "tensor")
        L.append(Minv @ np.einsum('q,qa,qb->ab', w, ph, ph[:, nodes]))
        fn.append(nodes)
    return M, D, np.stack(L), np.stack(fn)


@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_library_tables_of_the_hexahedral_element(P):
    from seigen_amd import _lib
    lib = _lib.load()
    nd, nf = (P + 1) ** 3, (P + 1) ** 2
    M, D, L, fn = _ops(P)

    def op(which, shape):
        out = np.empty(shape)
        n = lib.sg_reference_operator_cell(1, 3, P, which, 0, out.ctypes.data, out.nbytes)
        assert n == out.size, n
        return out
    scale = lambda a: np.abs(a).max()
    assert np.abs(op(2, (nd, nd)) - M).max() < 1e-14 * scale(M)
    assert np.abs(op(0, (3, nd, nd)) - D).max() < 1e-9 * scale(D)
    assert np.abs(op(1, (6, nd, nf)) - L).max() < 1e-9 * scale(L)
    assert np.array_equal(op(4, (6, nf)).astype(int), fn)
    xi = np.random.default_rng(P).uniform(0, 1, (17, 3))
    phi = np.empty((17, nd))
    assert lib.sg_tabulate_cell(1, 3, P, 17, xi.ctypes.data, phi.ctypes.data) == 0
    np.testing.assert_allclose(phi, refelem.el_tabulate(3, P, xi, "tensor")[0], atol=1e-12)
    # the C port's operator tables (oracle/cport.py) are the same ones
    from oracle.cport import reference_operators
    D2, L2, fn2 = reference_operators(3, P, "tensor")
    np.testing.assert_allclose(D2, D, atol=1e-11)
    np.testing.assert_allclose(L2, L, atol=1e-11)
    assert np.array_equal(fn2, fn)


def test_node_coordinates_and_neighbour_tables_of_a_hexahedral_block():
    from seigen_amd import _lib
    lib = _lib.load()
    P, n, L = 2, (3, 2, 4), (1.5, 1.0, 3.0)
    org = (0.5, -1.0, 0.25)
    cfg = _lib.SgConfig()
    cfg.dim, cfg.degree, cfg.diagonal = 3, P, 2
    for a in range(3):
        cfg.n[a] = n[a]
        cfg.h[a] = L[a] / n[a]
        cfg.origin[a] = org[a]
    X = np.empty((n[0] * n[1] * n[2], (P + 1) ** 3, 3))
    assert lib.sg_block_node_coords(C.byref(cfg), P, X.ctypes.data, X.nbytes) == 0
    m = omesh.structured(3, n, L, origin=org, quadrilateral=True)
    np.testing.assert_allclose(X, m.node_coords(P), atol=1e-13)
    h = np.array([L[a] / n[a] for a in range(3)])
    nf = (P + 1) ** 2
    nb = np.zeros((1, 6, 5), dtype=np.int32)
    nbn = np.zeros((1, 6, nf), dtype=np.int32)
    cn = np.zeros((1, 6, 3))
    jinv = np.zeros((1, 3, 3))
    assert lib.sg_mesh_tables(3, P, 2, h.ctypes.data, nb.ctypes.data, nbn.ctypes.data, cn.ctypes.data, jinv.ctypes.data) == 0
    nint = (n[0] - 1) * n[1] * n[2] + n[0] * (n[1] - 1) * n[2] + n[0] * n[1] * (n[2] - 1)
    assert m.nfaces == 6 and len(m.interior_facets) == nint
    stride = (1, n[0], n[0] * n[1])
    XO = m.node_coords(P)
    for (c1, f1, c2, f2) in m.interior_facets:
        for (c, f, co, fo) in ((c1, f1, c2, f2), (c2, f2, c1, f1)):
            axis, d, _, face, _ = nb[0, f]
            assert co == c + d * stride[axis] and face == fo
            nrm, area = m.facet_geometry(np.array([c]), np.array([f]))
            np.testing.assert_allclose(cn[0, f], nrm[0] * area[0] / abs(m.detJ[c]), atol=1e-14)
            mine = refelem.el_face_nodes(3, P, f, "tensor")
            np.testing.assert_allclose(XO[c, mine], XO[co, nbn[0, f]], atol=1e-13)
    np.testing.assert_allclose(jinv[0], np.diag(1.0 / h), atol=1e-14)


def test_oracle_on_hexahedra_reproduces_polynomials_and_converges():
    m = omesh.structured(3, (2, 3, 2), (1.0, 1.5, 0.8), quadrilateral=True)
    for P in (1, 2):
        E = ElasticOperators(m, P)
        X = m.node_coords(P)
        x, y, z = X[..., 0], X[..., 1], X[..., 2]
        u = np.stack([x ** P * y + z, x - 2.0 * y ** P * z, x * y * z ** P], axis=-1)      # in Q_P
        W = E.apply_G(u, 0.0, 0.5)
        du = np.zeros(X.shape[:2] + (3, 3))
        du[..., 0, 0] = P * x ** (P - 1) * y
        du[..., 0, 1] = x ** P
        du[..., 0, 2] = 1.0
        du[..., 1, 0] = 1.0
        du[..., 1, 1] = -2.0 * P * y ** (P - 1) * z
        du[..., 1, 2] = -2.0 * y ** P
        du[..., 2, 0] = y * z ** P
        du[..., 2, 1] = x * z ** P
        du[..., 2, 2] = P * x * y * z ** (P - 1)
        np.testing.assert_allclose(W, 0.5 * (du + np.swapaxes(du, -1, -2)), atol=1e-10)
    errs = []
    for N in (4, 8):
        em = Eigenmode3D(N, 2, 0.5 * (1.0 / N) / 2.0, hexahedral=True)
        u1, s1 = em.run(1.0)                                  # (errors() compares at the reference's hard-coded t = 5)
        X = em.elastic.node_coords()
        errs.append((l2_norm(em.mesh, 2, u1 - em.u_exact(X, 1.0)),
                     l2_norm(em.mesh, 2, s1 - em.s_exact(X, 1.0 + em.elastic.dt / 2.0))))
    assert math.log2(errs[0][0] / errs[1][0]) > 2.0 and math.log2(errs[0][1] / errs[1][1]) > 2.0, errs


def test_function_evaluation_integral_and_norms_on_hexahedra():
    from seigen_amd import Function, UnitCubeMesh, VectorFunctionSpace
    from seigen_amd.expression import Expression
    from seigen_amd.functionspace import evaluate_at, integral
    from seigen_amd.norms import norm
    mesh = UnitCubeMesh(3, 2, 2, hexahedral=True)
    assert mesh.num_cells() == 12 and mesh.cell_kind == 1
    U = VectorFunctionSpace(mesh, "DG", 2)
    assert U.nd == 27
    f = Function(U).interpolate(Expression(("x[0]*x[0]*x[1]", "1 + x[0] - 2*x[1]*x[2]*x[2]", "x[0]*x[1]*x[2]")))
    for p in ((0.3, 0.7, 0.2), (0.99, 0.01, 0.5), (1.0 / 3.0, 0.5, 0.5), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)):
        v = evaluate_at(f, p)
        np.testing.assert_allclose(v, [p[0] ** 2 * p[1], 1 + p[0] - 2 * p[1] * p[2] ** 2, p[0] * p[1] * p[2]], atol=1e-13)
    np.testing.assert_allclose(integral(f), [1.0 / 6.0, 1.5 - 1.0 / 3.0, 0.125], atol=1e-13)
    g = Function(U).interpolate(Expression(("x[0]", "x[1]*x[2]", "1")))
    assert abs(norm(g) - math.sqrt(1.0 / 3.0 + 1.0 / 9.0 + 1.0)) < 1e-13


def test_vtu_output_of_a_hexahedral_mesh(tmp_path):
    """Output path (seigen/elastic.py:221-232) on hexahedra: VTK_HEXAHEDRON cells with their own eight vertices; the
    probe of uy.py:36-43 interpolates trilinearly."""
    from seigen_amd import Function, UnitCubeMesh, VectorFunctionSpace
    from seigen_amd.vtu import VtuStream, probe, read_vtu, vertex_nodes
    mesh = UnitCubeMesh(2, 2, 3, hexahedral=True)
    U = VectorFunctionSpace(mesh, "DG", 2)
    u = Function(U, name="VelocityNew")
    X = U.node_coords()
    vals = np.stack([X[..., 0] + 2 * X[..., 1] - X[..., 2], X[..., 0] * X[..., 1] * X[..., 2], X[..., 2]], axis=-1)
    u.dat.data = vals.reshape(-1, 3)
    st = VtuStream("velocity", directory=str(tmp_path))
    f = st.write(u, 0.25)
    text = (tmp_path / f).read_text()
    assert " ".join(["12"] * 12) in text
    pts, data = read_vtu(str(tmp_path / f))
    vn = vertex_nodes(3, 2, True)
    assert pts.shape == (12 * 8, 3)
    corners = X[:, vn, :]
    np.testing.assert_allclose(pts, corners.reshape(-1, 3), atol=1e-15)
    # VTK_HEXAHEDRON: 0-1-2-3 counter-clockwise seen from above (normal along +z), 4-7 the same face shifted up
    e1, e2 = corners[:, 1] - corners[:, 0], corners[:, 3] - corners[:, 0]
    assert (np.cross(e1, e2)[:, 2] > 0).all()
    np.testing.assert_allclose(corners[:, 4:, :2], corners[:, :4, :2], atol=1e-15)
    assert (corners[:, 4:, 2] > corners[:, :4, 2]).all()
    xq = np.array([[0.31, 0.47, 0.9], [0.5, 0.5, 0.5], [1.0, 1.0, 1.0], [0.0, 0.9, 0.1]])
    got = probe(str(tmp_path / f), "VelocityNew", xq)
    np.testing.assert_allclose(got[:, 0], xq[:, 0] + 2 * xq[:, 1] - xq[:, 2], atol=1e-13)
    np.testing.assert_allclose(got[:, 1], xq[:, 0] * xq[:, 1] * xq[:, 2], atol=1e-13)      # trilinear: exact too


def test_hexahedral_operator_tables_factorise():
    """What the sum-factorised kernels rely on (csrc/api.cpp checks the same at create): D_r acts along the lines of
    direction r only, with one 1-D matrix; L_f touches the facet node with the node's transverse indices only."""
    for P in (1, 2, 3, 4):
        n1 = P + 1
        M, D, L, fn = _ops(P)
        node = lambda a0, a1, a2: a0 + n1 * (a1 + n1 * a2)
        D1 = np.array([[D[0][node(m, 0, 0), node(n, 0, 0)] for n in range(n1)] for m in range(n1)])
        I = np.eye(n1)
        np.testing.assert_allclose(D[0], np.kron(I, np.kron(I, D1)), atol=1e-9 * np.abs(D1).max())
        np.testing.assert_allclose(D[1], np.kron(I, np.kron(D1, I)), atol=1e-9 * np.abs(D1).max())
        np.testing.assert_allclose(D[2], np.kron(D1, np.kron(I, I)), atol=1e-9 * np.abs(D1).max())
        for f in range(6):
            r, side = f // 2, f % 2
            lift = np.array([L[side][node(m, 0, 0), 0] for m in range(n1)])
            for a in range(n1 ** 3):
                ai = (a % n1, (a // n1) % n1, a // (n1 * n1))
                tr = [ai[m] for m in range(3) if m != r]
                want = np.zeros(n1 * n1)
                want[tr[0] + n1 * tr[1]] = lift[ai[r]]
                np.testing.assert_allclose(L[f][a], want, atol=1e-9 * np.abs(lift).max())
