"""VTK output of DG fields: the role of ``File("velocity.pvd") << u`` in the reference
(``seigen/elastic.py:117-124``, ``:221-232``; read back by
``tests/explosive_source/uy.py:33-43`` through ``vtktools.ProbeData``).

As [upstream] Firedrake does for a discontinuous field, every cell is written with its own
vertices (no sharing between cells) and the field is sampled at them: a piecewise-linear,
discontinuous picture of the P_k solution.  ``VtuStream("velocity")`` writes ``velocity_0.vtu``,
``velocity_1.vtu`` ... and the index ``velocity.pvd``; in a multi-process run every rank writes
its block as ``velocity_<k>_r<rank>.vtu``.
"""
import os

import numpy as np

_VTK_CELL = {1: 3, 2: 5, 3: 10}   # VTK_LINE, VTK_TRIANGLE, VTK_TETRA


def vertex_nodes(dim, degree, quadrilateral=False):
    """Indices of the cell's vertices among its lattice nodes (first reference coordinate fastest); for a
    quadrilateral in VTK_QUAD order (counter-clockwise), for a hexahedron in VTK_HEXAHEDRON order (the bottom face
    counter-clockwise, then the top face)."""
    k = int(degree)
    if quadrilateral and dim == 3:
        bottom = [0, k, (k + 1) * (k + 1) - 1, k * (k + 1)]
        return bottom + [b + k * (k + 1) ** 2 for b in bottom]
    if quadrilateral:
        return [0, k, (k + 1) * (k + 1) - 1, k * (k + 1)]
    if dim == 1:
        return [0, k]
    if dim == 2:
        return [0, k, (k + 1) * (k + 2) // 2 - 1]
    nd = (k + 1) * (k + 2) * (k + 3) // 6
    return [0, k, (k + 1) * (k + 2) // 2 - 1, nd - 1]


def _ascii(a, fmt):
    return "\n".join(" ".join(fmt % v for v in row) for row in a)


def write_vtu(path, points, point_data):
    """points [cells, nv, dim]; point_data {name: [cells, nv] or [cells, nv, ncomp]} (vectors are
    padded to 3 components, tensors to 3 x 3, as VTK wants)."""
    ncells, nv, dim = points.shape
    npts = ncells * nv
    P = np.zeros((npts, 3))
    P[:, :dim] = points.reshape(npts, dim)
    conn = np.arange(npts, dtype=np.int64).reshape(ncells, nv)
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">\n')
        f.write('<UnstructuredGrid>\n<Piece NumberOfPoints="%d" NumberOfCells="%d">\n' % (npts, ncells))
        f.write('<Points>\n<DataArray type="Float64" NumberOfComponents="3" format="ascii">\n')
        f.write(_ascii(P, "%.17g"))
        f.write('\n</DataArray>\n</Points>\n<Cells>\n<DataArray type="Int64" Name="connectivity" format="ascii">\n')
        f.write(_ascii(conn, "%d"))
        f.write('\n</DataArray>\n<DataArray type="Int64" Name="offsets" format="ascii">\n')
        f.write(" ".join("%d" % ((c + 1) * nv) for c in range(ncells)))
        f.write('\n</DataArray>\n<DataArray type="UInt8" Name="types" format="ascii">\n')
        # 9: VTK_QUAD, 12: VTK_HEXAHEDRON
        f.write(" ".join([str(9 if (dim == 2 and nv == 4) else 12 if (dim == 3 and nv == 8) else _VTK_CELL[dim])] * ncells))
        f.write('\n</DataArray>\n</Cells>\n<PointData>\n')
        for name, arr in point_data.items():
            a = np.asarray(arr, dtype=np.float64)
            if a.ndim == 2:                                   # scalar
                out = a.reshape(npts, 1)
            elif a.ndim == 3:                                 # vector
                out = np.zeros((npts, 3))
                out[:, :a.shape[2]] = a.reshape(npts, a.shape[2])
            else:                                             # tensor
                d = a.shape[2]
                out = np.zeros((npts, 3, 3))
                out[:, :d, :d] = a.reshape(npts, d, d)
                out = out.reshape(npts, 9)
            f.write('<DataArray type="Float64" Name="%s" NumberOfComponents="%d" format="ascii">\n' % (name, out.shape[1]))
            f.write(_ascii(out, "%.17g"))
            f.write('\n</DataArray>\n')
        f.write('</PointData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n')


def read_vtu(path):
    """(points [npts, 3], {name: array}) of a file written by write_vtu (tests, probing)."""
    import xml.etree.ElementTree as ET
    root = ET.parse(path).getroot()
    piece = root.find("UnstructuredGrid").find("Piece")
    pts = np.array(piece.find("Points").find("DataArray").text.split(), dtype=np.float64).reshape(-1, 3)
    data = {}
    for da in piece.find("PointData").findall("DataArray"):
        nc = int(da.get("NumberOfComponents"))
        data[da.get("Name")] = np.array(da.text.split(), dtype=np.float64).reshape(-1, nc)
    return pts, data


class VtuStream(object):
    """One ``File("<name>.pvd")`` of the reference: numbered .vtu files plus the .pvd index."""

    def __init__(self, name, rank=0, world=1, directory="."):
        self.name, self.rank, self.world, self.dir = name, rank, world, directory
        self.count = 0
        self.files = []

    def write(self, function, time=None):
        space = function.function_space()
        dim, degree = space.mesh.dim, space.degree
        vn = vertex_nodes(dim, degree, space.mesh.quadrilateral)
        X = space.node_coords()[:, vn, :]
        vals = function.dat.data_cells[:, vn]
        suffix = "_r%d" % self.rank if self.world > 1 else ""
        fname = "%s_%d%s.vtu" % (self.name, self.count, suffix)
        write_vtu(os.path.join(self.dir, fname), X, {function.name(): vals})
        self.files.append((self.count if time is None else time, fname))
        self.count += 1
        if self.rank == 0:
            with open(os.path.join(self.dir, self.name + ".pvd"), "w") as f:
                f.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n<Collection>\n')
                for t, fn in self.files:
                    f.write('<DataSet timestep="%s" file="%s"/>\n' % (t, fn))
                f.write('</Collection>\n</VTKFile>\n')
        return fname


def probe(path, name, xs):
    """Values of point data `name` at the points `xs` [n, dim] of a file written by write_vtu: the
    containing cell is searched among all cells, the value interpolated linearly between its vertices
    (what ``vtktools.ProbeData`` does in ``tests/explosive_source/uy.py:36-43``).  On a grid line
    the lowest-numbered containing cell answers."""
    pts, data = read_vtu(path)
    xs = np.atleast_2d(np.asarray(xs, dtype=np.float64))
    dim = xs.shape[1]
    import xml.etree.ElementTree as ET
    piece = ET.parse(path).getroot().find("UnstructuredGrid").find("Piece")
    types = [da for da in piece.find("Cells").findall("DataArray") if da.get("Name") == "types"][0].text.split()
    quad = dim == 2 and len(types) > 0 and types[0] == "9"           # VTK_QUAD: vertices counter-clockwise from the low corner
    hexa = dim == 3 and len(types) > 0 and types[0] == "12"          # VTK_HEXAHEDRON: bottom face, then top face
    nv = 4 if quad else 8 if hexa else dim + 1
    P = pts[:, :dim].reshape(-1, nv, dim)
    V = data[name].reshape(P.shape[0], nv, -1)
    out = np.empty((xs.shape[0], V.shape[2]))
    if quad or hexa:
        # affine cells: reference coordinates from the edges 0-1 and 0-3 (and 0-4), multilinear interpolation of the corners
        T = np.stack([P[:, 1, :] - P[:, 0, :], P[:, 3, :] - P[:, 0, :]] + ([P[:, 4, :] - P[:, 0, :]] if hexa else []),
                     axis=2)                                                           # [cells, dim, dim]
        Tinv = np.linalg.inv(T)
        for k, x in enumerate(xs):
            st = np.einsum("cij,cj->ci", Tinv, x[None, :] - P[:, 0, :])
            inside = (st.min(axis=1) >= -1e-12) & (st.max(axis=1) <= 1.0 + 1e-12)
            if not inside.any():
                raise ValueError("point %r is outside the mesh" % (tuple(x),))
            c = int(np.nonzero(inside)[0][0])
            a, b = st[c][:2]
            w = np.array([(1 - a) * (1 - b), a * (1 - b), a * b, (1 - a) * b])
            if hexa:
                w = np.concatenate([(1 - st[c][2]) * w, st[c][2] * w])
            out[k] = w @ V[c]
        return out
    # barycentric coordinates of every query point in every cell
    T = np.transpose(P[:, 1:, :] - P[:, :1, :], (0, 2, 1))          # [cells, dim, dim]
    Tinv = np.linalg.inv(T)
    for k, x in enumerate(xs):
        lam = np.einsum("cij,cj->ci", Tinv, x[None, :] - P[:, 0, :])
        lam0 = 1.0 - lam.sum(axis=1)
        inside = (lam.min(axis=1) >= -1e-12) & (lam0 >= -1e-12)
        if not inside.any():
            raise ValueError("point %r is outside the mesh" % (tuple(x),))
        c = int(np.nonzero(inside)[0][0])
        w = np.concatenate([[lam0[c]], lam[c]])
        out[k] = w @ V[c]
    return out
