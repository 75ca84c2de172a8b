"""Simplicial meshes with EXPLICIT connectivity.  ORACLE (test infrastructure).

[upstream] ``UnitSquareMesh(N, N)`` / ``RectangleMesh(nx, ny, Lx, Ly)``
(``tests/eigenmode/eigenmode_2d.py:11``, ``tests/explosive_source/
explosive_source_lf4.py:10``) are nx*ny squares each cut into two triangles by
one fixed diagonal; ``UnitCubeMesh(N, N, N)`` (``eigenmode_3d.py:11``) is N^3
cubes each cut into six tetrahedra that share the cube's main diagonal (Kuhn /
Freudenthal split); ``IntervalMesh`` is n equal intervals.

The cell ORDER (cube-major with x fastest, class-minor) and the local vertex
order are this build's documented convention so fields can be compared array
to array with the HIP path; everything else here (facets, neighbours, normals)
is derived from vertex ids and coordinates with no structured-mesh shortcut.
"""
import itertools
import numpy as np
from . import refelem

# local vertex offsets (in units of cells) of the simplex classes inside one
# square / cube.  2-D "left" = Firedrake's default diagonal (from (i,j+1) to
# (i+1,j)); "right" = the (i,j)-(i+1,j+1) diagonal.
CLASSES_1D = [[(0,), (1,)]]
CLASSES_2D = {
    "left": [[(0, 0), (1, 0), (0, 1)], [(1, 1), (0, 1), (1, 0)]],
    "right": [[(0, 0), (1, 0), (1, 1)], [(0, 0), (1, 1), (0, 1)]],
}


def kuhn_classes():
    out = []
    for perm in itertools.permutations(range(3)):
        v = [np.zeros(3, dtype=int)]
        for ax in perm:
            w = v[-1].copy()
            w[ax] += 1
            v.append(w)
        out.append([tuple(int(c) for c in w) for w in v])
    return out


CLASSES_3D = kuhn_classes()


def class_table(dim, diagonal="left"):
    if dim == 1:
        return CLASSES_1D
    if dim == 2:
        return CLASSES_2D[diagonal]
    return CLASSES_3D


class Mesh(object):
    """vertices [nv, dim]; cells [nc, dim+1] vertex ids (kind "simplex"), or [nc, 2^dim] for the affine quadrilaterals
    / hexahedra of kind "tensor" (local vertex v at the corner whose coordinate m is bit m of v, refelem.py)."""

    def __init__(self, vertices, cells, kind="simplex"):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.cells = np.asarray(cells, dtype=np.int64)
        self.dim = self.vertices.shape[1]
        self.ncells = len(self.cells)
        self.kind = kind
        self.nfaces = refelem.el_nfaces(self.dim, kind)
        self._geometry()
        self._facets()

    def _geometry(self):
        d = self.dim
        X = self.vertices[self.cells]              # [nc, d+1, d]
        self.v0 = X[:, 0, :]
        # J[:, i, m] = d x_i / d xi_m
        if self.kind == "tensor":                  # affine: the edges from vertex 0 to the vertices 1, 2 (, 4)
            self.J = np.transpose(X[:, [1 << m for m in range(d)], :] - X[:, :1, :], (0, 2, 1))
        else:
            self.J = np.transpose(X[:, 1:d + 1, :] - X[:, :1, :], (0, 2, 1))
        self.detJ = np.linalg.det(self.J)
        self.Jinv = np.linalg.inv(self.J)          # [nc, m, i] = d xi_m / d x_i

    def _facets(self):
        """Interior facets (c1, f1, c2, f2) and exterior facets (c, f), found by matching the sorted vertex
        ids of every (cell, local face); listed in the order of the sorted vertex tuples, the two sides of an
        interior facet in ascending (cell, face) order."""
        d = self.dim
        nc = self.ncells
        nfc = self.nfaces
        fv = np.array([refelem.el_face_vertices(d, f, self.kind) for f in range(nfc)], dtype=np.int64).reshape(nfc, -1)
        keys = np.sort(self.cells[:, fv], axis=2).reshape(nc * nfc, -1)          # rows in (cell, face) order
        order = np.lexsort(keys.T[::-1])                                          # stable: ties keep (cell, face) order
        ks = keys[order]
        same = np.zeros(len(order), dtype=bool)
        same[:-1] = (ks[1:] == ks[:-1]).all(axis=1)                               # entry i and i + 1 share a facet
        if (same[:-1] & same[1:]).any():
            raise RuntimeError("non-manifold facet")
        first = np.nonzero(same)[0]
        paired = np.zeros(len(order), dtype=bool)
        paired[first] = True
        paired[first + 1] = True
        a, b = order[first], order[first + 1]
        self.interior_facets = np.stack([a // nfc, a % nfc, b // nfc, b % nfc], axis=1).astype(np.int64).reshape(-1, 4)
        e = order[~paired]
        self.exterior_facets = np.stack([e // nfc, e % nfc], axis=1).astype(np.int64).reshape(-1, 2)

    def facet_geometry(self, cell, f):
        """Outward unit normal and measure of local face f of the given cells
        (arrays), computed from vertex coordinates."""
        d = self.dim
        cell = np.asarray(cell)
        f = np.asarray(f)
        n = np.empty((len(cell), d))
        area = np.empty(len(cell))
        for ff in range(self.nfaces):
            sel = np.nonzero(f == ff)[0]
            if len(sel) == 0:
                continue
            fv = refelem.el_face_vertices(d, ff, self.kind)
            X = self.vertices[self.cells[cell[sel]]]      # [n, d+1, d]
            P0 = X[:, fv[0], :]
            opp = X[:, ff, :] if self.kind == "simplex" else X.mean(axis=1)   # a point on the inner side of the facet
            if d == 1:
                nn = np.ones((len(sel), 1))
                ar = np.ones(len(sel))
            elif d == 2:
                t = X[:, fv[1], :] - P0
                ar = np.linalg.norm(t, axis=1)
                nn = np.stack([t[:, 1], -t[:, 0]], axis=1) / ar[:, None]
            else:
                t1 = X[:, fv[1], :] - P0
                t2 = X[:, fv[2], :] - P0
                cr = np.cross(t1, t2)
                nrm = np.linalg.norm(cr, axis=1)
                ar = nrm if self.kind == "tensor" else 0.5 * nrm      # parallelogram / triangle
                nn = cr / nrm[:, None]
            # orient away from the opposite vertex
            sgn = np.sign(np.einsum('ij,ij->i', nn, P0 - opp))
            n[sel] = nn * sgn[:, None]
            area[sel] = ar
        return n, area

    def node_coords(self, P):
        """[nc, nd, dim] physical coordinates of the DG nodes."""
        xi = refelem.el_node_ref_coords(self.dim, P, self.kind)           # [nd, d]
        return self.v0[:, None, :] + np.einsum('cim,am->cai', self.J, xi)


def structured(dim, n, L, diagonal="left", origin=None, quadrilateral=False):
    """n = cells per axis (tuple), L = extents (tuple).  quadrilateral: the squares (2-D) or cubes (3-D, hexahedra)
    themselves are the cells ([upstream] ``UnitSquareMesh(N, N, quadrilateral=True)``, ``ExtrudedMesh`` of it)."""
    n = tuple(int(x) for x in n)
    L = tuple(float(x) for x in L)
    origin = (0.0,) * dim if origin is None else tuple(origin)
    if quadrilateral and dim == 1:
        raise ValueError("tensor-product cells: 2-D and 3-D")
    classes = ([[tuple((v >> m) & 1 for m in range(dim)) for v in range(1 << dim)]] if quadrilateral
               else class_table(dim, diagonal))
    nvx = [k + 1 for k in n]
    # vertex id with x fastest
    strides = [1]
    for a in range(1, dim):
        strides.append(strides[-1] * nvx[a - 1])
    grids = np.meshgrid(*[np.arange(k) for k in nvx], indexing='ij')
    nv = int(np.prod(nvx))
    vertices = np.zeros((nv, dim))
    ids = sum(g * s for g, s in zip(grids, strides)).ravel()
    for a in range(dim):
        # multiply before divide, as a plain "i*L/n" lattice
        vertices[ids, a] = origin[a] + grids[a].ravel() * (L[a] / n[a])
    # cube order: x fastest; cell = cube * ncls + class
    cube = np.stack(np.meshgrid(*[np.arange(k) for k in n], indexing='ij'), axis=-1)        # [n0, .., dim]
    cube = np.transpose(cube, tuple(reversed(range(dim))) + (dim,)).reshape(-1, dim)        # x fastest
    off = np.array(classes, dtype=np.int64)                                                 # [ncls, dim+1, dim]
    st = np.array(strides, dtype=np.int64)
    cells = ((cube[:, None, None, :] + off[None]) * st).sum(axis=-1).reshape(-1, off.shape[1])
    return Mesh(vertices, cells, "tensor" if quadrilateral else "simplex")


def UnitSquareMesh(nx, ny, diagonal="left", quadrilateral=False):
    return structured(2, (nx, ny), (1.0, 1.0), diagonal, quadrilateral=quadrilateral)


def RectangleMesh(nx, ny, Lx, Ly, diagonal="left", quadrilateral=False):
    return structured(2, (nx, ny), (Lx, Ly), diagonal, quadrilateral=quadrilateral)


def UnitCubeMesh(nx, ny, nz, hexahedral=False):
    return structured(3, (nx, ny, nz), (1.0, 1.0, 1.0), quadrilateral=hexahedral)


def BoxMesh(nx, ny, nz, Lx, Ly, Lz, hexahedral=False):
    return structured(3, (nx, ny, nz), (Lx, Ly, Lz), quadrilateral=hexahedral)


def IntervalMesh(n, L):
    return structured(1, (n,), (L,))
