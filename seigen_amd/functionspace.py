"""Function-space and `Function` stand-ins with the attributes the reference's
solver class and harnesses touch (``seigen/elastic.py:81-103``, ``:136-154``;
``tests/eigenmode/eigenmode_2d.py:30-36``): ``FunctionSpace``,
``VectorFunctionSpace``, ``TensorFunctionSpace``, ``Function.assign``,
``Function.interpolate``, ``Function.dat.data``, ``dof_count``.

Data layout follows [upstream] Firedrake: ``dat.data`` has shape
(nodes, *value_shape) with DG nodes numbered cell by cell; here additionally
exposed cell-blocked as ``dat.data_cells`` = (cells, nd, *value_shape).
"""
import ctypes as C
import numpy as np

from . import _lib
from .expression import Expression


def _nnodes(dim, degree, kind=0):
    if kind == 1 and dim > 1:
        return (degree + 1) ** dim
    return {1: degree + 1, 2: (degree + 1) * (degree + 2) // 2,
            3: (degree + 1) * (degree + 2) * (degree + 3) // 6}[dim]


def block_config(mesh, degree=1, device=0):
    """sg_config of this rank's block of `mesh`."""
    part = mesh.partition
    cfg = _lib.SgConfig()
    cfg.dim, cfg.degree = mesh.dim, degree
    for a in range(3):
        if a < mesh.dim:
            cfg.n[a] = part.n[a]
            cfg.h[a] = mesh.h[a]
            cfg.origin[a] = mesh.origin[a]
            cfg.cube0[a] = part.start[a]
        else:
            cfg.n[a], cfg.h[a], cfg.origin[a] = 1, 1.0, 0.0
    cfg.diagonal = 2 if mesh.quadrilateral else (1 if mesh.diagonal == "right" else 0)
    cfg.nbr_mask = part.nbr_mask
    cfg.device = device
    cfg.stream = None
    return cfg


class FunctionSpace(object):
    value_shape = ()

    def __init__(self, mesh, family, degree, name=None):
        if family == "DQ" and not mesh.quadrilateral:
            raise ValueError("family 'DQ' needs a quadrilateral mesh")
        if family not in ("DG", "Discontinuous Lagrange", "DQ"):
            raise NotImplementedError("seigen_amd implements the discontinuous-Galerkin path only "
                                      "(family='DG'); got %r" % (family,))
        if not (1 <= int(degree) <= 8):
            raise ValueError("degree must be 1..8")
        self.mesh = mesh
        self.family = "DG"
        self.degree = int(degree)
        self.name = name
        self.dim = mesh.dim
        self.nd = _nnodes(mesh.dim, self.degree, mesh.cell_kind)
        self._coords = None

    def ufl_element(self):
        return (self.family, self.degree, self.value_shape)

    @property
    def ncells(self):
        return int(np.prod(self.mesh.partition.n)) * self.mesh.cells_per_block

    @property
    def node_count(self):
        return self.ncells * self.nd

    @property
    def value_size(self):
        return int(np.prod(self.value_shape)) if self.value_shape else 1

    @property
    def dof_count(self):
        """nodes * value size on this rank (seigen/elastic.py:85)."""
        return self.node_count * self.value_size

    # spaces above this many nodes do not keep their coordinates (config 3: 55 M nodes = 1.3 GB per space)
    COORD_CACHE_MAX_NODES = 1 << 24

    def node_coords(self):
        """[cells, nd, dim] physical node coordinates of this rank's block."""
        if self._coords is not None:
            return self._coords
        cfg = block_config(self.mesh, min(self.degree, 4))
        out = np.empty((self.ncells, self.nd, self.dim))
        _lib.check(_lib.load().sg_block_node_coords(C.byref(cfg), self.degree, out.ctypes.data, out.nbytes))
        if self.node_count <= self.COORD_CACHE_MAX_NODES:
            self._coords = out
        return out


    def _chunk_specs(self, max_nodes):
        """(first cell, config of the slab, cells in it) over slabs of the block's last axis"""
        mesh, part = self.mesh, self.mesh.partition
        d = self.dim
        ncls = mesh.cells_per_block
        per_layer = int(np.prod(part.n[:d - 1])) * ncls if d > 1 else ncls
        layers = max(1, int(max_nodes // max(1, per_layer * self.nd)))
        for k0 in range(0, part.n[d - 1], layers):
            nl = min(layers, part.n[d - 1] - k0)
            cfg = block_config(mesh, min(self.degree, 4))
            cfg.n[d - 1] = nl
            cfg.cube0[d - 1] = part.start[d - 1] + k0
            yield k0 * per_layer, cfg, per_layer * nl

    def _chunk_coords(self, cfg, ncells):
        X = np.empty((ncells, self.nd, self.dim))
        _lib.check(_lib.load().sg_block_node_coords(C.byref(cfg), self.degree, X.ctypes.data, X.nbytes))
        return X

    def node_coords_chunks(self, max_nodes=1 << 22):
        """Yields (cell0, X[cells, nd, dim]) over slabs of the block's last axis, at most about
        `max_nodes` nodes at a time (a 128^3-cube P4 block has 440 M nodes = 10.6 GB of coordinates)."""
        for cell0, cfg, ncells in self._chunk_specs(max_nodes):
            yield cell0, self._chunk_coords(cfg, ncells)

    def map_chunks(self, fn, max_nodes=1 << 21):
        """Yields (cell0, fn(X)) slab by slab IN ORDER, the slabs' coordinates and fn evaluated by a pool of host threads
        (the coordinate builder and numpy's loops release the GIL): the set-up of a 128^3-cube block - 440 M nodes - is
        host-bound otherwise (interpolating a sponge Expression: 53 s on one thread).  SEIGEN_HOST_THREADS bounds the
        pool (default: the cores of the process, at most 16); at most pool + 2 slabs are alive at a time."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        nthreads = int(os.environ.get("SEIGEN_HOST_THREADS", "0")) or min(ncpu, 16)
        specs = list(self._chunk_specs(max_nodes))
        if nthreads <= 1 or len(specs) <= 1:
            for cell0, cfg, ncells in specs:
                yield cell0, fn(self._chunk_coords(cfg, ncells))
            return

        def work(spec):
            return fn(self._chunk_coords(spec[1], spec[2]))

        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            window, nxt = [], 0
            while nxt < len(specs) or window:
                while nxt < len(specs) and len(window) < nthreads + 2:
                    window.append((specs[nxt][0], pool.submit(work, specs[nxt])))
                    nxt += 1
                cell0, fut = window.pop(0)
                yield cell0, fut.result()


class VectorFunctionSpace(FunctionSpace):
    def __init__(self, mesh, family, degree, name=None):
        super(VectorFunctionSpace, self).__init__(mesh, family, degree, name)
        self.value_shape = (mesh.dim,)


class TensorFunctionSpace(FunctionSpace):
    def __init__(self, mesh, family, degree, name=None):
        super(TensorFunctionSpace, self).__init__(mesh, family, degree, name)
        self.value_shape = (mesh.dim, mesh.dim)


class _Dat(object):
    """``Function.dat``: ``data`` reads (and, by assignment, writes) the values."""

    def __init__(self, fn):
        self._fn = fn

    @property
    def data(self):
        return self._fn._get().reshape((-1,) + self._fn._space.value_shape)

    @data.setter
    def data(self, value):
        self._fn._set(np.asarray(value, dtype=np.float64))

    data_ro = data

    @property
    def data_cells(self):
        return self._fn._get()


class Function(object):
    """Host-resident unless bound to a device field of a solver (then reads
    download and ``assign``/``interpolate`` upload)."""

    def __init__(self, space, name=None):
        self._space = space
        self._name = name
        self._host = None        # host values, allocated on first use (a config-3 stress field is 4 GB)
        self._binding = None     # (HipBlock, field id)
        self.dat = _Dat(self)

    def name(self):
        return self._name

    def function_space(self):
        return self._space

    def _bind(self, block, field):
        self._binding = (block, field)
        self._host = None

    def _get(self):
        if self._binding is not None:
            block, field = self._binding
            return block.get_field(field)
        if self._host is None:
            self._host = np.zeros(self._shape())
        return self._host

    def _shape(self):
        return (self._space.ncells, self._space.nd) + self._space.value_shape

    def _set(self, arr):
        shape = (self._space.ncells, self._space.nd) + self._space.value_shape
        arr = np.ascontiguousarray(np.broadcast_to(arr.reshape(shape) if arr.size == int(np.prod(shape)) else arr,
                                                   shape), dtype=np.float64)
        if self._binding is not None:
            block, field = self._binding
            block.set_field(field, arr)
        else:
            self._host = arr.copy()

    def assign(self, other):
        if isinstance(other, Function):
            if other._space.ufl_element() != self._space.ufl_element():
                raise ValueError("assign between different function spaces")
            self._set(other._get())
        else:
            self._set(np.asarray(other, dtype=np.float64))
        return self

    def interpolate(self, expression):
        """Nodal interpolation [upstream]: evaluate at the node coordinates."""
        is_expr = isinstance(expression, Expression) or not callable(expression)
        if is_expr:
            scalar_ok = expression.value_shape == () and self._space.value_size == 1   # 1-D vector/tensor spaces
            if expression.value_shape != self._space.value_shape and not scalar_ok:
                raise ValueError("Expression shape %r does not match the function space %r"
                                 % (expression.value_shape, self._space.value_shape))
        space = self._space
        if space.node_count > space.COORD_CACHE_MAX_NODES:
            # large spaces: slab by slab, straight into the device field (bound) or the host array -
            # neither the coordinates nor the evaluation temporaries of the whole block ever exist
            if self._binding is None and self._host is None:
                self._host = np.empty(self._shape())
            def values(X):
                v = expression.evaluate(X) if is_expr else np.asarray(expression(X), dtype=np.float64)
                return np.ascontiguousarray(v, dtype=np.float64).reshape((X.shape[0], space.nd) + space.value_shape)

            # evaluated by a pool of host threads, handed over in order by this one (one host thread drives a handle)
            for cell0, v in space.map_chunks(values):
                if self._binding is not None:
                    block, field = self._binding
                    block.set_field_range(field, cell0, v)
                else:
                    self._host[cell0:cell0 + v.shape[0]] = v
            return self
        if is_expr:
            vals = expression.evaluate(space.node_coords())
            vals = vals.reshape(vals.shape[:2] + space.value_shape)
        else:
            vals = np.asarray(expression(space.node_coords()), dtype=np.float64)
        self._set(vals)
        return self

    def vector(self):
        return self.dat.data.ravel()


def locate(space, point):
    """(cell, xi) of the cell of this rank's block containing `point`, or None.
    Stand-in for the point location behind ``vtktools.vtu.ProbeData`` in the reference's
    receiver script (``tests/explosive_source/uy.py:36-43``)."""
    mesh, dim, P = space.mesh, space.dim, space.degree
    part = mesh.partition
    p = np.asarray(point, dtype=np.float64)[:dim]
    # candidate cubes: the one containing the point and, for a point on a grid line, the one
    # before it; a DG field is two-valued there and the lowest-numbered cell wins (deterministic)
    cands = [[]]
    for a in range(dim):
        t = (p[a] - mesh.origin[a]) / mesh.h[a]
        i = int(np.floor(t))
        opts = [i - 1, i] if abs(t - round(t)) < 1e-9 else [i]
        opts = [o - part.start[a] for o in opts if 0 <= o - part.start[a] < part.n[a]]
        cands = [c + [o] for c in cands for o in opts]
    ncls = mesh.cells_per_block
    X = space.node_coords()
    # lattice corners: nodes (0,..), (P,0,..), (0,P,..), (0,0,P)
    corner = {1: [0, P], 2: [0, P, space.nd - 1], 3: [0, P, (P + 1) * (P + 2) // 2 - 1, space.nd - 1]}[dim]
    if mesh.quadrilateral:
        # (0,0), (1,0), (0,1) of the unit square; (0,0,0), (1,0,0), (0,1,0), (0,0,1) of the unit cube
        corner = [0, P, P * (P + 1)] + ([P * (P + 1) ** 2] if dim == 3 else [])
    lins = []
    for cube in cands:
        lin, mul = 0, 1
        for a in range(dim):
            lin += cube[a] * mul
            mul *= part.n[a]
        lins.append(lin)
    for lin in sorted(lins):
        for k in range(ncls):
            cell = lin * ncls + k
            V = X[cell, corner]                       # [dim+1, dim]
            J = (V[1:] - V[0]).T
            xi = np.linalg.solve(J, p - V[0])
            if xi.min() >= -1e-12 and (xi.max() if mesh.quadrilateral else xi.sum()) <= 1.0 + 1e-12:
                return cell, xi
    return None


def evaluate_at(function, point, cell_xi=None):
    """Value of a DG Function at a point (owned by this rank), shape = value_shape."""
    from .norms import tabulate
    space = function.function_space()
    loc = locate(space, point) if cell_xi is None else cell_xi
    if loc is None:
        return None
    cell, xi = loc
    phi = tabulate(space.dim, space.degree, xi[None, :], space.mesh.cell_kind)[0]
    if function._binding is not None:
        block, field = function._binding
        vals = block.get_field_range(field, cell, 1)[0]
    else:
        vals = function._get()[cell]
    return np.tensordot(phi, vals, axes=(0, 0))


def _clip_polygon(poly, axis, bound, keep_below):
    """Sutherland-Hodgman: the part of the convex polygon `poly` [n, 2] with x[axis] <= bound
    (keep_below) or >= bound."""
    out = []
    n = len(poly)
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        ia = (a[axis] <= bound) if keep_below else (a[axis] >= bound)
        ib = (b[axis] <= bound) if keep_below else (b[axis] >= bound)
        if ia:
            out.append(a)
        if ia != ib:
            t = (bound - a[axis]) / (b[axis] - a[axis])
            out.append(a + t * (b - a))
    return np.array(out).reshape(-1, 2)


def project_box_indicator(space, lo, hi):
    """L2 projection onto the scalar DG space of the indicator function of the box [lo, hi] (2-D):
    per cell K, coefficients M_K^-1 int_{K n box} phi_a dx, the integral taken exactly (the clipped
    triangle is a convex polygon, fan-triangulated, collapsed Gauss-Jacobi rule of the basis degree).

    Not part of the reference, whose harness interpolates the indicator NODALLY
    (tests/explosive_source/explosive_source_lf4.py:36-40) - on a mesh coarser than the box the
    interpolant's integral then depends on which nodes happen to fall inside.  The projection's
    integral is the box area on every mesh: the source of the REF-C convergence study.
    Returns [cells, nd]; the local block of this rank."""
    from .norms import simplex_rule, tabulate
    if space.dim != 2:
        raise NotImplementedError("project_box_indicator: 2-D spaces")
    mesh, P, nd = space.mesh, space.degree, space.nd
    part = mesh.partition
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    if mesh.quadrilateral:
        # the cells are axis-parallel rectangles: K n box is a rectangle, integrated with a Gauss-Legendre
        # product rule exact for the basis (degree P per variable)
        from .norms import cell_rule
        xq, wq = cell_rule(2, P, 1)
        xm, wm = cell_rule(2, 2 * P, 1)
        pm = tabulate(2, P, xm, 1)
        Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wm, pm, pm))
        out = np.zeros((space.ncells, nd))
        h = np.asarray(mesh.h[:2])
        rng = []
        for a in range(2):
            i0 = int(np.floor((lo[a] - mesh.origin[a]) / mesh.h[a])) - part.start[a]
            i1 = int(np.floor((hi[a] - mesh.origin[a]) / mesh.h[a])) - part.start[a]
            rng.append(range(max(i0, 0), min(i1, part.n[a] - 1) + 1))
        for j in rng[1]:
            for i in rng[0]:
                c0 = np.array([mesh.origin[0] + (part.start[0] + i) * h[0], mesh.origin[1] + (part.start[1] + j) * h[1]])
                a0, a1 = np.maximum(lo, c0), np.minimum(hi, c0 + h)
                if (a1 - a0).min() <= 0:
                    continue
                xi = ((a0 + xq * (a1 - a0)) - c0) / h                  # reference points of the sub-rectangle's rule
                b = np.prod(a1 - a0) * (wq @ tabulate(2, P, xi, 1))
                out[j * part.n[0] + i] = Minv @ b / np.prod(h)
        return out
    xq, wq = simplex_rule(2, P)
    xm, wm = simplex_rule(2, 2 * P)
    pm = tabulate(2, P, xm)
    Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wm, pm, pm))      # reference mass, unit |det J|
    out = np.zeros((space.ncells, nd))
    X = space.node_coords()
    corner = [0, P, nd - 1]
    rng = []
    for a in range(2):
        i0 = int(np.floor((lo[a] - mesh.origin[a]) / mesh.h[a])) - part.start[a]
        i1 = int(np.floor((hi[a] - mesh.origin[a]) / mesh.h[a])) - part.start[a]
        rng.append(range(max(i0, 0), min(i1, part.n[a] - 1) + 1))
    for j in rng[1]:
        for i in rng[0]:
            for k in range(mesh.cells_per_block):
                cell = (j * part.n[0] + i) * mesh.cells_per_block + k
                V = X[cell, corner]
                poly = V.copy()
                for a in range(2):
                    if len(poly):
                        poly = _clip_polygon(poly, a, hi[a], True)
                    if len(poly):
                        poly = _clip_polygon(poly, a, lo[a], False)
                if len(poly) < 3:
                    continue
                J = (V[1:] - V[0]).T
                Jinv = np.linalg.inv(J)
                b = np.zeros(nd)
                for t in range(1, len(poly) - 1):
                    T = np.array([poly[0], poly[t], poly[t + 1]])
                    Jt = (T[1:] - T[0]).T
                    area2 = abs(np.linalg.det(Jt))
                    if area2 < 1e-300:
                        continue
                    xp = T[0] + xq @ Jt.T                              # physical quadrature points
                    xi = (xp - V[0]) @ Jinv.T
                    b += area2 * (wq @ tabulate(2, P, xi))
                out[cell] = Minv @ b / abs(np.linalg.det(J))
    return out


def integral(function):
    """int f dx of a DG Function, per value component (local block of this rank)."""
    from .norms import cell_rule, tabulate, _cell_volume_factor
    space = function.function_space()
    kind = space.mesh.cell_kind
    xq, wq = cell_rule(space.dim, space.degree, kind)
    w = wq @ tabulate(space.dim, space.degree, xq, kind)
    v = function.dat.data_cells
    return _cell_volume_factor(space.mesh) * np.tensordot(w, v.sum(axis=0), axes=(0, 0))
