"""Structured simplicial meshes: stand-ins for the Firedrake constructors the
reference's harnesses call (``UnitSquareMesh`` ``tests/eigenmode/eigenmode_2d.py:11``,
``UnitCubeMesh`` ``eigenmode_3d.py:11``, ``RectangleMesh``
``tests/explosive_source/explosive_source_lf4.py:10``, ``IntervalMesh``
``tests/pulse/pulse_1d_lf4.py``).

[upstream] these are nx*ny(*nz) squares (cubes) cut into 2 triangles (6 Kuhn
tetrahedra).  The mesh is never stored: cell/vertex numbering is implicit
(DESIGN.md "Mesh and numbering") and libseigen_hip derives neighbours from it.

Under ``torch.distributed`` (one process per GPU) the global mesh is split into
a Cartesian grid of blocks; each rank's solver owns one block (``Partition``).
"""
import numpy as np


def _factor_grid(world, dim, n):
    """Process grid for `world` ranks: the factorisation whose blocks have the smallest halo (facet-trace)
    surface per rank - 2 x 2 x 2 for eight ranks on a cube.  A block side normal to x costs no more
    than the others since the shell next to it is made of whole layout groups (csrc/handle.hpp shell_width_x;
    profiles/r03/launch_structure_one_device.txt); ties go to the grid with fewer cuts along x, then y (the x
    shell is 16 cubes thick, which leaves less of the block to the launch that overlaps the exchange)."""
    def grids(w, d):
        if d == 1:
            yield (w,)
            return
        for g in range(1, w + 1):
            if w % g == 0:
                for rest in grids(w // g, d - 1):
                    yield (g,) + rest

    best = None
    for grid in grids(world, dim):
        if any(grid[a] > n[a] for a in range(dim)):
            continue
        blk = [-(-n[a] // grid[a]) for a in range(dim)]                 # the largest block
        area = 0
        for a in range(dim):
            if grid[a] > 1:
                face = 1
                for b in range(dim):
                    if b != a:
                        face *= blk[b]
                area += face * (2 if grid[a] > 2 else 1)                 # an inner block has both sides
        key = (area,) + tuple(grid)
        if best is None or key < best[0]:
            best = (key, grid)
    if best is None:
        raise ValueError("more ranks than cells")
    return tuple(best[1])


class Partition(object):
    """Block of a Cartesian split of the global cell grid."""

    def __init__(self, n, rank=0, world=1, grid=None):
        self.dim = len(n)
        self.global_n = tuple(int(x) for x in n)
        self.world = world
        self.rank = rank
        self.grid = tuple(grid) if grid is not None else _factor_grid(world, self.dim, self.global_n)
        assert int(np.prod(self.grid)) == world, "process grid does not match world size"
        for a in range(self.dim):
            if self.grid[a] > self.global_n[a]:
                raise ValueError("more blocks than cells along axis %d" % a)
        self.coords = self.rank_to_coords(rank)
        self.start, self.n = [], []
        for a in range(self.dim):
            lo, hi = self._range(a, self.coords[a])
            self.start.append(lo)
            self.n.append(hi - lo)
        self.start, self.n = tuple(self.start), tuple(self.n)

    def _range(self, axis, c):
        n, g = self.global_n[axis], self.grid[axis]
        base, rem = divmod(n, g)
        lo = c * base + min(c, rem)
        return lo, lo + base + (1 if c < rem else 0)

    def rank_to_coords(self, rank):
        c = []
        for a in range(self.dim):
            c.append(rank % self.grid[a])
            rank //= self.grid[a]
        return tuple(c)

    def coords_to_rank(self, coords):
        r, mul = 0, 1
        for a in range(self.dim):
            r += coords[a] * mul
            mul *= self.grid[a]
        return r

    def neighbour(self, side):
        """Rank across block side (2*axis + hi), or None at the domain boundary."""
        axis, hi = side >> 1, side & 1
        c = list(self.coords)
        c[axis] += 1 if hi else -1
        if c[axis] < 0 or c[axis] >= self.grid[axis]:
            return None
        return self.coords_to_rank(c)

    @property
    def nbr_mask(self):
        m = 0
        for s in range(2 * self.dim):
            if self.neighbour(s) is not None:
                m |= 1 << s
        return m


class Mesh(object):
    def __init__(self, n, L, diagonal="left", quadrilateral=False):
        self.dim = len(n)
        self.n = tuple(int(x) for x in n)
        self.L = tuple(float(x) for x in L)
        if diagonal not in ("left", "right"):
            raise ValueError("diagonal must be 'left' or 'right'")
        if quadrilateral and self.dim == 1:
            raise ValueError("tensor-product cells need a 2-D (quadrilateral=True) or 3-D (hexahedral=True) mesh")
        self.diagonal = diagonal
        # [upstream] RectangleMesh / UnitSquareMesh(..., quadrilateral=True), and the hexahedral meshes extruded from
        # them: the squares / cubes are the cells and FunctionSpace(mesh, "DG", k) (seigen/elastic.py:81-82) becomes
        # the tensor-product element DQ_k.  `quadrilateral` is True for either (the flag means "tensor-product cells").
        self.quadrilateral = bool(quadrilateral)
        self.h = tuple(l / k for l, k in zip(self.L, self.n))
        self.origin = (0.0,) * self.dim
        self._partition = None

    @property
    def partition(self):
        """This rank's block (whole mesh when torch.distributed is not initialised)."""
        if self._partition is None:
            from .parallel import world
            rank, size = world()
            self._partition = Partition(self.n, rank, size)
        return self._partition

    def set_partition(self, partition):
        self._partition = partition

    def geometric_dimension(self):
        return self.dim

    @property
    def cells_per_block(self):
        return 1 if self.quadrilateral else {1: 1, 2: 2, 3: 6}[self.dim]

    @property
    def cell_kind(self):
        """0: simplices, 1: tensor-product cells (the `cell_type` of sg_tabulate_cell, include/seigen_hip.h)."""
        return 1 if self.quadrilateral else 0

    def num_cells(self):
        return int(np.prod(self.n)) * self.cells_per_block


def IntervalMesh(ncells, length):
    return Mesh((ncells,), (length,))


def UnitIntervalMesh(ncells):
    return Mesh((ncells,), (1.0,))


def RectangleMesh(nx, ny, Lx, Ly, diagonal="left", quadrilateral=False):
    return Mesh((nx, ny), (Lx, Ly), diagonal, quadrilateral)


def UnitSquareMesh(nx, ny, diagonal="left", quadrilateral=False):
    return Mesh((nx, ny), (1.0, 1.0), diagonal, quadrilateral)


def BoxMesh(nx, ny, nz, Lx, Ly, Lz, hexahedral=False):
    return Mesh((nx, ny, nz), (Lx, Ly, Lz), quadrilateral=hexahedral)


def UnitCubeMesh(nx, ny, nz, hexahedral=False):
    return Mesh((nx, ny, nz), (1.0, 1.0, 1.0), quadrilateral=hexahedral)
