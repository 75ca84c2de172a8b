"""Helpers for binding libseigen_hip from a host whose DG numbering differs:
the reference's Firedrake orders cells as DMPlex does and the nodes of a cell as FIAT does
[upstream]; libseigen_hip numbers cells cube-major / class-minor and nodes lattice-lexicographically
(DESIGN.md section 3).  One permutation, built once from node coordinates, translates."""
import numpy as np


def dg_permutation(ours, theirs, nd, decimals=9):
    """perm with ours[k] == theirs[perm[k]] for two [ncells*nd, dim] arrays holding the coordinates of the
    same DG nodes (every cell's nd nodes stored consecutively) in different cell and in-cell orders.

    DG nodes are duplicated wherever cells touch, so coordinates alone do not identify a node: cells
    are matched first (by their centroid = mean of their nodes, unique per cell), then the nodes
    inside each matched cell pair."""
    ours = np.asarray(ours, dtype=np.float64).reshape(-1, nd, np.shape(ours)[-1])
    theirs = np.asarray(theirs, dtype=np.float64).reshape(ours.shape)
    ncells, _, dim = ours.shape
    scale = max(float(np.abs(ours).max()), 1e-300)

    def keys(x):
        return np.round(x / scale, decimals) + 0.0          # + 0.0: no negative zeros

    def order(x):                                           # lexicographic row order
        return np.lexsort(tuple(x[:, a] for a in reversed(range(x.shape[1]))))

    co, ct = keys(ours.mean(axis=1)), keys(theirs.mean(axis=1))
    io, it = order(co), order(ct)
    if not np.array_equal(co[io], ct[it]):
        raise ValueError("the two node sets do not describe the same cells")
    cell_of = np.empty(ncells, dtype=np.int64)              # their cell for each of our cells
    cell_of[io] = it
    perm = np.empty(ncells * nd, dtype=np.int64)
    ko, kt = keys(ours), keys(theirs[cell_of])
    for c in range(ncells):
        jo, jt = order(ko[c]), order(kt[c])
        if not np.array_equal(ko[c][jo], kt[c][jt]):
            raise ValueError("cell %d: node coordinates differ" % c)
        loc = np.empty(nd, dtype=np.int64)
        loc[jo] = jt
        perm[c * nd:(c + 1) * nd] = cell_of[c] * nd + loc
    return perm
