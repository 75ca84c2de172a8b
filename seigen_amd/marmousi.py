"""Marmousi P-wave velocity model: the data file and lookup rule of ``seigen/marmousi.py``
(``data[floor(x/24)][-floor(y/24)]`` on a 384 x 122 grid of 24 m cells, ``:4-14``), wired into
the solver as per-cell Lame parameters (BASELINE config 5).

The reference script only writes the model to VTK and never feeds it to the solver; it also maps
row j = 0 to data[i][0] instead of the last row (``-0 == 0``) - here depth index j counts down from
the surface consistently (``data[i][121 - j]``).  The reference defines neither Vs nor density for
this model; the build assumes Vs = Vp / sqrt(3) (Poisson solid) and rho = 1, hence
mu = rho Vs^2, lambda = rho Vp^2 - 2 mu, applied per cell.
"""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "marmhard.dat")
NX, NY, H = 384, 122, 24.0


def load_model(path=DATA):
    """[384, 122] Vp in m/s (``numpy.loadtxt(path).reshape((384, 122))``, seigen/marmousi.py:5)."""
    return np.loadtxt(path).reshape((NX, NY))


def vp_at(data, x, y):
    """Nearest-cell lookup at physical points (y up, surface at y = NY*H)."""
    i = np.clip(np.floor(np.asarray(x) / H).astype(int), 0, NX - 1)
    j = np.clip(np.floor(np.asarray(y) / H).astype(int), 0, NY - 1)
    return data[i, NY - 1 - j]


def cell_material(space, data=None, density=1.0):
    """(lambda, mu) per cell of a function space's mesh block, from Vp at the cell centroids."""
    data = load_model() if data is None else data
    X = space.node_coords().mean(axis=1)          # centroids of the equispaced lattice = cell centroids
    vp = vp_at(data, X[:, 0], X[:, 1])
    vs = vp / np.sqrt(3.0)
    mu = density * vs ** 2
    lam = density * vp ** 2 - 2.0 * mu
    return lam, mu, vp
