"""Marmousi P-wave velocity model: the data file and lookup rule of ``seigen/marmousi.py``
(``data[floor(x/24)][-floor(y/24)]`` on a 384 x 122 grid of 24 m cells, ``:4-14``), wired into
the solver as per-cell material parameters (BASELINE config 5).

The reference's rule, read literally: with j = floor(y/24), Python's negative index ``-j`` is
row 122 - j for j >= 1 (row 121, the deepest, just above the bottom edge; depth counted down from
the surface) - and row 0, the SURFACE row, for j = 0 (``-0 == 0``): a one-row slip at the bottom
edge of the model.  ``rule="reference"`` reproduces that literally; the default ``rule="fixed"``
differs only there (j = 0 uses row 121 like j = 1).  The mesh of ``seigen/marmousi.py:16-21`` is
383 x 121 squares, so only j <= 120 occurs at cell centroids.

The reference script only writes the model to VTK and never feeds it to the solver; it defines
neither Vs nor density for it.  The build assumes Vs = Vp / sqrt(3) (Poisson solid) and a density
given by the caller (default 1; ``gardner_density`` offers the usual Vp-density relation), hence
mu = rho Vs^2, lambda = rho Vp^2 - 2 mu, applied per cell.
"""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "marmhard.dat")
NX, NY, H = 384, 122, 24.0


def load_model(path=DATA):
    """[384, 122] Vp in m/s (``numpy.loadtxt(path).reshape((384, 122))``, seigen/marmousi.py:5)."""
    return np.loadtxt(path).reshape((NX, NY))


def vp_at(data, x, y, rule="fixed"):
    """Nearest-cell lookup at physical points (``seigen/marmousi.py:8-11``), vectorised."""
    i = np.clip(np.floor(np.asarray(x) / H).astype(int), 0, NX - 1)
    j = np.clip(np.floor(np.asarray(y) / H).astype(int), 0, NY - 1)
    if rule == "reference":
        row = np.where(j == 0, 0, NY - j)          # data[i][-j]
    elif rule == "fixed":
        row = np.minimum(NY - j, NY - 1)
    else:
        raise ValueError("rule must be 'reference' or 'fixed'")
    return data[i, row]


def gardner_density(vp):
    """Gardner's relation rho = 0.31 Vp^0.25 (g/cm^3, Vp in m/s) in units of 1000 kg/m^3."""
    return 0.31 * np.asarray(vp, dtype=np.float64) ** 0.25


def cell_material(space, data=None, density=1.0, rule="fixed"):
    """(lambda, mu, vp) per cell of a function space's mesh block, from Vp at the cell centroids;
    `density`: a float, an array with one value per cell, or a callable vp -> rho."""
    data = load_model() if data is None else data
    X = space.node_coords().mean(axis=1)          # centroids of the equispaced lattice = cell centroids
    vp = vp_at(data, X[:, 0], X[:, 1], rule)
    rho = density(vp) if callable(density) else density
    vs = vp / np.sqrt(3.0)
    mu = rho * vs ** 2
    lam = rho * vp ** 2 - 2.0 * mu
    return lam, mu, vp
