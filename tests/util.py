"""Shared helpers for the parity tests: oracle <-> HIP plumbing."""
import numpy as np

from oracle import mesh as omesh
from oracle.lf4 import OracleLF4


def oracle_mesh(dim, n, L, diagonal="left"):
    if diagonal == "quadrilateral":       # the squares are the cells (tensor-product element)
        return omesh.structured(dim, n, L, quadrilateral=True)
    return omesh.structured(dim, n, L, diagonal)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    den = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / den


def seeded(shape, seed):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, size=shape)
