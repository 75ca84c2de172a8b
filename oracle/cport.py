"""ctypes wrapper of oracle/c/seigen_oracle.c (plain C + OpenMP restatement of the step).
ORACLE / TEST INFRASTRUCTURE - see oracle/__init__.py.  The mesh tables (neighbours, facet node
matching, scaled normals) are derived here from the explicit-connectivity oracle mesh by matching
node coordinates, and the reference operators D_r = Mhat^-1 Shat_r, L_f = Mhat^-1 Mface_f come
from the oracle's own quadrature - nothing is taken from the product library."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

from . import refelem

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "c")
LIB = os.path.join(SRC_DIR, "libseigen_oracle.so")


class SoMesh(C.Structure):
    _fields_ = [("dim", C.c_int), ("nd", C.c_int), ("nf", C.c_int), ("nfaces", C.c_int), ("ncells", C.c_long),
                ("Jinv", C.c_void_p), ("cn", C.c_void_p), ("nbr", C.c_void_p), ("nbr_node", C.c_void_p),
                ("fnode", C.c_void_p), ("D", C.c_void_p), ("L", C.c_void_p)]


def build(arch=None, force=False):
    if force or not os.path.exists(LIB):
        cmd = ["make", "-C", SRC_DIR] + (["-B", "ARCH=%s" % arch] if arch else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB


def reference_operators(dim, P):
    nd = refelem.nnodes(dim, P)
    xq, wq = refelem.simplex_quadrature(dim, 2 * P)
    phi, dphi = refelem.tabulate(dim, P, xq)
    Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wq, phi, phi))
    D = np.stack([Minv @ np.einsum('q,qa,qb->ab', wq, dphi[:, :, r], phi) for r in range(dim)])
    nf = refelem.nnodes(dim - 1, P) if dim > 1 else 1
    xf, wf = refelem.simplex_quadrature(dim - 1, 2 * P)
    bary = np.concatenate([1 - xf.sum(1, keepdims=True), xf], axis=1)
    V = np.vstack([np.zeros(dim), np.eye(dim)])
    L = np.zeros((dim + 1, nd, nf))
    fnode = np.zeros((dim + 1, nf), dtype=np.int32)
    for f in range(dim + 1):
        fn = refelem.face_nodes(dim, P, f)
        fnode[f] = fn
        ph, _ = refelem.tabulate(dim, P, bary @ V[refelem.face_vertices(dim, f)])
        Mf = np.einsum('q,qa,qb->ab', wf * math.factorial(dim - 1), ph, ph)
        L[f] = (Minv @ Mf)[:, fn]
    return D, L, fnode


class CPort(object):
    def __init__(self, mesh, P):
        self.lib = C.CDLL(build())
        self.lib.so_max_threads.restype = C.c_int
        self.mesh, self.P = mesh, P
        d = self.dim = mesh.dim
        D, L, fnode = reference_operators(d, P)
        nd, nf, nfaces, nc = D.shape[1], L.shape[2], d + 1, mesh.ncells
        self.nd = nd
        X = mesh.node_coords(P)
        nbr = -np.ones((nc, nfaces), dtype=np.int64)
        nbr_node = np.zeros((nc, nfaces, nf), dtype=np.int32)
        for (c1, f1, c2, f2) in mesh.interior_facets:
            for (ca, fa, cb, fb) in ((c1, f1, c2, f2), (c2, f2, c1, f1)):
                nbr[ca, fa] = cb
                Xa = X[ca, fnode[fa]]
                Xb = X[cb, fnode[fb]]
                dist = np.abs(Xa[:, None, :] - Xb[None, :, :]).max(axis=2)
                j = dist.argmin(axis=1)
                assert dist[np.arange(nf), j].max() < 1e-10
                nbr_node[ca, fa] = fnode[fb][j]
        cn = np.zeros((nc, nfaces, d))
        cells = np.arange(nc)
        for f in range(nfaces):
            n_, area = mesh.facet_geometry(cells, np.full(nc, f))
            cn[:, f, :] = n_ * (area / np.abs(mesh.detJ))[:, None]
        self._keep = [np.ascontiguousarray(a) for a in (mesh.Jinv, cn, nbr, nbr_node, fnode, D, L)]
        m = SoMesh()
        m.dim, m.nd, m.nf, m.nfaces, m.ncells = d, nd, nf, nfaces, nc
        (m.Jinv, m.cn, m.nbr, m.nbr_node, m.fnode, m.D, m.L) = [a.ctypes.data for a in self._keep]
        self.m = m

    def threads(self):
        return int(self.lib.so_max_threads())

    def set_threads(self, n):
        self.lib.so_set_threads(C.c_int(int(n)))

    def apply_F(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        out = np.empty((self.mesh.ncells, self.nd, self.dim))
        self.lib.so_apply_F(C.byref(self.m), T.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def apply_G(self, u, lam, mu):
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.empty((self.mesh.ncells, self.nd, self.dim, self.dim))
        self.lib.so_apply_G(C.byref(self.m), u.ctypes.data_as(C.c_void_p), C.c_double(lam), C.c_double(mu),
                            out.ctypes.data_as(C.c_void_p))
        return out

    def step(self, u, s, rho, dt, lam, mu, nsteps):
        u = np.ascontiguousarray(u, dtype=np.float64).copy()
        s = np.ascontiguousarray(s, dtype=np.float64).copy()
        w = [np.empty_like(u), np.empty_like(s), np.empty_like(u), np.empty_like(s)]
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.so_step(C.byref(self.m), p(u), p(s), p(w[0]), p(w[1]), p(w[2]), p(w[3]), C.c_double(rho),
                         C.c_double(dt), C.c_double(lam), C.c_double(mu), C.c_int(nsteps))
        return u, s
