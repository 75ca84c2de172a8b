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


def reference_operators(dim, P, kind="simplex"):
    """D_r = Mhat^-1 Shat_r, L_f = Mhat^-1 Mface_f restricted to the facet's nodes, facet node lists - by quadrature of
    the oracle's own basis (simplices, or kind "tensor": quadrilaterals, refelem.el_*)."""
    nd = refelem.el_nnodes(dim, P, kind)
    xq, wq = refelem.el_quadrature(dim, 2 * P, kind)
    phi, dphi = refelem.el_tabulate(dim, P, xq, kind)
    Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wq, phi, phi))
    D = np.stack([Minv @ np.einsum('q,qa,qb->ab', wq, dphi[:, :, r], phi) for r in range(dim)])
    nfaces = refelem.el_nfaces(dim, kind)
    bary, wf, fact = refelem.el_facet_rule(dim, 2 * P, kind)
    if kind == "tensor":
        nf = (P + 1) ** (dim - 1)
        V = np.array([[(v >> m) & 1 for m in range(dim)] for v in range(1 << dim)], dtype=np.float64)
    else:
        nf = refelem.nnodes(dim - 1, P) if dim > 1 else 1
        V = np.vstack([np.zeros(dim), np.eye(dim)])
    L = np.zeros((nfaces, nd, nf))
    fnode = np.zeros((nfaces, nf), dtype=np.int32)
    for f in range(nfaces):
        fn = refelem.el_face_nodes(dim, P, f, kind)
        fnode[f] = fn
        ph, _ = refelem.el_tabulate(dim, P, bary @ V[refelem.el_face_vertices(dim, f, kind)], kind)
        Mf = np.einsum('q,qa,qb->ab', wf * fact, ph, ph)
        L[f] = (Minv @ Mf)[:, fn]
    return D, L, fnode


def sponge_blocks(mesh, P, sigma_nodes, sigma_degree):
    """(slot [nc] -> block or -1, B [nblocks, nd, nd]) with B = M_K^-1 int sigma phi_a phi_b dx for the cells
    where the DG_q field sigma (nodal values [nc, nd_q]) is not identically zero: the blocks of
    oracle.forms.ScalarOperators.absorption_matrix (elastic.py:207-208), built for those cells only so that
    full-size meshes need no global sparse operators."""
    d, kind = mesh.dim, getattr(mesh, "kind", "simplex")
    sig = np.asarray(sigma_nodes, dtype=np.float64).reshape(mesh.ncells, -1)
    cells = np.nonzero(np.abs(sig).max(axis=1) > 0)[0]
    slot = -np.ones(mesh.ncells, dtype=np.int64)
    slot[cells] = np.arange(len(cells))
    xq, wq = refelem.el_quadrature(d, 2 * P + sigma_degree, kind)
    phi, _ = refelem.el_tabulate(d, P, xq, kind)
    psi, _ = refelem.el_tabulate(d, sigma_degree, xq, kind)
    xm, wm = refelem.el_quadrature(d, 2 * P, kind)
    pm, _ = refelem.el_tabulate(d, P, xm, kind)
    Minv = np.linalg.inv(np.einsum('q,qa,qb->ab', wm, pm, pm))          # |det J| cancels against the integral's
    sig_q = np.einsum('qc,nc->nq', psi, sig[cells])
    loc = np.einsum('q,nq,qa,qb->nab', wq, sig_q, phi, phi)
    return slot, np.einsum('ab,nbc->nac', Minv, loc)


class SoExtra(C.Structure):
    _fields_ = [("lam", C.c_void_p), ("mu", C.c_void_p), ("rho", C.c_void_p), ("rho_physical", C.c_int),
                ("sponge_slot", C.c_void_p), ("sponge_B", C.c_void_p), ("src_nnz", C.c_long), ("src_node", C.c_void_p),
                ("src_val", C.c_void_p), ("src_nsteps", C.c_long)]


class CPort(object):
    def __init__(self, mesh, P):
        # SEIGEN_ORACLE_LIB: another build of the same source (the sanitizer build of oracle/c/Makefile `asan`)
        self.lib = C.CDLL(os.environ.get("SEIGEN_ORACLE_LIB") or build())
        self.lib.so_max_threads.restype = C.c_int
        self.mesh, self.P = mesh, P
        d = self.dim = mesh.dim
        D, L, fnode = reference_operators(d, P, getattr(mesh, "kind", "simplex"))
        nd, nf, nfaces, nc = D.shape[1], L.shape[2], L.shape[0], mesh.ncells
        if nd > 125 or nf > 25:         # MAXND / MAXNF of seigen_oracle.c (its per-cell work arrays live on the stack)
            raise ValueError("oracle C port: at most 125 nodes per cell and 25 per facet")
        self.nd = nd
        X = mesh.node_coords(P)
        nbr = -np.ones((nc, nfaces), dtype=np.int64)
        nbr_node = np.zeros((nc, nfaces, nf), dtype=np.int32)
        IF = mesh.interior_facets
        for (A, FA, B, FB) in ((0, 1, 2, 3), (2, 3, 0, 1)):          # both sides of every interior facet
            nbr[IF[:, A], IF[:, FA]] = IF[:, B]
            for fa in range(nfaces):
                for fb in range(nfaces):
                    sel = np.nonzero((IF[:, FA] == fa) & (IF[:, FB] == fb))[0]
                    for lo in range(0, len(sel), 1 << 16):           # bounded temporaries
                        k = sel[lo:lo + (1 << 16)]
                        Xa = X[IF[k, A]][:, fnode[fa]]               # [n, nf, d]
                        Xb = X[IF[k, B]][:, fnode[fb]]
                        dist = np.abs(Xa[:, :, None, :] - Xb[:, None, :, :]).max(axis=3)
                        j = dist.argmin(axis=2)
                        assert np.take_along_axis(dist, j[:, :, None], axis=2).max() < 1e-10
                        nbr_node[IF[k, A], fa] = fnode[fb][j]
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

        self.extra = None

    def set_extra(self, lam=None, mu=None, rho=None, rho_physical=False, absorb=None, src_nodes=None, src_values=None,
                  sponge=None):
        """Optional ingredients of so_step_ex.  lam, mu, rho: one value per cell; absorb: the numpy oracle's
        block-diagonal Minv int sigma phi phi (ElasticOperators.absorb, scipy sparse) - its non-zero cell blocks
        are handed over densely (or sponge = (slot, B) from sponge_blocks); src_nodes [nnz] flat scalar nodes, src_values [nsteps, nnz, d, d]."""
        ex = SoExtra()
        keep = []

        def arr(a, dt=np.float64):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        nc, nd = self.mesh.ncells, self.nd
        if lam is not None:
            ex.lam = arr(np.broadcast_to(np.asarray(lam, dtype=np.float64).ravel(), (nc,)))
        if mu is not None:
            ex.mu = arr(np.broadcast_to(np.asarray(mu, dtype=np.float64).ravel(), (nc,)))
        if rho is not None:
            ex.rho = arr(np.broadcast_to(np.asarray(rho, dtype=np.float64).ravel(), (nc,)))
        ex.rho_physical = int(bool(rho_physical))
        if absorb is not None:
            A = absorb.tocsr()
            rows = np.nonzero(np.diff(A.indptr))[0]
            cells = np.unique(rows // nd)
            slot = -np.ones(nc, dtype=np.int64)
            slot[cells] = np.arange(len(cells))
            B = np.zeros((len(cells), nd, nd))
            coo = A.tocoo()
            assert (coo.row // nd == coo.col // nd).all()
            B[slot[coo.row // nd], coo.row % nd, coo.col % nd] = coo.data
            ex.sponge_slot, ex.sponge_B = arr(slot, np.int64), arr(B)
        if sponge is not None:
            ex.sponge_slot, ex.sponge_B = arr(sponge[0], np.int64), arr(sponge[1])
        if src_nodes is not None and len(src_nodes):
            src_values = np.asarray(src_values, dtype=np.float64).reshape(-1, len(src_nodes), self.dim * self.dim)
            ex.src_nnz, ex.src_nsteps = len(src_nodes), src_values.shape[0]
            ex.src_node, ex.src_val = arr(src_nodes, np.int64), arr(src_values)
        self._keep_extra = keep
        self.extra = ex

    def step_ex(self, u, s, rho, dt, lam, mu, nsteps, step0=0, inplace=False):
        """nsteps LF4 steps with everything set by set_extra (scalars lam / mu / rho where no array was given)."""
        if not inplace:
            u = np.ascontiguousarray(u, dtype=np.float64).copy()
            s = np.ascontiguousarray(s, dtype=np.float64).copy()
        w = self._work_arrays(u, s)
        self.work = w          # after the call: w[0] = utemp, w[1] = sh1 of the last step (the product's UH / SH buffers)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        ex = C.byref(self.extra) if self.extra is not None else None
        self.lib.so_step_ex(C.byref(self.m), ex, p(u), p(s), p(w[0]), p(w[1]), p(w[2]), p(w[3]), C.c_double(rho),
                            C.c_double(dt), C.c_double(lam), C.c_double(mu), C.c_long(step0), C.c_int(nsteps))
        return u, s

    def _work_arrays(self, u, s):
        """the four stage buffers of so_step / so_step_ex, kept between calls (a timed loop of one-step calls would
        otherwise page in 4 fresh arrays per call)"""
        w = getattr(self, "_work", None)
        if w is None or w[0].shape != u.shape or w[1].shape != s.shape:
            w = self._work = [np.empty_like(u), np.empty_like(s), np.empty_like(u), np.empty_like(s)]
        return w

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
        w = self._work_arrays(u, s)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.so_step(C.byref(self.m), p(u), p(s), p(w[0]), p(w[1]), p(w[2]), p(w[3]), C.c_double(rho),
                         C.c_double(dt), C.c_double(lam), C.c_double(mu), C.c_int(nsteps))
        return u, s
