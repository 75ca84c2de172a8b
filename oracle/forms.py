"""Assembly of the velocity / stress right-hand sides by quadrature.
ORACLE (test infrastructure) - see oracle/__init__.py.

Restates ``seigen/elastic.py:204-219`` (forms ``f`` and ``g``) and the
element-wise inverse mass of ``seigen/elastic.py:369-385`` the way Firedrake
executes them [upstream]: a cell integral over every cell, an interior-facet
integral over every interior facet (both restrictions '+' and '-'), an
exterior-facet integral over boundary facets, each by numerical quadrature of
tabulated basis functions, scatter-added into a global vector which is then
multiplied by the block-diagonal inverse mass matrix.

Because every term of ``f``/``g`` is linear in the coefficient and acts
component-wise through scalar-basis integrals, the assembly is stored as
sparse scalar matrices (N = cells * nd scalar DG nodes):

  K_j [ (c,a),(c,b) ] = int_c  d(phi_a)/dx_j  phi_b  dx          (cell, dx)
  Fc_j[ (c,a),(c',b)] = int_f  phi_a^c  1/2 phi_b^{c'}  n^c_j dS (interior, dS)
  Bc_j[ (c,a),(c,b) ] = int_f  phi_a  phi_b  n_j ds              (exterior, ds)
  M   [ (c,a),(c,b) ] = int_c  phi_a phi_b dx

and then, with test functions w = phi_a e_i and v = phi_a e_i (x) e_j:

  f(w; T)_{a,i}  = sum_j ( -K_j + Fc_j ) T_ij            elastic.py:206
                   - int sigma phi_a u_i                  elastic.py:207-208
  g(v; u)_{a,ij} = l  delta_ij sum_k (-K_k + Fc_k + Bc_k) u_k      elastic.py:213-214
                 + mu ( (-K_j + Fc_j + Bc_j) u_i                    elastic.py:214-216
                      + (-K_i + Fc_i + Bc_i) u_j )
                 + int phi_a S_ij                         elastic.py:217-218

(`avg(s0)*n('+')` with test `w('+')` and the '-' twin give each side of a
facet its own outward normal; `jump(v, n)` = v+ n+ + v- n- does the same for
``g``; no ``ds`` term in ``f`` means a traction-free boundary.)
"""
import numpy as np
import scipy.sparse as sp
from . import refelem


class ScalarOperators(object):
    """All scalar DG operators of one (mesh, degree)."""

    def __init__(self, mesh, P, qdeg_extra=0):
        self.mesh = mesh
        self.P = P
        d = self.dim = mesh.dim
        kind = self.kind = getattr(mesh, "kind", "simplex")
        nd = self.nd = refelem.el_nnodes(d, P, kind)
        nc = mesh.ncells
        self.N = nc * nd

        # ---- cell integrals (dx) ------------------------------------------------
        xq, wq = refelem.el_quadrature(d, 2 * P + qdeg_extra, kind)
        phi, dphi = refelem.el_tabulate(d, P, xq, kind)          # [q,a], [q,a,r]
        absdet = np.abs(mesh.detJ)
        Mref = np.einsum('q,qa,qb->ab', wq, phi, phi)
        self.Mref = Mref
        # physical gradients: d phi_a/dx_j = sum_r Jinv[r,j] dphi[a,r]
        gphi = np.einsum('crj,qar->cqaj', mesh.Jinv, dphi)     # [c,q,a,j]
        Kloc = np.einsum('q,c,cqaj,qb->jcab', wq, absdet, gphi, phi)   # [j,c,a,b]
        Mloc = absdet[:, None, None] * Mref[None]
        self.Minv_loc = np.linalg.inv(Mloc)                    # elastic.py:376-382
        self.Mloc = Mloc
        rows = (np.arange(nc)[:, None, None] * nd + np.arange(nd)[None, :, None]) + np.zeros((1, 1, nd), dtype=np.int64)
        cols = (np.arange(nc)[:, None, None] * nd + np.arange(nd)[None, None, :]) + np.zeros((1, nd, 1), dtype=np.int64)
        self._rows, self._cols = rows.ravel(), cols.ravel()

        def blockdiag(loc):
            return sp.csr_matrix((loc.ravel(), (self._rows, self._cols)), shape=(self.N, self.N))

        self._blockdiag = blockdiag
        self.M = blockdiag(Mloc)
        self.Minv = blockdiag(self.Minv_loc)
        self.K = [blockdiag(Kloc[j]) for j in range(d)]

        # ---- facet integrals (dS, ds) ---------------------------------------------
        # facet rule: weights of the facet's vertices at every point (barycentric on simplices, multilinear on the
        # faces of tensor cells), weights wf, and fact with  sum(wf) * fact = 1
        baryf, wf, fact = refelem.el_facet_rule(d, 2 * P + qdeg_extra, kind)
        nqf = len(wf)

        def trace_table(cells, faces, xphys):
            """phi of `cells` at physical points xphys [n,q,d] -> [n,q,nd]."""
            xi = np.einsum('nmi,nqi->nqm', mesh.Jinv[cells], xphys - mesh.v0[cells][:, None, :])
            ph, _ = refelem.el_tabulate(d, P, xi.reshape(-1, d), kind)
            return ph.reshape(len(cells), nqf, nd)

        def facet_points(cells, faces):
            X = np.empty((len(cells), nqf, d))
            for ff in range(refelem.el_nfaces(d, kind)):
                sel = np.nonzero(faces == ff)[0]
                if len(sel) == 0:
                    continue
                fv = refelem.el_face_vertices(d, ff, kind)
                V = mesh.vertices[mesh.cells[cells[sel]][:, fv]]      # [n, facet vertices, d]
                X[sel] = np.einsum('qv,nvi->nqi', baryf, V)
            return X

        Fc = [sp.csr_matrix((self.N, self.N)) for _ in range(d)]
        IF = mesh.interior_facets
        if len(IF):
            cp, fp, cm, fm = IF[:, 0], IF[:, 1], IF[:, 2], IF[:, 3]
            X = facet_points(cp, fp)
            np_, area = mesh.facet_geometry(cp, fp)
            nm_, area_m = mesh.facet_geometry(cm, fm)
            assert np.allclose(np_, -nm_) and np.allclose(area, area_m)
            php = trace_table(cp, fp, X)
            phm = trace_table(cm, fm, X)
            wts = wf[None, :] * (area * fact)[:, None]                # [n,q]
            blocks = {}
            for (tc, tph, tn) in ((cp, php, np_), (cm, phm, nm_)):        # test side
                for (uc, uph) in ((cp, php), (cm, phm)):                  # avg() picks 1/2 of each side
                    loc = 0.5 * np.einsum('nq,nqa,nqb->nab', wts, tph, uph)
                    r = (tc[:, None, None] * nd + np.arange(nd)[None, :, None]) + np.zeros((1, 1, nd), dtype=np.int64)
                    c = (uc[:, None, None] * nd + np.arange(nd)[None, None, :]) + np.zeros((1, nd, 1), dtype=np.int64)
                    for j in range(d):
                        Fc[j] = Fc[j] + sp.csr_matrix(((loc * tn[:, j][:, None, None]).ravel(), (r.ravel(), c.ravel())),
                                                      shape=(self.N, self.N))
        self.Fc = Fc

        Bc = [sp.csr_matrix((self.N, self.N)) for _ in range(d)]
        EF = mesh.exterior_facets
        if len(EF):
            ce, fe = EF[:, 0], EF[:, 1]
            X = facet_points(ce, fe)
            ne, area = mesh.facet_geometry(ce, fe)
            phe = trace_table(ce, fe, X)
            wts = wf[None, :] * (area * fact)[:, None]
            loc = np.einsum('nq,nqa,nqb->nab', wts, phe, phe)
            r = (ce[:, None, None] * nd + np.arange(nd)[None, :, None]) + np.zeros((1, 1, nd), dtype=np.int64)
            c = (ce[:, None, None] * nd + np.arange(nd)[None, None, :]) + np.zeros((1, nd, 1), dtype=np.int64)
            for j in range(d):
                Bc[j] = Bc[j] + sp.csr_matrix(((loc * ne[:, j][:, None, None]).ravel(), (r.ravel(), c.ravel())),
                                              shape=(self.N, self.N))
        self.Bc = Bc

        # assembled right-hand-side operators (before M^-1)
        self.AF = [(-self.K[j] + self.Fc[j]).tocsr() for j in range(d)]
        self.AG = [(-self.K[j] + self.Fc[j] + self.Bc[j]).tocsr() for j in range(d)]
        # with the element-wise inverse mass folded in (elastic.py:358-367)
        self.DF = [(self.Minv @ A).tocsr() for A in self.AF]
        self.DG = [(self.Minv @ A).tocsr() for A in self.AG]

    def absorption_matrix(self, sigma_nodes, sigma_degree):
        """Minv * int sigma phi_a phi_b dx with sigma a DG_q field given by its
        nodal values [nc, nd_q] (elastic.py:207-208; the DG4 sponge of
        tests/explosive_source/explosive_source_lf4.py:43-45)."""
        mesh, d, P = self.mesh, self.dim, self.P
        xq, wq = refelem.el_quadrature(d, 2 * P + sigma_degree, self.kind)
        phi, _ = refelem.el_tabulate(d, P, xq, self.kind)
        psi, _ = refelem.el_tabulate(d, sigma_degree, xq, self.kind)
        sig_q = np.einsum('qc,nc->nq', psi, np.asarray(sigma_nodes).reshape(mesh.ncells, -1))
        loc = np.einsum('q,n,nq,qa,qb->nab', wq, np.abs(mesh.detJ), sig_q, phi, phi)
        loc = np.einsum('nab,nbc->nac', self.Minv_loc, loc)
        return self._blockdiag(loc)


class ElasticOperators(object):
    """F and G acting on AoS fields u [nc, nd, d], T [nc, nd, d, d]."""

    def __init__(self, mesh, P):
        self.ops = ScalarOperators(mesh, P)
        self.mesh, self.P = mesh, P
        self.dim, self.nd = mesh.dim, self.ops.nd
        self.absorb = None          # sparse N x N or None
        self.source = None          # callable t -> S [nc, nd, d, d] or None

    def set_absorption(self, sigma_nodes, sigma_degree):
        self.absorb = self.ops.absorption_matrix(sigma_nodes, sigma_degree)

    def _flat(self, a):
        return a.reshape(self.ops.N, -1)

    def apply_F(self, T, u_abs=None):
        """uh = Minv f(w; T, u_abs)   (elastic.py:156-160, 204-209, 358-367)."""
        d = self.dim
        Tf = T.reshape(self.ops.N, d, d)
        out = np.zeros((self.ops.N, d))
        for i in range(d):
            for j in range(d):
                out[:, i] += self.ops.DF[j] @ Tf[:, i, j]
        if self.absorb is not None:
            out -= self.absorb @ u_abs.reshape(self.ops.N, d)
        return out.reshape(self.mesh.ncells, self.nd, d)

    def apply_G(self, u, lam, mu, S=None):
        """sh = Minv g(v; u, l, mu, source)  (elastic.py:162-166, 211-219).
        lam/mu: floats or per-cell arrays (build-defined extension: each cell
        scales its own g)."""
        d = self.dim
        nc, nd = self.mesh.ncells, self.nd
        uf = u.reshape(self.ops.N, d)
        W = np.empty((self.ops.N, d, d))      # W[:, i, k] = weak d u_i / d x_k
        for i in range(d):
            for k in range(d):
                W[:, i, k] = self.ops.DG[k] @ uf[:, i]
        tr = np.einsum('nkk->n', W)
        lam = np.broadcast_to(np.asarray(lam, dtype=np.float64).reshape(-1, 1), (nc, nd)).reshape(-1) \
            if np.ndim(lam) else lam
        mu = np.broadcast_to(np.asarray(mu, dtype=np.float64).reshape(-1, 1), (nc, nd)).reshape(-1) \
            if np.ndim(mu) else mu
        out = np.zeros((self.ops.N, d, d))
        ltr = lam * tr
        for i in range(d):
            out[:, i, i] += ltr
        sym = W + np.transpose(W, (0, 2, 1))
        if np.ndim(mu):
            out += mu[:, None, None] * sym
        else:
            out += mu * sym
        out = out.reshape(nc, nd, d, d)
        if S is not None:
            out = out + S
        return out
