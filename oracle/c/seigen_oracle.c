/*
 * seigen_oracle.c - plain C restatement of the explicit velocity-stress DG step.
 * ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker and the timed CPU
 * baseline ("port") of bench.py.  Nothing under seigen_amd/ links or loads this file.
 *
 * It restates, for an explicit-connectivity simplicial mesh handed over by oracle/cport.py,
 *   Minv f(w; T, u_abs)   seigen/elastic.py:204-209 + :358-367
 *   Minv g(v; u)          seigen/elastic.py:211-219 + :358-367
 *   the LF4 step          seigen/elastic.py:291-304, :340-352
 * in the affine reference-matrix form
 *   F_i  = -sum_r D_r (Jinv_rj T_ij) + sum_f L_f [ cn_f,j 1/2 (T+ + T-)_ij ]      (no boundary term)
 *   W_ik = -Jinv_rk (D_r u_i) + sum_f cn_f,k L_f u^_i ;  sh_ij = lam d_ij W_kk + mu (W_ij + W_ji)
 * with D_r = Mhat^-1 Shat_r and L_f = Mhat^-1 Mface_f computed by the numpy oracle's quadrature.
 * OpenMP over cells; no blocking, no vector intrinsics: a straightforward multi-core port.
 */
#include <stdlib.h>
#include <string.h>

typedef struct {
  int dim, nd, nf, nfaces;
  long ncells;
  const double* Jinv;    /* [cell][r][j] */
  const double* cn;      /* [cell][face][j]  = |F|/|detJ| * outward normal */
  const long* nbr;       /* [cell][face] neighbour cell, -1 on the boundary */
  const int* nbr_node;   /* [cell][face][nf] neighbour's element node at my facet node */
  const int* fnode;      /* [face][nf] my element node of each facet node */
  const double* D;       /* [r][a][b] */
  const double* L;       /* [face][a][b'] */
} so_mesh;

/* What the eigenmode set-up does not need (all optional, NULL / 0 = absent):
 *   sponge   Minv int sigma phi_a phi_b of the cells that carry sigma, as dense blocks computed by the numpy
 *            oracle's quadrature (elastic.py:207-208; explosive_source_lf4.py:43-45)
 *   source   nodal values of the stress source on its support, per step (elastic.py:217-218, :285-288)
 *   lam, mu  one value per cell (build-defined heterogeneous extension)
 *   rho      one value per cell; physical = 0: u1 = rho u0 + ..., 1: u1 = u0 + (...)/rho */
typedef struct {
  const double* lam;         /* [cell] or NULL */
  const double* mu;          /* [cell] or NULL */
  const double* rho;         /* [cell] or NULL */
  int rho_physical;
  const long* sponge_slot;   /* [cell] -> block or -1, or NULL */
  const double* sponge_B;    /* [block][a][b] */
  long src_nnz;              /* source nodes */
  const long* src_node;      /* [nnz] flat scalar node = cell * nd + a */
  const double* src_val;     /* [step][nnz][dim*dim] */
  long src_nsteps;           /* steps covered by src_val (no source afterwards) */
} so_extra;

#define MAXD 3
#define MAXND 125  /* nodes per cell: 35 on a P4 tetrahedron, 125 on a DQ_4 hexahedron */
#define MAXNF 25   /* facet nodes: 15 on a P4 triangle, 25 on a DQ_4 square (hexahedra) */

void so_apply_F_ex(const so_mesh* m, const double* T, const double* u_abs, const so_extra* ex, double* out) {
  const int d = m->dim, nd = m->nd, nf = m->nf, nfaces = m->nfaces, nc = d * d;
#pragma omp parallel for schedule(static)
  for (long c = 0; c < m->ncells; ++c) {
    const double* Tc = T + c * nd * nc;
    const double* Ji = m->Jinv + c * d * d;
    double Tt[MAXND][MAXD][MAXD]; /* [b][i][r] */
    for (int b = 0; b < nd; ++b)
      for (int i = 0; i < d; ++i)
        for (int r = 0; r < d; ++r) {
          double s = 0;
          for (int j = 0; j < d; ++j) s += Ji[r * d + j] * Tc[b * nc + i * d + j];
          Tt[b][i][r] = s;
        }
    double acc[MAXND][MAXD];
    for (int a = 0; a < nd; ++a)
      for (int i = 0; i < d; ++i) acc[a][i] = 0;
    for (int r = 0; r < d; ++r)
      for (int a = 0; a < nd; ++a) {
        const double* Dr = m->D + ((long)r * nd + a) * nd;
        for (int b = 0; b < nd; ++b) {
          double dv = Dr[b];
          for (int i = 0; i < d; ++i) acc[a][i] -= dv * Tt[b][i][r];
        }
      }
    for (int f = 0; f < nfaces; ++f) {
      long nb = m->nbr[c * nfaces + f];
      if (nb < 0) continue; /* traction-free: no ds term in f */
      const double* cnf = m->cn + (c * nfaces + f) * d;
      const double* Tn = T + nb * nd * nc;
      double fl[MAXNF][MAXD];
      for (int bp = 0; bp < nf; ++bp) {
        int on = m->fnode[f * nf + bp];
        int nn = m->nbr_node[(c * nfaces + f) * nf + bp];
        for (int i = 0; i < d; ++i) {
          double s = 0;
          for (int j = 0; j < d; ++j) s += 0.5 * (Tc[on * nc + i * d + j] + Tn[nn * nc + i * d + j]) * cnf[j];
          fl[bp][i] = s;
        }
      }
      for (int a = 0; a < nd; ++a) {
        const double* Lf = m->L + ((long)f * nd + a) * nf;
        for (int bp = 0; bp < nf; ++bp)
          for (int i = 0; i < d; ++i) acc[a][i] += Lf[bp] * fl[bp][i];
      }
    }
    if (ex && ex->sponge_slot && u_abs && ex->sponge_slot[c] >= 0) { /* - Minv int sigma phi u_abs */
      const double* B = ex->sponge_B + ex->sponge_slot[c] * nd * nd;
      const double* ua = u_abs + c * nd * d;
      for (int a = 0; a < nd; ++a)
        for (int b = 0; b < nd; ++b)
          for (int i = 0; i < d; ++i) acc[a][i] -= B[a * nd + b] * ua[b * d + i];
    }
    for (int a = 0; a < nd; ++a)
      for (int i = 0; i < d; ++i) out[(c * nd + a) * d + i] = acc[a][i];
  }
}

void so_apply_F(const so_mesh* m, const double* T, double* out) { so_apply_F_ex(m, T, 0, 0, out); }

/* srcv: this step's source values [nnz][dim*dim] or NULL */
void so_apply_G_ex(const so_mesh* m, const double* u, double lam0, double mu0, const so_extra* ex, const double* srcv,
                   double* out) {
  const int d = m->dim, nd = m->nd, nf = m->nf, nfaces = m->nfaces, nc = d * d;
#pragma omp parallel for schedule(static)
  for (long c = 0; c < m->ncells; ++c) {
    const double* uc = u + c * nd * d;
    const double* Ji = m->Jinv + c * d * d;
    double W[MAXND][MAXD][MAXD]; /* [a][i][k] */
    for (int a = 0; a < nd; ++a)
      for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) W[a][i][k] = 0;
    for (int r = 0; r < d; ++r)
      for (int a = 0; a < nd; ++a) {
        const double* Dr = m->D + ((long)r * nd + a) * nd;
        double R[MAXD] = {0, 0, 0};
        for (int b = 0; b < nd; ++b)
          for (int i = 0; i < d; ++i) R[i] += Dr[b] * uc[b * d + i];
        for (int i = 0; i < d; ++i)
          for (int k = 0; k < d; ++k) W[a][i][k] -= Ji[r * d + k] * R[i];
      }
    for (int f = 0; f < nfaces; ++f) {
      long nb = m->nbr[c * nfaces + f];
      const double* cnf = m->cn + (c * nfaces + f) * d;
      double fl[MAXNF][MAXD];
      for (int bp = 0; bp < nf; ++bp) {
        int on = m->fnode[f * nf + bp];
        for (int i = 0; i < d; ++i) {
          if (nb < 0) {
            fl[bp][i] = uc[on * d + i]; /* ds terms use the own trace */
          } else {
            int nn = m->nbr_node[(c * nfaces + f) * nf + bp];
            fl[bp][i] = 0.5 * (uc[on * d + i] + u[(nb * nd + nn) * d + i]);
          }
        }
      }
      for (int a = 0; a < nd; ++a) {
        const double* Lf = m->L + ((long)f * nd + a) * nf;
        double lu[MAXD] = {0, 0, 0};
        for (int bp = 0; bp < nf; ++bp)
          for (int i = 0; i < d; ++i) lu[i] += Lf[bp] * fl[bp][i];
        for (int i = 0; i < d; ++i)
          for (int k = 0; k < d; ++k) W[a][i][k] += cnf[k] * lu[i];
      }
    }
    const double lam = (ex && ex->lam) ? ex->lam[c] : lam0, mu = (ex && ex->mu) ? ex->mu[c] : mu0;
    for (int a = 0; a < nd; ++a) {
      double tr = 0;
      for (int k = 0; k < d; ++k) tr += W[a][k][k];
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
          out[(c * nd + a) * nc + i * d + j] = mu * (W[a][i][j] + W[a][j][i]) + (i == j ? lam * tr : 0.0);
    }
  }
  if (ex && srcv) /* Minv int phi S = S nodally: the source lives in the same space (elastic.py:217-218) */
    for (long z = 0; z < ex->src_nnz; ++z)
      for (int ij = 0; ij < nc; ++ij) out[ex->src_node[z] * nc + ij] += srcv[z * nc + ij];
}

void so_apply_G(const so_mesh* m, const double* u, double lam, double mu, double* out) {
  so_apply_G_ex(m, u, lam, mu, 0, 0, out);
}

static void axpy3(long n, double* y, double a, const double* x0, double b, const double* x1, double c, const double* x2) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) y[i] = a * x0[i] + b * x1[i] + c * x2[i];
}

/* nsteps LF4 steps without source / sponge (the eigenmode set-up).  u, s: state in / out;
 * uh, sh, u2, s2: work arrays of the sizes of u and s. */
void so_step(const so_mesh* m, double* u, double* s, double* uh, double* sh, double* u2, double* s2, double rho, double dt,
             double lam, double mu, int nsteps) {
  const long nu = m->ncells * m->nd * m->dim, ns = nu * m->dim;
  const double c3 = dt * dt * dt / 24.0;
  for (int k = 0; k < nsteps; ++k) {
    so_apply_F(m, s, uh);                 /* uh1   elastic.py:292 */
    so_apply_G(m, uh, lam, mu, sh);       /* stemp :293 */
    so_apply_F(m, sh, u2);                /* uh2   :294 */
    axpy3(nu, u, rho, u, dt, uh, c3, u2); /* u1    :295-296, :341-345 */
    so_apply_G(m, u, lam, mu, sh);        /* sh1   :300 */
    so_apply_F(m, sh, uh);                /* utemp :301 */
    so_apply_G(m, uh, lam, mu, s2);       /* sh2   :302 */
    axpy3(ns, s, 1.0, s, dt, sh, c3, s2); /* s1    :303-304, :348-352 */
  }
}

/* The same with sponge, source and per-cell material / density.  step0 = index of the first step in
 * ex->src_val.  u_abs of the two f applications of the velocity update is u0, of utemp it is u1
 * (elastic.py:157-160, :169-172, :187-190). */
void so_step_ex(const so_mesh* m, const so_extra* ex, double* u, double* s, double* uh, double* sh, double* u2, double* s2,
                double rho0, double dt, double lam, double mu, long step0, int nsteps) {
  const long nu = m->ncells * m->nd * m->dim, ns = nu * m->dim, per = (long)m->nd * m->dim;
  const double c3 = dt * dt * dt / 24.0;
  for (int k = 0; k < nsteps; ++k) {
    const double* sv = (ex && ex->src_nnz > 0 && step0 + k < ex->src_nsteps)
                           ? ex->src_val + (step0 + k) * ex->src_nnz * m->dim * m->dim : 0;
    so_apply_F_ex(m, s, u, ex, uh);             /* uh1 */
    so_apply_G_ex(m, uh, lam, mu, ex, sv, sh);  /* stemp */
    so_apply_F_ex(m, sh, u, ex, u2);            /* uh2 */
    if (ex && ex->rho) {
#pragma omp parallel for schedule(static)
      for (long i = 0; i < nu; ++i) {
        const double r = ex->rho[i / per];
        u[i] = ex->rho_physical ? u[i] + (dt * uh[i] + c3 * u2[i]) / r : r * u[i] + dt * uh[i] + c3 * u2[i];
      }
    } else if (ex && ex->rho_physical) {
      axpy3(nu, u, 1.0, u, dt / rho0, uh, c3 / rho0, u2);
    } else {
      axpy3(nu, u, rho0, u, dt, uh, c3, u2);
    }
    so_apply_G_ex(m, u, lam, mu, ex, sv, sh);   /* sh1 */
    so_apply_F_ex(m, sh, u, ex, uh);            /* utemp */
    so_apply_G_ex(m, uh, lam, mu, ex, sv, s2);  /* sh2 */
    axpy3(ns, s, 1.0, s, dt, sh, c3, s2);
  }
}

int so_max_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void so_set_threads(int n) {
#ifdef _OPENMP
  extern void omp_set_num_threads(int);
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
