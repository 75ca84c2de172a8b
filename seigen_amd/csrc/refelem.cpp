// Reference-element matrices by exact integration of monomials on the unit
// simplex, in long double.  Host-side setup code (runs once per sg_create).
#include "refelem.hpp"

#include <cmath>
#include <stdexcept>

namespace sg {

typedef long double real;

int num_nodes(int dim, int P) {
  if (dim == 0) return 1;
  if (dim == 1) return P + 1;
  if (dim == 2) return (P + 1) * (P + 2) / 2;
  return (P + 1) * (P + 2) * (P + 3) / 6;
}

void lattice_points(int dim, int P, std::vector<int>& out) {
  out.clear();
  if (dim == 0) return;
  if (dim == 1) {
    for (int a = 0; a <= P; ++a) out.push_back(a);
  } else if (dim == 2) {
    for (int a2 = 0; a2 <= P; ++a2)
      for (int a1 = 0; a1 <= P - a2; ++a1) {
        out.push_back(a1);
        out.push_back(a2);
      }
  } else {
    for (int a3 = 0; a3 <= P; ++a3)
      for (int a2 = 0; a2 <= P - a3; ++a2)
        for (int a1 = 0; a1 <= P - a3 - a2; ++a1) {
          out.push_back(a1);
          out.push_back(a2);
          out.push_back(a3);
        }
  }
}

static real factorial(int n) {
  real f = 1;
  for (int i = 2; i <= n; ++i) f *= i;
  return f;
}

// int over the unit dim-simplex of xi^g = prod g_i! / (|g| + dim)!
static real mono_integral(int dim, const int* g) {
  int s = 0;
  real num = 1;
  for (int i = 0; i < dim; ++i) {
    s += g[i];
    num *= factorial(g[i]);
  }
  return num / factorial(s + dim);
}

// in-place Gauss-Jordan inverse with partial pivoting, n x n row-major
static void invert(std::vector<real>& A, int n) {
  std::vector<real> B(n * n, 0);
  for (int i = 0; i < n; ++i) B[i * n + i] = 1;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (fabsl(A[r * n + c]) > fabsl(A[piv * n + c])) piv = r;
    if (A[piv * n + c] == 0) throw std::runtime_error("singular matrix in reference element");
    if (piv != c)
      for (int k = 0; k < n; ++k) {
        std::swap(A[c * n + k], A[piv * n + k]);
        std::swap(B[c * n + k], B[piv * n + k]);
      }
    real d = 1 / A[c * n + c];
    for (int k = 0; k < n; ++k) {
      A[c * n + k] *= d;
      B[c * n + k] *= d;
    }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      real f = A[r * n + c];
      if (f == 0) continue;
      for (int k = 0; k < n; ++k) {
        A[r * n + k] -= f * A[c * n + k];
        B[r * n + k] -= f * B[c * n + k];
      }
    }
  }
  A.swap(B);
}

// Monomial coefficients of the Lagrange basis: phi_a = sum_m C[m*nd + a] xi^gamma_m,
// gamma_m = lattice point m (the lattice doubles as the exponent set of P_k).
static std::vector<real> lagrange_coeffs(int dim, int P, const std::vector<int>& lat) {
  int nd = num_nodes(dim, P);
  std::vector<real> V(nd * nd);
  for (int a = 0; a < nd; ++a)
    for (int m = 0; m < nd; ++m) {
      real v = 1;
      for (int i = 0; i < dim; ++i) {
        real x = (real)lat[a * dim + i] / (real)P;
        for (int e = 0; e < lat[m * dim + i]; ++e) v *= x;
      }
      V[a * nd + m] = v;  // V[a][m] = mono_m(node_a)
    }
  invert(V, nd);  // V C = I  ->  C = V^-1, C[m][a]
  return V;
}

struct MassPack {
  std::vector<real> M, Minv;
};

static MassPack mass_matrix(int dim, int P) {
  int nd = num_nodes(dim, P);
  MassPack mp;
  mp.M.assign(nd * nd, 0);
  if (dim == 0) {
    mp.M[0] = 1;
    mp.Minv = mp.M;
    return mp;
  }
  std::vector<int> lat;
  lattice_points(dim, P, lat);
  std::vector<real> C = lagrange_coeffs(dim, P, lat);
  // Mmono[m][n]
  std::vector<real> Mm(nd * nd);
  int g[3];
  for (int m = 0; m < nd; ++m)
    for (int n = 0; n < nd; ++n) {
      for (int i = 0; i < dim; ++i) g[i] = lat[m * dim + i] + lat[n * dim + i];
      Mm[m * nd + n] = mono_integral(dim, g);
    }
  // M = C^T Mm C
  std::vector<real> T(nd * nd, 0);
  for (int m = 0; m < nd; ++m)
    for (int b = 0; b < nd; ++b) {
      real s = 0;
      for (int n = 0; n < nd; ++n) s += Mm[m * nd + n] * C[n * nd + b];
      T[m * nd + b] = s;
    }
  for (int a = 0; a < nd; ++a)
    for (int b = 0; b < nd; ++b) {
      real s = 0;
      for (int m = 0; m < nd; ++m) s += C[m * nd + a] * T[m * nd + b];
      mp.M[a * nd + b] = s;
    }
  mp.Minv = mp.M;
  invert(mp.Minv, nd);
  return mp;
}

static int lattice_index(int dim, int P, const std::vector<int>& lat, const int* al) {
  int nd = num_nodes(dim, P);
  for (int a = 0; a < nd; ++a) {
    bool eq = true;
    for (int i = 0; i < dim; ++i) eq = eq && (lat[a * dim + i] == al[i]);
    if (eq) return a;
  }
  return -1;
}

RefElem make_refelem(int dim, int P) {
  if (dim < 1 || dim > 3 || P < 1 || P > 4) throw std::runtime_error("dim must be 1..3 and degree 1..4");
  RefElem re;
  re.dim = dim;
  re.P = P;
  re.nd = num_nodes(dim, P);
  re.nf = num_nodes(dim - 1, P);
  re.nfaces = dim + 1;
  lattice_points(dim, P, re.lattice);
  const int nd = re.nd, nf = re.nf;
  std::vector<real> C = lagrange_coeffs(dim, P, re.lattice);
  MassPack mp = mass_matrix(dim, P);

  // Shat_r[a][b] = int d(phi_a)/d(xi_r) phi_b
  re.D.assign((size_t)dim * nd * nd, 0.0);
  int g[3];
  for (int r = 0; r < dim; ++r) {
    // Sm[m][n] = int d(mono_m)/dxi_r mono_n
    std::vector<real> Sm(nd * nd, 0);
    for (int m = 0; m < nd; ++m) {
      int gr = re.lattice[m * dim + r];
      if (gr == 0) continue;
      for (int n = 0; n < nd; ++n) {
        for (int i = 0; i < dim; ++i) g[i] = re.lattice[m * dim + i] + re.lattice[n * dim + i];
        g[r] -= 1;
        Sm[m * nd + n] = (real)gr * mono_integral(dim, g);
      }
    }
    std::vector<real> T(nd * nd, 0), S(nd * nd, 0);
    for (int m = 0; m < nd; ++m)
      for (int b = 0; b < nd; ++b) {
        real s = 0;
        for (int n = 0; n < nd; ++n) s += Sm[m * nd + n] * C[n * nd + b];
        T[m * nd + b] = s;
      }
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b) {
        real s = 0;
        for (int m = 0; m < nd; ++m) s += C[m * nd + a] * T[m * nd + b];
        S[a * nd + b] = s;
      }
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b) {
        real s = 0;
        for (int k = 0; k < nd; ++k) s += mp.Minv[a * nd + k] * S[k * nd + b];
        re.D[((size_t)r * nd + a) * nd + b] = (double)s;
      }
  }
  re.Mhat.resize(nd * nd);
  re.Minv.resize(nd * nd);
  for (int i = 0; i < nd * nd; ++i) {
    re.Mhat[i] = (double)mp.M[i];
    re.Minv[i] = (double)mp.Minv[i];
  }

  // facets: face f is opposite vertex f; its nodes have barycentric lambda_f = 0.
  // The restriction of the P_k Lagrange basis to a facet is the (dim-1)-simplex P_k
  // Lagrange basis on the facet's own lattice, so Mface_unit = (dim-1)! * Mhat^{(dim-1)}.
  MassPack fm = mass_matrix(dim - 1, P);
  std::vector<int> flat;
  lattice_points(dim - 1, P, flat);
  real fact = factorial(dim - 1);
  re.fnode.assign((size_t)re.nfaces * nf, -1);
  re.L.assign((size_t)re.nfaces * nd * nf, 0.0);
  for (int f = 0; f <= dim; ++f) {
    // element nodes on this face, in increasing element-node order
    std::vector<int> nodes, fidx;  // fidx: index in the (dim-1) lattice
    for (int a = 0; a < nd; ++a) {
      int full[4];
      int s = 0;
      for (int i = 0; i < dim; ++i) {
        full[i + 1] = re.lattice[a * dim + i];
        s += full[i + 1];
      }
      full[0] = P - s;
      if (full[f] != 0) continue;
      // barycentrics w.r.t. the facet's vertices (all vertices but f, increasing);
      // drop the first one to get the (dim-1)-lattice coordinates
      int beta[3], nb = 0;
      bool first = true;
      for (int v = 0; v <= dim; ++v) {
        if (v == f) continue;
        if (first) {
          first = false;
          continue;
        }
        beta[nb++] = full[v];
      }
      nodes.push_back(a);
      fidx.push_back(dim == 1 ? 0 : lattice_index(dim - 1, P, flat, beta));
    }
    if ((int)nodes.size() != nf) throw std::runtime_error("facet node count mismatch");
    for (int b = 0; b < nf; ++b) re.fnode[(size_t)f * nf + b] = nodes[b];
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nf; ++b) {
        real s = 0;
        for (int k = 0; k < nf; ++k)
          s += mp.Minv[a * nd + nodes[k]] * fact * fm.M[fidx[k] * nf + fidx[b]];
        re.L[((size_t)f * nd + a) * nf + b] = (double)s;
      }
  }
  return re;
}

void tabulate(int dim, int P, int npts, const double* xi, double* phi) {
  int nd = num_nodes(dim, P);
  std::vector<int> lat;
  lattice_points(dim, P, lat);
  std::vector<real> C = lagrange_coeffs(dim, P, lat);
  std::vector<real> mono(nd);
  for (int p = 0; p < npts; ++p) {
    for (int m = 0; m < nd; ++m) {
      real v = 1;
      for (int i = 0; i < dim; ++i)
        for (int e = 0; e < lat[m * dim + i]; ++e) v *= (real)xi[p * dim + i];
      mono[m] = v;
    }
    for (int a = 0; a < nd; ++a) {
      real s = 0;
      for (int m = 0; m < nd; ++m) s += C[m * nd + a] * mono[m];
      phi[(size_t)p * nd + a] = (double)s;
    }
  }
}

std::vector<double> sponge_tensor(int dim, int P, int q) {
  int nd = num_nodes(dim, P), nq = num_nodes(dim, q);
  std::vector<int> latP, latQ;
  lattice_points(dim, P, latP);
  lattice_points(dim, q, latQ);
  std::vector<real> CP = lagrange_coeffs(dim, P, latP);
  std::vector<real> CQ = lagrange_coeffs(dim, q, latQ);
  MassPack mp = mass_matrix(dim, P);
  // I3[m][k][n] = int mono_m mono_k mono_n
  std::vector<real> T0((size_t)nd * nq * nd), T1((size_t)nd * nq * nd);
  int g[3];
  for (int m = 0; m < nd; ++m)
    for (int k = 0; k < nq; ++k)
      for (int n = 0; n < nd; ++n) {
        for (int i = 0; i < dim; ++i) g[i] = latP[m * dim + i] + latQ[k * dim + i] + latP[n * dim + i];
        T0[((size_t)m * nq + k) * nd + n] = mono_integral(dim, g);
      }
  // contract n -> b
  for (int m = 0; m < nd; ++m)
    for (int k = 0; k < nq; ++k)
      for (int b = 0; b < nd; ++b) {
        real s = 0;
        for (int n = 0; n < nd; ++n) s += T0[((size_t)m * nq + k) * nd + n] * CP[n * nd + b];
        T1[((size_t)m * nq + k) * nd + b] = s;
      }
  // contract k -> c
  for (int m = 0; m < nd; ++m)
    for (int c = 0; c < nq; ++c)
      for (int b = 0; b < nd; ++b) {
        real s = 0;
        for (int k = 0; k < nq; ++k) s += CQ[k * nq + c] * T1[((size_t)m * nq + k) * nd + b];
        T0[((size_t)m * nq + c) * nd + b] = s;
      }
  // contract m -> a'
  for (int a = 0; a < nd; ++a)
    for (int c = 0; c < nq; ++c)
      for (int b = 0; b < nd; ++b) {
        real s = 0;
        for (int m = 0; m < nd; ++m) s += CP[m * nd + a] * T0[((size_t)m * nq + c) * nd + b];
        T1[((size_t)a * nq + c) * nd + b] = s;
      }
  // apply Minv over a'
  std::vector<double> A((size_t)nd * nq * nd);
  for (int a = 0; a < nd; ++a)
    for (int c = 0; c < nq; ++c)
      for (int b = 0; b < nd; ++b) {
        real s = 0;
        for (int k = 0; k < nd; ++k) s += mp.Minv[a * nd + k] * T1[((size_t)k * nq + c) * nd + b];
        A[((size_t)a * nq + c) * nd + b] = (double)s;
      }
  return A;
}

}  // namespace sg
