// Reference-element matrices by exact integration of monomials on the unit
// simplex, in long double.  Host-side setup code (runs once per sg_create).
#include "refelem.hpp"

#include <cmath>
#include <stdexcept>

namespace sg {

typedef long double real;

int num_nodes(int dim, int P, int kind) {
  if (dim == 0) return 1;
  if (dim == 1) return P + 1;
  if (kind == KIND_TENSOR) return dim == 2 ? (P + 1) * (P + 1) : (P + 1) * (P + 1) * (P + 1);
  if (dim == 2) return (P + 1) * (P + 2) / 2;
  return (P + 1) * (P + 2) * (P + 3) / 6;
}

void lattice_points(int dim, int P, std::vector<int>& out, int kind) {
  out.clear();
  if (dim == 0) return;
  if (dim == 1) {
    for (int a = 0; a <= P; ++a) out.push_back(a);
  } else if (kind == KIND_TENSOR) {
    for (int a3 = 0; a3 <= (dim == 3 ? P : 0); ++a3)
      for (int a2 = 0; a2 <= P; ++a2)
        for (int a1 = 0; a1 <= P; ++a1) {
          out.push_back(a1);
          out.push_back(a2);
          if (dim == 3) out.push_back(a3);
        }
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

// int over the unit dim-simplex of xi^g = prod g_i! / (|g| + dim)!;  over the unit square / cube: prod 1 / (g_i + 1)
static real mono_integral(int dim, const int* g, int kind = KIND_SIMPLEX) {
  int s = 0;
  real num = 1;
  if (kind == KIND_TENSOR && dim > 1) {
    for (int i = 0; i < dim; ++i) num /= (real)(g[i] + 1);
    return num;
  }
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
  int nd = (int)lat.size() / (dim > 0 ? dim : 1);
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

static MassPack mass_matrix(int dim, int P, int kind = KIND_SIMPLEX) {
  int nd = num_nodes(dim, P, kind);
  MassPack mp;
  mp.M.assign(nd * nd, 0);
  if (dim == 0) {
    mp.M[0] = 1;
    mp.Minv = mp.M;
    return mp;
  }
  std::vector<int> lat;
  lattice_points(dim, P, lat, kind);
  std::vector<real> C = lagrange_coeffs(dim, P, lat);
  // Mmono[m][n]
  std::vector<real> Mm(nd * nd);
  int g[3];
  for (int m = 0; m < nd; ++m)
    for (int n = 0; n < nd; ++n) {
      for (int i = 0; i < dim; ++i) g[i] = lat[m * dim + i] + lat[n * dim + i];
      Mm[m * nd + n] = mono_integral(dim, g, kind);
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

// The tensor-product element from the interval element (Kronecker products: node a = a1 + (P+1) (a2 + (P+1) a3)):
//   Mhat = M1 x M1 (x M1),  D_r = D1 along axis r, the identity along the others,
//   L_f[a][b] = M1^-1[a_axis][at] delta(transverse(a), b)   (face f: axis f / 2 at node at = 0 or P; facet node b counts
//   the transverse lattice coordinates, the lower axis fastest),
// so the conditioning is the interval's (a 25 x 25 Vandermonde matrix of tensor monomials would cost five digits).
// dim = 2: quadrilaterals (4 faces); dim = 3: hexahedra (6 faces: x = 0, x = 1, y = 0, y = 1, z = 0, z = 1).
static RefElem make_tensor_refelem(int dim, int P) {
  if (dim != 2 && dim != 3) throw std::runtime_error("tensor-product cells: quadrilaterals (2-D) and hexahedra (3-D)");
  const RefElem r1 = make_refelem(1, P, KIND_SIMPLEX);
  const int n1 = P + 1;
  int nd = 1, nf = 1;
  for (int i = 0; i < dim; ++i) nd *= n1;
  for (int i = 0; i < dim - 1; ++i) nf *= n1;
  RefElem re;
  re.dim = dim;
  re.P = P;
  re.kind = KIND_TENSOR;
  re.nd = nd;
  re.nf = nf;
  re.nfaces = 2 * dim;
  lattice_points(dim, P, re.lattice, KIND_TENSOR);
  re.Mhat.assign((size_t)nd * nd, 0.0);
  re.Minv.assign((size_t)nd * nd, 0.0);
  re.D.assign((size_t)dim * nd * nd, 0.0);
  auto coord = [&](int a, int axis) { return re.lattice[(size_t)a * dim + axis]; };
  for (int a = 0; a < nd; ++a)
    for (int b = 0; b < nd; ++b) {
      double m = 1.0, mi = 1.0;
      for (int i = 0; i < dim; ++i) {
        m *= r1.Mhat[coord(a, i) * n1 + coord(b, i)];
        mi *= r1.Minv[coord(a, i) * n1 + coord(b, i)];
      }
      re.Mhat[(size_t)a * nd + b] = m;
      re.Minv[(size_t)a * nd + b] = mi;
      for (int r = 0; r < dim; ++r) {
        bool same = true;
        for (int i = 0; i < dim; ++i)
          if (i != r && coord(a, i) != coord(b, i)) same = false;
        if (same) re.D[((size_t)r * nd + a) * nd + b] = r1.D[coord(a, r) * n1 + coord(b, r)];
      }
    }
  re.fnode.assign((size_t)re.nfaces * nf, -1);
  re.L.assign((size_t)re.nfaces * nd * nf, 0.0);
  // position of node a in the node list of a facet across `axis`: its transverse coordinates, the lower axis fastest
  auto across = [&](int a, int axis) {
    int idx = 0, mul = 1;
    for (int i = 0; i < dim; ++i)
      if (i != axis) {
        idx += coord(a, i) * mul;
        mul *= n1;
      }
    return idx;
  };
  for (int f = 0; f < re.nfaces; ++f) {
    const int axis = f / 2, at = (f % 2) ? P : 0;
    for (int a = 0; a < nd; ++a) {
      if (coord(a, axis) == at) re.fnode[(size_t)f * nf + across(a, axis)] = a;
      re.L[((size_t)f * nd + a) * nf + across(a, axis)] = r1.Minv[coord(a, axis) * n1 + at];
    }
  }
  return re;
}

RefElem make_refelem(int dim, int P, int kind) {
  if (dim < 1 || dim > 3 || P < 1 || P > 4) throw std::runtime_error("dim must be 1..3 and degree 1..4");
  if (kind != KIND_SIMPLEX && !(kind == KIND_TENSOR && dim >= 2))
    throw std::runtime_error("tensor-product cells: quadrilaterals (2-D) and hexahedra (3-D)");
  if (kind == KIND_TENSOR) return make_tensor_refelem(dim, P);
  RefElem re;
  re.dim = dim;
  re.P = P;
  re.kind = kind;
  re.nd = num_nodes(dim, P, kind);
  re.nf = num_nodes(dim - 1, P, kind);
  re.nfaces = kind == KIND_TENSOR ? 2 * dim : dim + 1;
  lattice_points(dim, P, re.lattice, kind);
  const int nd = re.nd, nf = re.nf;
  std::vector<real> C = lagrange_coeffs(dim, P, re.lattice);
  MassPack mp = mass_matrix(dim, P, kind);

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
        Sm[m * nd + n] = (real)gr * mono_integral(dim, g, kind);
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
  MassPack fm = mass_matrix(dim - 1, P, kind);
  std::vector<int> flat;
  lattice_points(dim - 1, P, flat, kind);
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

void tabulate(int dim, int P, int npts, const double* xi, double* phi, int kind) {
  if (kind == KIND_TENSOR && dim >= 2) {
    const int n1 = P + 1;
    int nd = 1;
    for (int i = 0; i < dim; ++i) nd *= n1;
    std::vector<double> x(npts);
    std::vector<std::vector<double>> p1((size_t)dim, std::vector<double>((size_t)npts * n1));
    for (int i = 0; i < dim; ++i) {
      for (int p = 0; p < npts; ++p) x[p] = xi[(size_t)dim * p + i];
      tabulate(1, P, npts, x.data(), p1[(size_t)i].data(), KIND_SIMPLEX);
    }
    for (int p = 0; p < npts; ++p)
      for (int a = 0; a < nd; ++a) {
        double v = 1.0;
        int rest = a;
        for (int i = 0; i < dim; ++i) {
          v *= p1[(size_t)i][(size_t)p * n1 + rest % n1];
          rest /= n1;
        }
        phi[(size_t)p * nd + a] = v;
      }
    return;
  }
  int nd = num_nodes(dim, P, kind);
  std::vector<int> lat;
  lattice_points(dim, P, lat, kind);
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

std::vector<double> sponge_tensor(int dim, int P, int q, int kind) {
  if (kind == KIND_TENSOR && dim >= 2) {   // A[a][c][b] = prod over the axes of A1[a_i][c_i][b_i]
    const std::vector<double> A1 = sponge_tensor(1, P, q, KIND_SIMPLEX);
    const int n1 = P + 1, q1 = q + 1;
    int ndt = 1, nqt = 1;
    for (int i = 0; i < dim; ++i) {
      ndt *= n1;
      nqt *= q1;
    }
    std::vector<double> A((size_t)ndt * nqt * ndt);
    for (int a = 0; a < ndt; ++a)
      for (int c = 0; c < nqt; ++c)
        for (int b = 0; b < ndt; ++b) {
          double v = 1.0;
          int ra = a, rc = c, rb = b;
          for (int i = 0; i < dim; ++i) {
            v *= A1[((size_t)(ra % n1) * q1 + rc % q1) * n1 + rb % n1];
            ra /= n1;
            rc /= q1;
            rb /= n1;
          }
          A[((size_t)a * nqt + c) * ndt + b] = v;
        }
    return A;
  }
  int nd = num_nodes(dim, P, kind), nq = num_nodes(dim, q, kind);
  std::vector<int> latP, latQ;
  lattice_points(dim, P, latP, kind);
  lattice_points(dim, q, latQ, kind);
  std::vector<real> CP = lagrange_coeffs(dim, P, latP);
  std::vector<real> CQ = lagrange_coeffs(dim, q, latQ);
  MassPack mp = mass_matrix(dim, P, kind);
  // I3[m][k][n] = int mono_m mono_k mono_n
  std::vector<real> T0((size_t)nd * nq * nd), T1((size_t)nd * nq * nd);
  int g[3];
  for (int m = 0; m < nd; ++m)
    for (int k = 0; k < nq; ++k)
      for (int n = 0; n < nd; ++n) {
        for (int i = 0; i < dim; ++i) g[i] = latP[m * dim + i] + latQ[k * dim + i] + latP[n * dim + i];
        T0[((size_t)m * nq + k) * nd + n] = mono_integral(dim, g, kind);
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
