// Reference simplex operators for equispaced Lagrange P_k (host side, setup only) - and, kind = 1, the same
// for the tensor-product element DQ_k on the unit square: (k+1)^2 equispaced nodes, first coordinate fastest;
// faces 0: x = 0, 1: x = 1, 2: y = 0, 3: y = 1 (what FunctionSpace(mesh, "DG", k), seigen/elastic.py:81-82,
// is on a quadrilateral mesh [upstream]).
//
// Replaces what `assemble(inner(w,u)*dx, inverse=True)` (seigen/elastic.py:376-382)
// and the TSFC-generated element kernels tabulate per cell [upstream]: on affine
// simplices with constant coefficients every integral of elastic.py:204-219 is
// a reference matrix times a geometric factor, so the element-wise inverse
// mass can be folded into the operators once:
//   D_r = Mhat^-1 Shat_r,   Shat_r[a][b] = int dphi_a/dxi_r phi_b      (volume)
//   L_f = Mhat^-1[:, face f nodes] * Mface_unit                        (facet lift)
// with Mface_unit the facet mass matrix normalised to unit facet measure.
#pragma once
#include <vector>

namespace sg {

struct RefElem {
  int dim = 0, P = 0, nd = 0, nf = 0, nfaces = 0, kind = 0;
  std::vector<int> lattice;   // [nd][dim]
  std::vector<int> fnode;     // [nfaces][nf] element-node index of each facet node
  std::vector<double> Mhat;   // [nd][nd]
  std::vector<double> Minv;   // [nd][nd]
  std::vector<double> D;      // [dim][nd][nd]
  std::vector<double> L;      // [nfaces][nd][nf]
  // monomial expansion phi_a = sum_m C[m][a] xi^gamma_m (long double kept as double pairs is
  // not needed: the sponge tensor is built inside refelem.cpp)
};

constexpr int KIND_SIMPLEX = 0, KIND_TENSOR = 1;
int num_nodes(int dim, int P, int kind = KIND_SIMPLEX);
void lattice_points(int dim, int P, std::vector<int>& out, int kind = KIND_SIMPLEX);  // [nd][dim], first coord fastest
RefElem make_refelem(int dim, int P, int kind = KIND_SIMPLEX);

// phi[p][a] = Lagrange basis a of P_k at reference point xi[p][:]  (degree up to 8)
void tabulate(int dim, int P, int npts, const double* xi, double* phi, int kind = KIND_SIMPLEX);

// A[a][c][b] = sum_a' Minv[a][a'] int phi_a' psi_c phi_b, psi in P_q.  Size nd*nq*nd.
// (absorption term -inner(w, sigma*u0)*dx, elastic.py:207-208, with sigma in DG_q)
std::vector<double> sponge_tensor(int dim, int P, int q, int kind = KIND_SIMPLEX);

}  // namespace sg
