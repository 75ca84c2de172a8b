// What sg_set_absorption derives from the nodal sigma of every cell - plain C++, no device, no HIP header: built by
// `make host-asan` too and walked by tools/host_asan_driver.cpp.
//
// The absorption term of the velocity equation, -inner(w, sigma u0) dx (seigen/elastic.py:207-208) with sigma in any DG_q
// space (:136-141), is per cell a matrix B_e = Mhat^-1 int sigma phi_a phi_b = sum_c sigma_c A[:, c, :] (A: refelem.hpp
// sponge_tensor).  A cell gets, in this order of preference,
//   nothing           sigma = 0 on all its nodes
//   a scalar          sigma is one value on all nodes: B_e = sigma I (families with StageArgs::sponge_sigma)
//   dim + 1 numbers   sigma is AFFINE in the reference coordinates, sigma = s_0 + sum_k s_k xi_k (every cell inside a linear
//                     ramp): B_e = s_0 I + sum_k s_k X_k with the element-constant X_k = Mhat^-1 int xi_k phi_a phi_b
//                     (families whose F stages read B_e u_abs from a pre-pass; kernels.hpp launch_sponge_pre_affine)
//   a matrix          anything else; cells with the same nodal sigma share one
#pragma once
#include <cstdint>
#include <vector>

#include "refelem.hpp"

namespace sg {

struct SpongeRequest {
  int dim = 0, degree = 0, kind = 0;   // the velocity element (refelem.hpp KIND_*)
  int sigma_degree = 0;
  int64_t ncells = 0;
  int ncls = 1, gw = 1;                // classes per cube, cells per item of the family's layout (mesh_tables.hpp)
  bool want_scalar = false;            // the family applies a cell-constant sigma without a matrix
  bool pre_family = false;             // the family reads B_e u_abs from a pre-pass: slots number CELLS, not matrices
  bool try_affine = false;             // ... and its affine cells take dim + 1 numbers
  bool line_layout = false;            // pre-pass results in the layout of the fields: slot = item' * gw + column
};

struct SpongePlan {
  std::vector<int32_t> slot;        // [cell] -> slot, or -1
  std::vector<double> sig;          // [cell] (want_scalar): 0 none, a value = constant sigma, NaN = the cell has a slot
  std::vector<double> B;            // [matrix][nd][nd]
  std::vector<int32_t> mat_of;      // [slot] -> matrix, -1 for an affine (or unused) slot       (pre_family)
  std::vector<int32_t> mat_slots;   // the slots that have a matrix, ascending                     (pre_family)
  std::vector<int32_t> cells;       // [slot] -> cell (0 for an unused slot of the line layout)   (pre_family)
  std::vector<double> aff_coef;     // [slot][dim + 1], zeros where the slot has a matrix          (naffine > 0)
  int32_t nslots = 0, nmat = 0, naffine = 0;
  // the element-constant operators of the affine cells (naffine > 0)
  std::vector<double> Xd;           // [dim][nd][nd] dense
  std::vector<double> X;            // [dim][nd][W] rows in ELL form over the union of the patterns
  std::vector<int32_t> col;         // [nd][W] (identity where dense)
  int W = 0;
  bool dense = true;
  std::vector<int32_t> items;       // items (cube group * ncls + class) that hold an affine cell
  std::vector<int32_t> item_slots;  // [item][gw] -> slot of the cell if it is affine, else -1
};

// sigma_nodes [ncells][nq] in the host numbering; throws std::invalid_argument on a bad request
SpongePlan plan_sponge(const SpongeRequest& rq, const double* sigma_nodes);

// is sigma_c = s_0 + sum_k s_k xi_k(c) on all nq nodes of the DG_q lattice (to 64 ulp of the largest value)?  s: dim + 1
bool sponge_affine_fit(const double* sigma, const std::vector<int>& latQ, int nq, int dim, int q, const int* vtx, double* s);

}  // namespace sg
