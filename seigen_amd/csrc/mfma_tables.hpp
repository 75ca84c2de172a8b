// Host-side construction of the MFMA A-operand fragment tables for the 3-D
// high-order stage kernels (kernels_mfma.hip).
//
// v_mfma_f64_16x16x4_f64 takes A as one f64 per lane: lane l holds
// A[row = l & 15][k = l >> 4].  A table is a list of 16x4 tiles ("fragments"),
// each stored as 64 consecutive doubles in lane order, so a wave fetches a
// fragment with one conflict-free ds_read_b64.
//
// The node rows are covered by floor(nd / 16) such tiles; the remaining nd % 16 rows
// (3 of 35 at degree 4, 4 of 20 at degree 3) go through v_mfma_f64_4x4x4_4b_f64, which
// costs about 1/6 of a 16x16x4 issue slot (tools/ubench_mfma4.hip) instead of a full,
// mostly padded, 16-row tile.  Its four 4x4 blocks are the four 4-cell groups of the 16
// cells, so B is the same register as for the large shape (k = l >> 4, cell = l & 15), A is
// the 4x4 tile replicated per block: lane l holds A[row = l & 3][k = l >> 4], and lane l
// of the result holds row (l >> 4) of cell (l & 15).  "Small" fragments are stored in
// that lane order.
#pragma once
#include <vector>

#include "refelem.hpp"

namespace sg {

// 2-D tile kernels: elements with more nodes than this use one 16-row tile per k-step, the others 4-row tiles
// (kernels_tile2d.hip, tile2d_frags_*)
#ifndef SG_T2_LARGE_FROM
#define SG_T2_LARGE_FROM 8
#endif

struct MfmaGeom {
  int nd, nf;
  int ks;    // k-steps over the element nodes  = ceil(nd / 4)
  int ksf;   // k-steps over the facet nodes    = ceil(nf / 4)
  int mtf;   // full 16-row tiles over the nodes = floor(nd / 16)
  int nsm;   // 4-row tiles over the remaining rows = ceil((nd % 16) / 4)
  int mtt;   // row tiles of either kind = mtf + nsm (large ones first)
  int s4;    // row-quads over the nodes = ceil(nd / 4)
};

MfmaGeom mfma_geom(const RefElem& re);

// F volume: out = sum_r (-D_r) T~_r  as one product with K = 3 * (4*ks):
//   frag (t, kk), kk = ks*r + k0:  A[row][col] = -E_r[row0(t) + row][4 k0 + col]
// where row tile t < mtf is large (row0 = 16 t) and t >= mtf small (row0 = 16 mtf + 4 (t - mtf))
// (E_r = D_r minus the own-trace half of the central flux, see mfma_tables.cpp)
std::vector<double> mfma_frags_F(const RefElem& re);
// G volume: the row tiles of E_0, E_1, E_2 one after the other:
//   frag ((r*mtt + t)*ks + k0): A[row][col] = E_r[row0(t) + row][4 k0 + col]
std::vector<double> mfma_frags_G(const RefElem& re);
// `count` dense element-constant matrices M[r][nd][nd] as row tiles: frag ((r*mtt + t)*ks + k0): A[row][col] = M_r[row0(t) + row][4 k0 + col]
// (the X_k of cells with an affine sponge, kernels_mfma.hip sponge_affine_mfma)
std::vector<double> mfma_frags_dense(const RefElem& re, const double* M, int count);
// facet lifts (shared by F and G): frag ((f*mtt + t)*ksf + k0): A[row][col] = 1/2 L_f[row0(t) + row][4 k0 + col]
// (the 1/2 of the central flux average: both kernels lift +-1/2 of a neighbour trace)
std::vector<double> mfma_frags_L(const RefElem& re);

// Factorised G volume (kernels_mfma.hip mfma_stage_G<.., FACT = 1>).  The three D_r = Mhat^-1 Shat_r of degree p have rank
// dim P_{p-1} (they differentiate: 20 of 35 at degree 4, 10 of 20 at degree 3) and SHARE their row space, so
//     D_r = P_r Q,   Q [rk x nd] an orthonormal basis of that row space,   P_r = D_r Q^T [nd x rk]:
// y_i = Q u_i once per velocity component, then z_ri = P_r y_i per direction, instead of three dense D_r u_i:
// at degree 4 the volume phase takes 8640 matrix cycles per 16 cells instead of 11664.  (These are the UNFOLDED
// D_r: the own-trace half of the central flux, which mfma_frags_G folds into E_r and which has full rank, goes
// back to the lifts.)  The results of the first product come out of the MFMA in exactly the register layout the
// second one wants its B operand in (row 4 reg + q <-> k-step reg, lane group q).
//   Q: frag (t * ks + k0): A[row][col] = Q[row0q(t) + row][4 k0 + col], floor(rk / 16) large row tiles, then
//      ceil((rk % 16) / 4) small ones;   P: frag ((r * mtt + t) * kr + k0): A[row][col] = P_r[row0(t) + row][4 k0 + col],
//      kr = ceil(rk / 4) k-steps over the rank.
struct MfmaFactGeom {
  int rk;    // rank of the D_r
  int qlt;   // large row tiles of Q = floor(rk / 16)
  int qst;   // small row tiles of Q = ceil((rk % 16) / 4)
  int kr;    // k-steps over the rank = ceil(rk / 4)
};
MfmaFactGeom mfma_fact_geom(const RefElem& re);
// fills Q [rk][nd] and P [3][nd][rk] (row-major); throws if the numerical rank is not dim P_{p-1}
void mfma_factorise_D(const RefElem& re, std::vector<double>& Q, std::vector<double>& Pm);
std::vector<double> mfma_frags_Q(const RefElem& re);
std::vector<double> mfma_frags_P(const RefElem& re);

// The same three tables for the float kernels (v_mfma_f32_16x16x4_f32): every row tile is a
// 16-row tile (ceil(nd / 16) of them, the last one zero-padded), and lane l of tile t holds the
// operator row of node 16 t + 4 (l & 3) + ((l & 15) >> 2): the f32 shape returns row 4 q + reg in
// register reg of lane group q where the f64 shape returns row 4 reg + q, and with this permutation
// both leave node 16 t + 4 reg + q there.  Fragment order as above with mtt = ceil(nd / 16).
std::vector<float> mfma32_frags_F(const RefElem& re);
std::vector<float> mfma32_frags_G(const RefElem& re);
std::vector<float> mfma32_frags_L(const RefElem& re);

// ---- 2-D tile kernels (kernels_tile2d.hip) ----------------------------------------------------
// nd <= 15: one 16-row fragment per (operator, k-step) at degrees 3 and 4 (lane l: A[row = l & 15][k = l >> 4]),
// ceil(nd / 4) 4-row fragments at degrees 1 and 2 (lane l: A[row = 4 t + (l & 3)][k = l >> 4]).
//   V: frag ((r*ks + k0)*rt + t): sign * E_r[row][4 k0 + col]   (sign = -1 for F, +1 for G)
//   L: frag ((f*ksf + k0)*rt + t): 1/2 L_f[row][4 k0 + col]
// with E_r = D_r - 1/2 (L_0 R_0 - L_{r+1} R_{r+1}) (the fold of mfma_tables.cpp with the 2-D normal scale).
std::vector<double> tile2d_frags_V(const RefElem& re, double sign);
std::vector<double> tile2d_frags_L(const RefElem& re);
// float kernels (sg_config::dtype = 1): one zero-padded 16-row tile per (operator, k-step), rows permuted for the
// C/D layout of v_mfma_f32_16x16x4_f32 (mfma_tables.cpp)
std::vector<float> tile2d_frags32_V(const RefElem& re, double sign);
std::vector<float> tile2d_frags32_L(const RefElem& re);

}  // namespace sg
