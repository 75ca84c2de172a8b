#include "mfma_tables.hpp"

#include <cmath>
#include <cstring>
#include <stdexcept>

#include "kernels.hpp"

#include <cstddef>
using std::size_t;

namespace sg {

MfmaGeom mfma_geom(const RefElem& re) {
  MfmaGeom g;
  g.nd = re.nd;
  g.nf = re.nf;
  g.ks = (re.nd + 3) / 4;
  g.ksf = (re.nf + 3) / 4;
  g.mtf = re.nd / 16;
  g.nsm = (re.nd % 16 + 3) / 4;
  g.mtt = g.mtf + g.nsm;
  g.s4 = (re.nd + 3) / 4;
  return g;
}

// E_r = D_r - C_r with the own-trace half of the central flux folded in:
//   C_r = 1/4 (L_0 R_0 - L_{r+1} R_{r+1}),  R_f = restriction to the nodes of facet f.
// On an affine tetrahedron (c n)_f = -grad(lambda_f)/2 and grad(lambda_f) is row f-1 of
// Jinv (minus the sum of the rows for f = 0), so the own-side part of
//   sum_f L_f [ (c n)_f . 1/2 T+ ]   is   sum_r C_r (Jinv_r . T),
// i.e. it has the shape of the volume term.  The facet lifts then only carry the
// NEIGHBOUR half (and a correction on domain-boundary facets, kernels_mfma.hip).
// In d dimensions (c n)_f = -grad(lambda_f)/(d-1)!, so the factor is 1/(2 (d-1)!): 1/4 on tetrahedra, 1/2 on triangles.
// Quadrilaterals (tensor-product element): (c n) of the facets 2r / 2r + 1 is -/+ row r of Jinv, so
//   C_r = 1/2 (L_{2r+1} R_{2r+1} - L_{2r} R_{2r}).
static inline double Eval(const RefElem& re, int r, int a, int b) {
  if (a >= re.nd || b >= re.nd) return 0.0;
  if (re.kind == KIND_TENSOR) {
    double v = re.D[((size_t)r * re.nd + a) * re.nd + b];
    for (int bf = 0; bf < re.nf; ++bf) {
      if (re.fnode[(size_t)(2 * r + 1) * re.nf + bf] == b) v -= 0.5 * re.L[((size_t)(2 * r + 1) * re.nd + a) * re.nf + bf];
      if (re.fnode[(size_t)(2 * r) * re.nf + bf] == b) v += 0.5 * re.L[((size_t)(2 * r) * re.nd + a) * re.nf + bf];
    }
    return v;
  }
  const double cfold = (re.dim == 3) ? 0.25 : 0.5;
  double v = re.D[((size_t)r * re.nd + a) * re.nd + b];
  for (int bf = 0; bf < re.nf; ++bf) {
    if (re.fnode[(size_t)0 * re.nf + bf] == b) v -= cfold * re.L[((size_t)0 * re.nd + a) * re.nf + bf];
    if (re.fnode[(size_t)(r + 1) * re.nf + bf] == b) v += cfold * re.L[((size_t)(r + 1) * re.nd + a) * re.nf + bf];
  }
  return v;
}

// node row that lane l of row tile t holds in the A operand
static inline int tile_row(const MfmaGeom& g, int t, int l) {
  return (t < g.mtf) ? 16 * t + (l & 15) : 16 * g.mtf + 4 * (t - g.mtf) + (l & 3);
}

std::vector<double> mfma_frags_F(const RefElem& re) {
  MfmaGeom g = mfma_geom(re);
  std::vector<double> out((size_t)g.mtt * 3 * g.ks * 64, 0.0);
  for (int t = 0; t < g.mtt; ++t)
    for (int r = 0; r < 3; ++r)
      for (int k0 = 0; k0 < g.ks; ++k0) {
        size_t frag = (size_t)t * 3 * g.ks + (size_t)g.ks * r + k0;
        for (int l = 0; l < 64; ++l) out[frag * 64 + l] = -Eval(re, r, tile_row(g, t, l), 4 * k0 + (l >> 4));
      }
  return out;
}

std::vector<double> mfma_frags_G(const RefElem& re) {
  MfmaGeom g = mfma_geom(re);
  std::vector<double> out((size_t)3 * g.mtt * g.ks * 64, 0.0);
  for (int r = 0; r < 3; ++r)
    for (int t = 0; t < g.mtt; ++t)
      for (int k0 = 0; k0 < g.ks; ++k0) {
        size_t frag = ((size_t)r * g.mtt + t) * g.ks + k0;
        for (int l = 0; l < 64; ++l) out[frag * 64 + l] = Eval(re, r, tile_row(g, t, l), 4 * k0 + (l >> 4));
      }
  return out;
}

// row tiles of dense element-constant nd x nd matrices M_0 .. M_{count-1} (row-major), in the order of mfma_frags_G
std::vector<double> mfma_frags_dense(const RefElem& re, const double* M, int count) {
  MfmaGeom g = mfma_geom(re);
  std::vector<double> out((size_t)count * g.mtt * g.ks * 64, 0.0);
  for (int r = 0; r < count; ++r)
    for (int t = 0; t < g.mtt; ++t)
      for (int k0 = 0; k0 < g.ks; ++k0) {
        size_t frag = ((size_t)r * g.mtt + t) * g.ks + k0;
        for (int l = 0; l < 64; ++l) {
          const int a = tile_row(g, t, l), b = 4 * k0 + (l >> 4);
          out[frag * 64 + l] = (a < re.nd && b < re.nd) ? M[((size_t)r * re.nd + a) * re.nd + b] : 0.0;
        }
      }
  return out;
}

std::vector<double> mfma_frags_L(const RefElem& re) {
  MfmaGeom g = mfma_geom(re);
  std::vector<double> out((size_t)re.nfaces * g.mtt * g.ksf * 64, 0.0);
  for (int f = 0; f < re.nfaces; ++f)
    for (int t = 0; t < g.mtt; ++t)
      for (int k0 = 0; k0 < g.ksf; ++k0) {
        size_t frag = ((size_t)f * g.mtt + t) * g.ksf + k0;
        for (int l = 0; l < 64; ++l) {
          int a = tile_row(g, t, l), b = 4 * k0 + (l >> 4);
          out[frag * 64 + l] = (a < re.nd && b < re.nf) ? 0.5 * re.L[((size_t)f * re.nd + a) * re.nf + b] : 0.0;
        }
      }
  return out;
}

// ---- factorised G volume ------------------------------------------------------------------------
MfmaFactGeom mfma_fact_geom(const RefElem& re) {
  MfmaFactGeom f;
  f.rk = re.P >= 2 ? num_nodes(re.dim, re.P - 1) : 1;     // dim P_{p-1}
  f.qlt = f.rk / 16;
  f.qst = (f.rk % 16 + 3) / 4;
  f.kr = (f.rk + 3) / 4;
  return f;
}

// Orthonormal basis of the common row space of D_0, D_1, D_2 by modified Gram-Schmidt with pivoting over their
// 3 nd rows (long double; two orthogonalisation sweeps per vector), then P_r = D_r Q^T.
void mfma_factorise_D(const RefElem& re, std::vector<double>& Q, std::vector<double>& Pm) {
  const int nd = re.nd, d = re.dim, rk = mfma_fact_geom(re).rk;
  std::vector<long double> rows((size_t)d * nd * nd);
  for (size_t i = 0; i < rows.size(); ++i) rows[i] = re.D[i];
  std::vector<long double> q((size_t)rk * nd, 0.0L);
  long double first = 0.0L;
  for (int k = 0; k < rk; ++k) {
    int best = -1;
    long double bn = -1.0L;
    for (int r = 0; r < d * nd; ++r) {
      long double n2 = 0.0L;
      for (int j = 0; j < nd; ++j) n2 += rows[(size_t)r * nd + j] * rows[(size_t)r * nd + j];
      if (n2 > bn) {
        bn = n2;
        best = r;
      }
    }
    if (k == 0) first = bn;
    if (!(bn > 1e-20L * first)) throw std::runtime_error("D_r: rank below dim P_{p-1}");
    long double* v = &q[(size_t)k * nd];
    for (int j = 0; j < nd; ++j) v[j] = rows[(size_t)best * nd + j];
    for (int sweep = 0; sweep < 2; ++sweep)
      for (int m = 0; m < k; ++m) {
        long double dot = 0.0L;
        for (int j = 0; j < nd; ++j) dot += v[j] * q[(size_t)m * nd + j];
        for (int j = 0; j < nd; ++j) v[j] -= dot * q[(size_t)m * nd + j];
      }
    long double nrm = 0.0L;
    for (int j = 0; j < nd; ++j) nrm += v[j] * v[j];
    nrm = std::sqrt(nrm);
    for (int j = 0; j < nd; ++j) v[j] /= nrm;
    for (int r = 0; r < d * nd; ++r) {       // deflate every row
      long double dot = 0.0L;
      for (int j = 0; j < nd; ++j) dot += rows[(size_t)r * nd + j] * v[j];
      for (int j = 0; j < nd; ++j) rows[(size_t)r * nd + j] -= dot * v[j];
    }
  }
  long double left = 0.0L;
  for (long double x : rows) left = std::max(left, std::fabs(x));
  if (left > 1e-11L * std::sqrt(first)) throw std::runtime_error("D_r: rank above dim P_{p-1}");
  Q.assign((size_t)rk * nd, 0.0);
  for (size_t i = 0; i < Q.size(); ++i) Q[i] = (double)q[i];
  Pm.assign((size_t)d * nd * rk, 0.0);
  for (int r = 0; r < d; ++r)
    for (int a = 0; a < nd; ++a)
      for (int k = 0; k < rk; ++k) {
        long double s = 0.0L;
        for (int j = 0; j < nd; ++j) s += (long double)re.D[((size_t)r * nd + a) * nd + j] * q[(size_t)k * nd + j];
        Pm[((size_t)r * nd + a) * rk + k] = (double)s;
      }
}

std::vector<double> mfma_frags_Q(const RefElem& re) {
  const MfmaGeom g = mfma_geom(re);
  const MfmaFactGeom f = mfma_fact_geom(re);
  std::vector<double> Q, Pm;
  mfma_factorise_D(re, Q, Pm);
  std::vector<double> out((size_t)(f.qlt + f.qst) * g.ks * 64, 0.0);
  for (int t = 0; t < f.qlt + f.qst; ++t)
    for (int k0 = 0; k0 < g.ks; ++k0)
      for (int l = 0; l < 64; ++l) {
        const int row = t < f.qlt ? 16 * t + (l & 15) : 16 * f.qlt + 4 * (t - f.qlt) + (l & 3), col = 4 * k0 + (l >> 4);
        out[((size_t)t * g.ks + k0) * 64 + l] = (row < f.rk && col < re.nd) ? Q[(size_t)row * re.nd + col] : 0.0;
      }
  return out;
}

std::vector<double> mfma_frags_P(const RefElem& re) {
  const MfmaGeom g = mfma_geom(re);
  const MfmaFactGeom f = mfma_fact_geom(re);
  std::vector<double> Q, Pm;
  mfma_factorise_D(re, Q, Pm);
  std::vector<double> out((size_t)3 * g.mtt * f.kr * 64, 0.0);
  for (int r = 0; r < 3; ++r)
    for (int t = 0; t < g.mtt; ++t)
      for (int k0 = 0; k0 < f.kr; ++k0)
        for (int l = 0; l < 64; ++l) {
          const int row = tile_row(g, t, l), col = 4 * k0 + (l >> 4);
          out[(((size_t)r * g.mtt + t) * f.kr + k0) * 64 + l] = (row < re.nd && col < f.rk) ? Pm[((size_t)r * re.nd + row) * f.rk + col] : 0.0;
        }
  return out;
}

// ---- float tables --------------------------------------------------------------------------------
static inline int tile_row32(int t, int l) { return 16 * t + 4 * (l & 3) + ((l & 15) >> 2); }

std::vector<float> mfma32_frags_F(const RefElem& re) {
  const int ks = (re.nd + 3) / 4, mtt = (re.nd + 15) / 16;
  std::vector<float> out((size_t)mtt * 3 * ks * 64, 0.0f);
  for (int t = 0; t < mtt; ++t)
    for (int r = 0; r < 3; ++r)
      for (int k0 = 0; k0 < ks; ++k0) {
        size_t frag = (size_t)t * 3 * ks + (size_t)ks * r + k0;
        for (int l = 0; l < 64; ++l) out[frag * 64 + l] = (float)-Eval(re, r, tile_row32(t, l), 4 * k0 + (l >> 4));
      }
  return out;
}

std::vector<float> mfma32_frags_G(const RefElem& re) {
  const int ks = (re.nd + 3) / 4, mtt = (re.nd + 15) / 16;
  std::vector<float> out((size_t)3 * mtt * ks * 64, 0.0f);
  for (int r = 0; r < 3; ++r)
    for (int t = 0; t < mtt; ++t)
      for (int k0 = 0; k0 < ks; ++k0) {
        size_t frag = ((size_t)r * mtt + t) * ks + k0;
        for (int l = 0; l < 64; ++l) out[frag * 64 + l] = (float)Eval(re, r, tile_row32(t, l), 4 * k0 + (l >> 4));
      }
  return out;
}

std::vector<float> mfma32_frags_L(const RefElem& re) {
  const int ksf = (re.nf + 3) / 4, mtt = (re.nd + 15) / 16;
  std::vector<float> out((size_t)re.nfaces * mtt * ksf * 64, 0.0f);
  for (int f = 0; f < re.nfaces; ++f)
    for (int t = 0; t < mtt; ++t)
      for (int k0 = 0; k0 < ksf; ++k0) {
        size_t frag = ((size_t)f * mtt + t) * ksf + k0;
        for (int l = 0; l < 64; ++l) {
          int a = tile_row32(t, l), b = 4 * k0 + (l >> 4);
          out[frag * 64 + l] = (a < re.nd && b < re.nf) ? (float)(0.5 * re.L[((size_t)f * re.nd + a) * re.nf + b]) : 0.0f;
        }
      }
  return out;
}

// ---- 2-D tile kernels ---------------------------------------------------------------------------
static inline int tile2d_row(bool large, int t, int l) { return large ? (l & 15) : 4 * t + (l & 3); }

// Row tiles: elements with more than 16 nodes (DQ_4: 25) use ceil(nd / 16) 16-row tiles; the tables hold tile 0's
// fragments of every (operator, k-step), then tile 1's (the kernels work the tiles off one after the other).
static inline int tile2d_ntiles(const RefElem& re, bool large) { return large ? (re.nd + 15) / 16 : 1; }

std::vector<double> tile2d_frags_V(const RefElem& re, double sign) {
  const int ks = (re.nd + 3) / 4, s4 = (re.nd + 3) / 4;
  const bool large = re.nd > SG_T2_LARGE_FROM;
  const int rt = large ? 1 : s4, mt = tile2d_ntiles(re, large);
  std::vector<double> out((size_t)mt * 2 * ks * rt * 64, 0.0);
  for (int tile = 0; tile < mt; ++tile)
    for (int r = 0; r < 2; ++r)
      for (int k0 = 0; k0 < ks; ++k0)
        for (int t = 0; t < rt; ++t) {
          size_t frag = (((size_t)tile * 2 + r) * ks + k0) * rt + t;
          for (int l = 0; l < 64; ++l)
            out[frag * 64 + l] = sign * Eval(re, r, 16 * tile + tile2d_row(large, t, l), 4 * k0 + (l >> 4));
        }
  return out;
}

std::vector<double> tile2d_frags_L(const RefElem& re) {
  const int ksf = (re.nf + 3) / 4, s4 = (re.nd + 3) / 4;
  const bool large = re.nd > SG_T2_LARGE_FROM;
  const int rt = large ? 1 : s4, mt = tile2d_ntiles(re, large);
  std::vector<double> out((size_t)mt * re.nfaces * ksf * rt * 64, 0.0);
  for (int tile = 0; tile < mt; ++tile)
    for (int f = 0; f < re.nfaces; ++f)
      for (int k0 = 0; k0 < ksf; ++k0)
        for (int t = 0; t < rt; ++t) {
          size_t frag = (((size_t)tile * re.nfaces + f) * ksf + k0) * rt + t;
          for (int l = 0; l < 64; ++l) {
            int a = 16 * tile + tile2d_row(large, t, l), b = 4 * k0 + (l >> 4);
            out[frag * 64 + l] = (a < re.nd && b < re.nf) ? 0.5 * re.L[((size_t)f * re.nd + a) * re.nf + b] : 0.0;
          }
        }
  return out;
}

// The same two tables for the float tile kernels: always zero-padded 16-row tiles (ceil(nd / 16) of them), and MFMA
// row i holds node 4 (i & 3) + (i >> 2) of its tile - v_mfma_f32_16x16x4_f32 returns row 4 (lane >> 4) + reg, and the
// kernels keep "accumulator register m of lane group q = node row 4 m + q" in both precisions.
static inline int tile2d_row32(int l) { return 4 * (l & 3) + ((l & 15) >> 2); }

std::vector<float> tile2d_frags32_V(const RefElem& re, double sign) {
  const int ks = (re.nd + 3) / 4, mt = (re.nd + 15) / 16;
  std::vector<float> out((size_t)mt * 2 * ks * 64, 0.0f);
  for (int tile = 0; tile < mt; ++tile)
    for (int r = 0; r < 2; ++r)
      for (int k0 = 0; k0 < ks; ++k0)
        for (int l = 0; l < 64; ++l)
          out[(((size_t)tile * 2 + r) * ks + k0) * 64 + l] = (float)(sign * Eval(re, r, 16 * tile + tile2d_row32(l), 4 * k0 + (l >> 4)));
  return out;
}

std::vector<float> tile2d_frags32_L(const RefElem& re) {
  const int ksf = (re.nf + 3) / 4, mt = (re.nd + 15) / 16;
  std::vector<float> out((size_t)mt * re.nfaces * ksf * 64, 0.0f);
  for (int tile = 0; tile < mt; ++tile)
    for (int f = 0; f < re.nfaces; ++f)
      for (int k0 = 0; k0 < ksf; ++k0)
        for (int l = 0; l < 64; ++l) {
          const int a = 16 * tile + tile2d_row32(l), b = 4 * k0 + (l >> 4);
          out[(((size_t)tile * re.nfaces + f) * ksf + k0) * 64 + l] =
              (a < re.nd && b < re.nf) ? (float)(0.5 * re.L[((size_t)f * re.nd + a) * re.nf + b]) : 0.0f;
        }
  return out;
}

MfmaConst mfma_const(const MeshDev& md) {
  MfmaConst c;
  std::memset(&c, 0, sizeof(c));
  for (int a = 0; a < 3; ++a) c.n[a] = md.n[a];
  c.halo_per_cube = md.halo_per_cube;
  for (int s = 0; s < 6; ++s) c.has_nbr[s] = md.has_nbr[s];
  c.ncube = md.ncube;
  c.ncube_pad = md.ncube_pad;
  const int nf = md.nf;
  auto pack = [&](const uint8_t* row, int ks) {
    uint32_t w = 0;
    for (int q = 0; q < 4; ++q) {
      const int bb = (4 * ks + q < nf) ? 4 * ks + q : 0;   // padded rows repeat facet node 0 (they meet zero lift columns)
      w |= (uint32_t)row[bb] << (8 * q);
    }
    return w;
  };
  for (int f = 0; f < 4; ++f)
    for (int ks = 0; ks < MK_KSF; ++ks) c.fw[f][ks] = pack(md.fnode[f], ks);
  for (int k = 0; k < 6; ++k) {
    MfmaClassConst& kc = c.cls[k];
    for (int f = 0; f < 4; ++f) {
      kc.nb_axis[f] = md.nb_axis[k][f];
      kc.nb_dir[f] = md.nb_dir[k][f];
      kc.nb_cls[f] = md.nb_cls[k][f];
      kc.nb_face[f] = md.nb_face[k][f];
      const int kn = md.nb_cls[k][f], fn = md.nb_face[k][f];
      kc.slot_ord[f] = (kn >= 0 && kn < MAX_CLS && fn >= 0 && fn < MAX_FACES) ? md.face_ord[kn][fn] : 0;
      for (int ks = 0; ks < MK_KSF; ++ks) {
        kc.nbw[f][ks] = pack(md.nb_node[k][f], ks);
        kc.nfw[f][ks] = pack(md.nb_fnode[k][f], ks);
      }
    }
  }
  return c;
}


// Hexahedra DQ_3 / DQ_4 (kernels_hexm.hip): the line operators of the sum-factorised element on a block's cubes.
//   E[k][a'][a] = -D1[a'][a] / h_k + 1/2 (c n)_{2k} lift1[0][a'] [a == 0] + 1/2 (c n)_{2k+1} lift1[1][a'] [a == P]
//                 (the own-trace half of the central flux folded onto the line's end nodes),
//   lw[k][s][a'] = 1/2 (c n)_{2k+s} lift1[s][a']   (what the neighbour's value across facet 2k + s is lifted with),
// then the A operands of the x pass in lane order (v_mfma_f64_4x4x4_4b: lane l holds row l & 3, column l >> 4 of a
// 4x4 block, the same for its four blocks): AX[k'][k][l] = E[0][4k' + (l & 3)][4k + (l >> 4)] and, for the extra
// k-step that carries the two x-facet traces (lane group 0: facet 0, group 1: facet 1), AT[k'][l] = lw[0][l >> 4][4k' + (l & 3)].
std::vector<double> hexm_table(int P, const double* D1, const double* lift1, const MeshDev& md) {
  const int n1 = P + 1, ksx = (n1 + 3) / 4;
  const int off_lw = 3 * n1 * n1, off_ax = off_lw + 6 * n1, off_at = off_ax + ksx * ksx * 64;
  std::vector<double> t((size_t)off_at + (size_t)ksx * 64, 0.0);
  auto E = [&](int k, int ap, int a) -> double& { return t[(size_t)(k * n1 + ap) * n1 + a]; };
  auto LW = [&](int k, int s2, int ap) -> double& { return t[(size_t)off_lw + (size_t)(k * 2 + s2) * n1 + ap]; };
  for (int k = 0; k < 3; ++k) {
    const double ih = md.Jinv[0][k][k];
    const double c0 = md.cn[0][2 * k][k], c1 = md.cn[0][2 * k + 1][k];
    for (int ap = 0; ap < n1; ++ap) {
      LW(k, 0, ap) = 0.5 * c0 * lift1[ap];
      LW(k, 1, ap) = 0.5 * c1 * lift1[n1 + ap];
      for (int a = 0; a < n1; ++a) E(k, ap, a) = -ih * D1[ap * n1 + a];
      E(k, ap, 0) += LW(k, 0, ap);
      E(k, ap, P) += LW(k, 1, ap);
    }
  }
  for (int kp = 0; kp < ksx; ++kp)
    for (int l = 0; l < 64; ++l) {
      const int row = 4 * kp + (l & 3);
      for (int ks = 0; ks < ksx; ++ks) {
        const int col = 4 * ks + (l >> 4);
        t[(size_t)off_ax + (size_t)(kp * ksx + ks) * 64 + l] = (row < n1 && col < n1) ? E(0, row, col) : 0.0;
      }
      t[(size_t)off_at + (size_t)kp * 64 + l] = (row < n1 && (l >> 4) < 2) ? LW(0, l >> 4, row) : 0.0;
    }
  return t;
}

// 2-D tile kernels (kernels_tile2d.hip): kernarg copy of what they need from the mesh tables
T2Const tile2d_const(const MeshDev& md) {
  T2Const C;
  C.n0 = md.n[0];
  C.n1 = md.n[1];
  C.ncube = (int32_t)md.ncube;
  C.ngroups = (int32_t)(md.ncube_pad / 16);
  C.halo_per_cube = md.halo_per_cube;
  C.gpr = (md.n[0] + 15) / 16 + 1;
  C.inv_n0 = 1.0 / (double)md.n[0];
  for (int s = 0; s < 4; ++s) C.has_nbr[s] = md.has_nbr[s];
  const int ksf = (md.nf + 3) / 4;
  auto pack = [&](auto entry, int ks) {
    uint32_t wd = 0;
    for (int qq = 0; qq < 4; ++qq) {
      const int bb = (4 * ks + qq < md.nf) ? 4 * ks + qq : 0;  // padded rows meet zero lift columns
      wd |= (uint32_t)entry(bb) << (8 * qq);
    }
    return wd;
  };
  std::memset(C.tpw, 0, sizeof(C.tpw));
  std::memset(C.cls, 0, sizeof(C.cls));
  for (int f = 0; f < md.nfaces; ++f) {
    for (int ks = 0; ks < 2; ++ks) C.tpw[f][ks] = ks < ksf ? pack([&](int bb) { return md.fnode[f][bb]; }, ks) : 0u;
    for (int k = 0; k < md.ncls; ++k) {
      T2Class& K = C.cls[k];
      K.nb_axis[f] = md.nb_axis[k][f];
      K.nb_dir[f] = md.nb_dir[k][f];
      K.nb_cls[f] = md.nb_cls[k][f];
      const int ord = md.face_ord[md.nb_cls[k][f]][md.nb_face[k][f]];
      K.slot_ord[f] = ord < 0 ? 0 : ord;
      for (int ks = 0; ks < 2; ++ks) {
        K.tfw[f][ks] = ks < ksf ? pack([&](int bb) { return md.nb_node[k][f][bb]; }, ks) : 0u;
        K.tgw[f][ks] = ks < ksf ? pack([&](int bb) { return md.nb_fnode[k][f][bb]; }, ks) : 0u;
      }
      for (int j = 0; j < 2; ++j) K.cn[f][j] = md.cn[k][f][j];
    }
  }
  for (int k = 0; k < md.ncls; ++k)
    for (int r = 0; r < 2; ++r)
      for (int j = 0; j < 2; ++j) C.cls[k].Jinv[r][j] = md.Jinv[k][r][j];
  return C;
}

void mfma_trace_offsets(const MeshDev& md, int ncomp, std::vector<int32_t>& tab) {
  tab.assign((size_t)3 * 6 * 4 * 4 * 4, 0);
  const int nf = md.nf;
  for (int var = 0; var < 3; ++var)
    for (int k = 0; k < 6; ++k)
      for (int f = 0; f < 4; ++f)
        for (int q = 0; q < 4; ++q)
          for (int ks = 0; ks < 4; ++ks) {
            const int bb = (4 * ks + q < nf) ? 4 * ks + q : 0;      // padded rows repeat facet node 0 (they meet zero lift columns)
            int v;
            if (var == 0) v = (int)md.nb_node[k][f][bb] * ncomp * 16;
            else if (var == 1) v = (int)md.nb_fnode[k][f][bb] * 3;
            else v = (int)md.fnode[f][bb] * ncomp * 16;
            tab[(size_t)((((var * 6 + k) * 4 + f) * 4 + q) * 4 + ks)] = v;
          }
}

// The neighbour of (cube c, class k) across facet f, exactly as the stage kernels used to derive it per item
// (kernels_mfma.hip nbr_ref): same cube (axis < 0), the cube one step along `axis`, the domain boundary, or a
// neighbour block's packed trace.  A cell slot is ((c / 16) * 6 + class) * 16 + c % 16: the position of the cell's
// 16-lane column in the interleaved layout, independent of node and component counts.
void build_nbr_table(const MeshDev& md, std::vector<int32_t>& tab) {
  const int64_t ngroups = md.ncube_pad / 16;
  tab.assign((size_t)ngroups * 6 * 64, -1);
  const int64_t n0 = md.n[0], n1 = md.n[1], n2 = md.n[2];
  for (int64_t g = 0; g < ngroups; ++g)
    for (int k = 0; k < 6; ++k)
      for (int w = 0; w < 16; ++w) {
        const int64_t c = g * 16 + w;
        int32_t* e = &tab[(size_t)(((g * 6 + k) * 16 + w) * 4)];
        if (c >= md.ncube) continue;                      // layout padding: treated as domain boundary (reads its own slot)
        const int64_t cc[3] = {c % n0, (c / n0) % n1, c / (n0 * n1)};
        for (int f = 0; f < 4; ++f) {
          const int axis = md.nb_axis[k][f], kn = md.nb_cls[k][f];
          if (axis < 0) {
            e[f] = (int32_t)((g * 6 + kn) * 16 + w);
            continue;
          }
          const int dir = md.nb_dir[k][f];
          const int64_t cn = cc[axis] + dir, nax = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
          if (cn >= 0 && cn < nax) {
            const int64_t nc = c + dir * (axis == 0 ? 1 : (axis == 1 ? n0 : n0 * n1));
            e[f] = (int32_t)(((nc >> 4) * 6 + kn) * 16 + (nc & 15));
            continue;
          }
          const int side = 2 * axis + (dir > 0 ? 1 : 0);
          if (md.has_nbr[side]) {
            const int64_t c2 = axis == 0 ? cc[1] + n1 * cc[2] : (axis == 1 ? cc[0] + n0 * cc[2] : cc[0] + n0 * cc[1]);
            const int64_t slot = c2 * md.halo_per_cube + md.face_ord[kn][md.nb_face[k][f]];
            e[f] = (int32_t)(-2 - slot);
          }                                               // else: -1, the domain boundary
        }
      }
}

}  // namespace sg
