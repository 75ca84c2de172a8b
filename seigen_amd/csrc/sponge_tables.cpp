// plan_sponge: what every cell gets for the absorption term (sponge_tables.hpp).  Plain C++: no device, no HIP header.
#include "sponge_tables.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>
#include <unordered_map>

namespace sg {

bool sponge_affine_fit(const double* sigma, const std::vector<int>& latQ, int nq, int dim, int q, const int* vtx, double* s) {
  // the fit from the vertex nodes, the verdict from all nodes
  s[0] = sigma[vtx[0]];
  double big = std::fabs(s[0]);
  for (int k = 0; k < dim; ++k) {
    s[1 + k] = sigma[vtx[1 + k]] - s[0];
    big = std::max(big, std::fabs(sigma[vtx[1 + k]]));
  }
  const double tol = 64 * std::numeric_limits<double>::epsilon() * big;
  for (int c = 0; c < nq; ++c) {
    double v = s[0];
    for (int k = 0; k < dim; ++k) v += s[1 + k] * ((double)latQ[(size_t)c * dim + k] / q);
    if (!(std::fabs(v - sigma[c]) <= tol)) return false;      // (a NaN in sigma is never affine)
  }
  return true;
}

SpongePlan plan_sponge(const SpongeRequest& rq, const double* sigma_nodes) {
  if (!sigma_nodes || rq.dim < 1 || rq.dim > 3 || rq.degree < 1 || rq.sigma_degree < 1 || rq.ncells < 0 || rq.ncls < 1 || rq.gw < 1)
    throw std::invalid_argument("bad sponge request");
  if (rq.ncells >= ((int64_t)1 << 31)) throw std::invalid_argument("cell indices of the sponge tables are 32-bit");
  const int d = rq.dim, q = rq.sigma_degree;
  const int nd = num_nodes(d, rq.degree, rq.kind), nq = num_nodes(d, q, rq.kind);
  SpongePlan pl;
  pl.slot.assign((size_t)rq.ncells, -1);
  if (rq.want_scalar) pl.sig.assign((size_t)rq.ncells, 0.0);
  // B_e[a][b] = sum_c A[a][c][b] sigma_{e,c}
  const std::vector<double> A = sponge_tensor(d, rq.degree, q, rq.kind);
  const bool try_affine = rq.pre_family && rq.try_affine;
  std::vector<int> latQ;
  int vtx[4] = {0, 0, 0, 0};
  if (try_affine) {
    lattice_points(d, q, latQ, rq.kind);
    for (int c = 0; c < nq; ++c) {      // the nodes at the origin and at q e_k: where the affine fit is read off
      int sum = 0, which = -1;
      for (int k = 0; k < d; ++k) {
        sum += latQ[(size_t)c * d + k];
        if (latQ[(size_t)c * d + k] == q) which = k;
      }
      if (sum == 0) vtx[0] = c;
      if (sum == q && which >= 0) vtx[1 + which] = c;
    }
  }
  // Cells with the same nodal sigma share one matrix (A is the reference element's): the strips of the reference's scripts
  // have a few dozen distinct edge cells, a sigma that depends on one coordinate n0 x classes - the matrix table stays in
  // the caches.  Families with a pre-pass number their cells (slot -> cell, slot -> matrix); the others look the matrix up
  // by the slot itself.
  std::unordered_map<std::string, int32_t> mat_id;
  for (int64_t e = 0; e < rq.ncells; ++e) {
    const double* sg_ = sigma_nodes + (size_t)e * nq;
    bool nz = false, same = true;
    for (int c = 0; c < nq; ++c) {
      nz = nz || (sg_[c] != 0.0);
      same = same && (sg_[c] == sg_[0]);
    }
    if (!nz) continue;
    if (rq.want_scalar) {
      pl.sig[(size_t)e] = same ? sg_[0] : std::numeric_limits<double>::quiet_NaN();
      if (same) continue;
    }
    if (try_affine) {
      double sfit[4] = {0, 0, 0, 0};
      if (sponge_affine_fit(sg_, latQ, nq, d, q, vtx, sfit)) {
        pl.aff_coef.resize((size_t)(pl.nslots + 1) * (d + 1), 0.0);
        for (int k = 0; k <= d; ++k) pl.aff_coef[(size_t)pl.nslots * (d + 1) + k] = sfit[k];
        pl.slot[(size_t)e] = pl.nslots++;
        pl.mat_of.push_back(-1);
        pl.naffine += 1;
        continue;
      }
    }
    const std::string key(reinterpret_cast<const char*>(sg_), (size_t)nq * sizeof(double));
    auto found = mat_id.find(key);
    int32_t m;
    if (found != mat_id.end()) {
      m = found->second;
    } else {
      m = pl.nmat++;
      mat_id.emplace(key, m);
      const size_t base = pl.B.size();
      pl.B.resize(base + (size_t)nd * nd, 0.0);
      for (int a = 0; a < nd; ++a)
        for (int c = 0; c < nq; ++c) {
          const double s = sg_[c];
          if (s == 0.0) continue;
          const double* Arow = &A[((size_t)a * nq + c) * nd];
          double* Brow = &pl.B[base + (size_t)a * nd];
          for (int b = 0; b < nd; ++b) Brow[b] += Arow[b] * s;
        }
    }
    if (rq.pre_family) {
      pl.mat_slots.push_back(pl.nslots);
      pl.slot[(size_t)e] = pl.nslots++;
      pl.mat_of.push_back(m);
    } else {
      pl.slot[(size_t)e] = m;
      pl.nslots = pl.nmat;
    }
  }
  if (!rq.pre_family) return pl;
  pl.aff_coef.resize(pl.naffine > 0 ? (size_t)pl.nslots * (d + 1) : 0, 0.0);

  const int gw = rq.gw, ncls = rq.ncls;
  const int64_t ncube = (rq.ncells + ncls - 1) / ncls, ngroups = (ncube + gw - 1) / gw;
  auto cell_of = [&](int64_t g, int k, int w) -> int64_t {      // -1: padding of the last group
    const int64_t c = g * gw + w;
    return c < ncube && c * ncls + k < rq.ncells ? c * ncls + k : -1;
  };
  // line layout: a cell's slot becomes (index of its item among the items that hold a slot) * gw + its column
  if (rq.line_layout && pl.nslots > 0) {
    std::vector<int32_t> renum((size_t)pl.nslots, -1);
    int64_t nit = 0;
    for (int64_t g = 0; g < ngroups; ++g)
      for (int k = 0; k < ncls; ++k) {
        bool any = false;
        for (int w = 0; w < gw; ++w) {
          const int64_t e = cell_of(g, k, w);
          if (e >= 0 && pl.slot[(size_t)e] >= 0) {
            renum[(size_t)pl.slot[(size_t)e]] = (int32_t)(nit * gw + w);
            any = true;
          }
        }
        if (any) nit += 1;
      }
    if (nit * gw >= ((int64_t)1 << 31)) throw std::invalid_argument("too many sponge items for 32-bit slots");
    const int32_t nnew = (int32_t)(nit * gw);
    std::vector<int32_t> mat2((size_t)nnew, -1);
    std::vector<double> coef2(pl.aff_coef.empty() ? 0 : (size_t)nnew * (d + 1), 0.0);
    for (int32_t o = 0; o < pl.nslots; ++o) {
      mat2[(size_t)renum[(size_t)o]] = pl.mat_of[(size_t)o];
      if (!coef2.empty())
        for (int k = 0; k <= d; ++k) coef2[(size_t)renum[(size_t)o] * (d + 1) + k] = pl.aff_coef[(size_t)o * (d + 1) + k];
    }
    for (int32_t& ms : pl.mat_slots) ms = renum[(size_t)ms];
    std::sort(pl.mat_slots.begin(), pl.mat_slots.end());
    for (int64_t e = 0; e < rq.ncells; ++e)
      if (pl.slot[(size_t)e] >= 0) pl.slot[(size_t)e] = renum[(size_t)pl.slot[(size_t)e]];
    pl.mat_of.swap(mat2);
    pl.aff_coef.swap(coef2);
    pl.nslots = nnew;
  }
  pl.cells.assign((size_t)pl.nslots, 0);
  for (int64_t e = 0; e < rq.ncells; ++e)
    if (pl.slot[(size_t)e] >= 0) pl.cells[(size_t)pl.slot[(size_t)e]] = (int32_t)e;
  if (pl.naffine == 0) return pl;

  // X_k = Mhat^-1 int xi_k phi_a phi_b = sum_c xi_k(c) A1[a][c][b] over the nodes c of the degree-1 element (xi_k is one of its
  // basis functions on a simplex, a sum of them on a tensor-product cell)
  const std::vector<double> A1 = sponge_tensor(d, rq.degree, 1, rq.kind);
  std::vector<int> lat1;
  lattice_points(d, 1, lat1, rq.kind);
  const int n1 = num_nodes(d, 1, rq.kind);
  pl.Xd.assign((size_t)d * nd * nd, 0.0);
  for (int k = 0; k < d; ++k)
    for (int a = 0; a < nd; ++a)
      for (int c = 0; c < n1; ++c) {
        if (lat1[(size_t)c * d + k] == 0) continue;
        for (int b = 0; b < nd; ++b) pl.Xd[((size_t)k * nd + a) * nd + b] += A1[((size_t)a * n1 + c) * nd + b];
      }
  // rows in ELL form over the union of the d patterns: dense on simplices, d P + 1 entries on tensor-product cells
  // (multiplication by xi_k acts along one line of the cell)
  std::vector<std::vector<int32_t>> pat((size_t)nd);
  pl.W = 0;
  for (int a = 0; a < nd; ++a) {
    for (int b = 0; b < nd; ++b) {
      bool any = false;
      for (int k = 0; k < d; ++k) any = any || std::fabs(pl.Xd[((size_t)k * nd + a) * nd + b]) > 1e-14;
      if (any) pat[(size_t)a].push_back(b);
    }
    pl.W = std::max(pl.W, (int)pat[(size_t)a].size());
  }
  pl.dense = pl.W == nd;
  pl.X.assign((size_t)d * nd * pl.W, 0.0);
  pl.col.assign((size_t)nd * pl.W, 0);
  for (int a = 0; a < nd; ++a)
    for (int j = 0; j < pl.W; ++j) {
      const bool real = pl.dense || j < (int)pat[(size_t)a].size();
      const int b = pl.dense ? j : (real ? pat[(size_t)a][(size_t)j] : a);      // padding: a zero entry on the diagonal
      pl.col[(size_t)a * pl.W + j] = b;
      for (int k = 0; k < d; ++k) pl.X[((size_t)k * nd + a) * pl.W + j] = real ? pl.Xd[((size_t)k * nd + a) * nd + b] : 0.0;
    }
  // the items that hold an affine cell, and their cells' slots
  for (int64_t g = 0; g < ngroups; ++g)
    for (int k = 0; k < ncls; ++k) {
      auto affine_slot = [&](int w) -> int32_t {
        const int64_t e = cell_of(g, k, w);
        if (e < 0) return -1;
        const int32_t s = pl.slot[(size_t)e];
        return (s >= 0 && pl.mat_of[(size_t)s] < 0) ? s : -1;
      };
      bool any = false;
      for (int w = 0; w < gw && !any; ++w) any = affine_slot(w) >= 0;
      if (!any) continue;
      pl.items.push_back((int32_t)(g * ncls + k));
      for (int w = 0; w < gw; ++w) pl.item_slots.push_back(affine_slot(w));
    }
  return pl;
}

}  // namespace sg
