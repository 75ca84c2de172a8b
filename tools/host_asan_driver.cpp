// CPU sanitizer driver (SURVEY 5 "sanitizers"; `make -C seigen_amd/csrc host-asan`): everything of libseigen_hip that
// needs no device - reference elements, mesh tables, MFMA fragment tables, the device-free C-ABI entry points - built with
// -fsanitize=address,undefined and walked over every (dim, degree, cell type, diagonal) the library accepts, plus the
// argument errors the entry points must refuse.  Exit code 0 and no sanitizer report = clean.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hostlogic.hpp"
#include "kernels.hpp"
#include "mfma_tables.hpp"
#include "sponge_tables.hpp"

using namespace sg;

static int nfail = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      nfail += 1;                                                           \
    }                                                                       \
  } while (0)

static double checksum(const std::vector<double>& v) {
  double s = 0;
  for (double x : v) s += std::fabs(x);
  return s;
}

static void reference_operators(int cell_type, int dim, int degree) {
  for (int which = 0; which <= 4; ++which) {
    const int q = which == 3 ? 4 : 0;
    const int64_t n = sg_reference_operator_cell(cell_type, dim, degree, which, q, nullptr, 0);
    EXPECT(n > 0);
    if (n <= 0) continue;
    std::vector<double> v((size_t)n);
    EXPECT(sg_reference_operator_cell(cell_type, dim, degree, which, q, v.data(), v.size() * sizeof(double)) == n);
    EXPECT(std::isfinite(checksum(v)));
    // a wrong size must be refused, not written through
    EXPECT(sg_reference_operator_cell(cell_type, dim, degree, which, q, v.data(), (v.size() - 1) * sizeof(double)) == SG_ERR_ARG);
  }
  // tabulation at the lattice points = the identity (Lagrange basis), degrees up to 8
  for (int P = 1; P <= 8; ++P) {
    std::vector<int> lat;
    lattice_points(dim, P, lat, cell_type);
    const int nd = num_nodes(dim, P, cell_type);
    std::vector<double> xi((size_t)nd * dim), phi((size_t)nd * nd);
    for (size_t i = 0; i < xi.size(); ++i) xi[i] = (double)lat[i] / P;
    EXPECT(sg_tabulate_cell(cell_type, dim, P, nd, xi.data(), phi.data()) == SG_OK);
    double worst = 0;
    for (int p = 0; p < nd; ++p)
      for (int a = 0; a < nd; ++a) worst = std::fmax(worst, std::fabs(phi[(size_t)p * nd + a] - (p == a ? 1.0 : 0.0)));
    EXPECT(worst < 1e-9);
  }
}

static void mesh_tables(int dim, int degree, int diagonal) {
  const int kind = diagonal == SG_DIAGONAL_QUAD ? KIND_TENSOR : KIND_SIMPLEX;
  RefElem re = make_refelem(dim, degree, kind);
  const double h[3] = {0.5, 0.25, 2.0};
  std::vector<int32_t> nb((size_t)MAX_CLS * MAX_FACES * 5), nbn((size_t)MAX_CLS * MAX_FACES * (MAX_NF + 1));
  std::vector<double> cn((size_t)MAX_CLS * MAX_FACES * 3), jinv((size_t)MAX_CLS * 9);
  EXPECT(sg_mesh_tables(dim, degree, diagonal, h, nb.data(), nbn.data(), cn.data(), jinv.data()) == SG_OK);
}

static void mfma_tables_3d(int degree) {
  RefElem re = make_refelem(3, degree, KIND_SIMPLEX);
  EXPECT(!mfma_frags_F(re).empty() && !mfma_frags_G(re).empty() && !mfma_frags_L(re).empty());
  EXPECT(!mfma32_frags_F(re).empty() && !mfma32_frags_G(re).empty() && !mfma32_frags_L(re).empty());
  if (degree >= 3) {
    std::vector<double> Q, Pm;
    mfma_factorise_D(re, Q, Pm);
    EXPECT(!mfma_frags_Q(re).empty() && !mfma_frags_P(re).empty());
  }
  // the per-block tables of the MFMA path on small blocks, with and without neighbour blocks, ragged in x
  const int shapes[4][3] = {{16, 2, 2}, {5, 3, 2}, {33, 1, 3}, {1, 1, 1}};
  for (const auto& n : shapes)
    for (int mask : {0, 63, 5}) {
      MeshDev md;
      std::memset(&md, 0, sizeof(md));
      md.nd = re.nd;
      md.nf = re.nf;
      const double h[3] = {1.0 / n[0], 1.0 / n[1], 1.0 / n[2]};
      build_mesh_tables(3, degree, 0, h, re.fnode.data(), re.lattice.data(), md);
      for (int a = 0; a < 3; ++a) md.n[a] = n[a];
      for (int s = 0; s < 6; ++s) md.has_nbr[s] = (mask >> s) & 1;
      md.gw = 16;
      md.ncube = (int64_t)n[0] * n[1] * n[2];
      md.ncube_pad = (md.ncube + 15) / 16 * 16;
      const MfmaConst mk = mfma_const(md);
      EXPECT(mk.ncube == md.ncube);
      std::vector<int32_t> ft, tab;
      mfma_trace_offsets(md, 9, ft);
      mfma_trace_offsets(md, 3, ft);
      build_nbr_table(md, tab);
      EXPECT(tab.size() == (size_t)(md.ncube_pad / 16) * 6 * 64);
    }
}

static void tile_tables_2d(int degree, int kind) {
  RefElem re = make_refelem(2, degree, kind);
  EXPECT(!tile2d_frags_V(re, 1.0).empty() && !tile2d_frags_V(re, -1.0).empty() && !tile2d_frags_L(re).empty());
  EXPECT(!tile2d_frags32_V(re, 1.0).empty() && !tile2d_frags32_L(re).empty());
  MeshDev md;
  std::memset(&md, 0, sizeof(md));
  md.nd = re.nd;
  md.nf = re.nf;
  const double h[3] = {0.1, 0.2, 1.0};
  build_mesh_tables(2, degree, kind == KIND_TENSOR ? SG_DIAGONAL_QUAD : 0, h, re.fnode.data(), re.lattice.data(), md);
  md.n[0] = 7;
  md.n[1] = 3;
  md.n[2] = 1;
  md.gw = 16;
  md.ncube = 21;
  md.ncube_pad = 32;
  (void)tile2d_const(md);
}

static void regions_and_coords() {
  for (int dim = 1; dim <= 3; ++dim)
    for (int diagonal : {0, 1, 2}) {
      if (diagonal == 2 && dim == 1) continue;
      for (int degree = 1; degree <= 4; ++degree)
        for (int mask = 0; mask < (1 << (2 * dim)); mask += (dim == 3 ? 7 : 1)) {
          sg_config cfg;
          std::memset(&cfg, 0, sizeof(cfg));
          cfg.dim = dim;
          cfg.degree = degree;
          cfg.n[0] = 19;
          cfg.n[1] = dim > 1 ? 5 : 1;
          cfg.n[2] = dim > 2 ? 4 : 1;
          for (int a = 0; a < 3; ++a) cfg.h[a] = 0.5 + a;
          cfg.diagonal = diagonal;
          cfg.nbr_mask = mask;
          cfg.cube0[0] = 3;
          (void)choose_kernel_path(cfg);
          for (int region = 0; region <= 4; ++region) {
            int32_t boxes[SG_MAX_REGION_BOXES * 6];
            const int nb = sg_region_boxes(&cfg, region, boxes, SG_MAX_REGION_BOXES);
            EXPECT(nb >= 0 && nb <= SG_MAX_REGION_BOXES);
            EXPECT(sg_region_boxes(&cfg, region, boxes, 0) == nb);   // count only: nothing written
          }
          if (mask == 0) {
            sg_config small = cfg;
            small.n[0] = 3;
            small.n[1] = dim > 1 ? 2 : 1;
            small.n[2] = dim > 2 ? 2 : 1;
            NodeGeom G;
            EXPECT(G.init(&small, degree));
            const size_t cells = (size_t)small.n[0] * small.n[1] * small.n[2] * G.ncls;
            std::vector<double> X(cells * G.nq * dim);
            EXPECT(sg_block_node_coords(&small, degree, X.data(), X.size() * sizeof(double)) == SG_OK);
            EXPECT(sg_block_node_coords(&small, degree, X.data(), X.size() * sizeof(double) - 8) == SG_ERR_ARG);
          }
        }
    }
  EXPECT(sg_region_boxes(nullptr, 0, nullptr, 0) == SG_ERR_ARG);
}

// sg_set_absorption's host half (csrc/sponge_tables.cpp): every kind of cell - none, constant, affine, general - side by side on
// ragged blocks (a last group with padding), every family flavour (scalar or not, pre-pass or not, records or lines)
static void sponge_plans(int dim, int degree, int kind, int q) {
  const int nd = num_nodes(dim, degree, kind), nq = num_nodes(dim, q, kind);
  std::vector<int> latQ;
  lattice_points(dim, q, latQ, kind);
  const int ncls = kind == KIND_TENSOR ? 1 : (dim == 1 ? 1 : (dim == 2 ? 2 : 6));
  for (int gw : {1, 16, 64}) {
    const int64_t ncube = 37, ncells = ncube * ncls;      // 37 cubes: the last group is padded at gw = 16 and 64
    std::vector<double> sigma((size_t)ncells * nq, 0.0);
    unsigned seed = 12345u + (unsigned)(dim * 100 + degree * 10 + q);
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (double)(seed >> 8) / (double)(1u << 24); };
    std::vector<int> want((size_t)ncells);
    for (int64_t e = 0; e < ncells; ++e) {
      const int what = (int)(rnd() * 4.0) & 3;      // 0 none, 1 constant, 2 affine, 3 general
      want[(size_t)e] = what;
      const double s0 = 1.0 + 9.0 * rnd(), g[3] = {rnd() - 0.5, rnd() - 0.5, rnd() - 0.5};
      for (int c = 0; c < nq; ++c) {
        double v = 0.0;
        if (what == 1) v = s0;
        if (what == 2) {
          v = s0;
          for (int k = 0; k < dim; ++k) v += g[k] * (double)latQ[(size_t)c * dim + k] / q;
        }
        if (what == 3) v = 10.0 * rnd();
        sigma[(size_t)e * nq + c] = v;
      }
    }
    for (int flavour = 0; flavour < 5; ++flavour) {
      SpongeRequest rq;
      rq.dim = dim; rq.degree = degree; rq.kind = kind; rq.sigma_degree = q; rq.ncells = ncells; rq.ncls = ncls; rq.gw = gw;
      rq.want_scalar = flavour >= 1;
      rq.pre_family = flavour >= 2;
      rq.try_affine = flavour >= 3;
      rq.line_layout = flavour == 4;
      const SpongePlan pl = plan_sponge(rq, sigma.data());
      EXPECT((int64_t)pl.slot.size() == ncells && pl.B.size() == (size_t)pl.nmat * nd * nd);
      int naff = 0;
      for (int64_t e = 0; e < ncells; ++e) {
        const int what = want[(size_t)e], sl = pl.slot[(size_t)e];
        const bool scalar = rq.want_scalar && what == 1;
        EXPECT((sl >= 0) == (what != 0 && !scalar));
        if (rq.want_scalar) {
          const double sg_ = pl.sig[(size_t)e];
          EXPECT(what == 0 ? sg_ == 0.0 : (what == 1 ? sg_ == sigma[(size_t)e * nq] : sg_ != sg_));
        }
        if (sl < 0) continue;
        EXPECT(sl < pl.nslots);
        if (!rq.pre_family) continue;
        EXPECT(pl.cells[(size_t)sl] == (int32_t)e);
        if (rq.line_layout) EXPECT(sl % gw == (int)((e / ncls) % gw));      // the cell's column of its item
        const bool affine = pl.mat_of[(size_t)sl] < 0;
        // q = 1 on a simplex: every sigma is affine; a constant cell of a family without scalars is affine too
        const bool must = what == 2 || (what == 1 && !rq.want_scalar) || (what == 3 && q == 1 && kind == KIND_SIMPLEX);
        EXPECT(affine == (rq.try_affine && must));
        if (affine) {
          naff += 1;
          for (int c = 0; c < nq; ++c) {      // the coefficients reproduce the nodal sigma
            double v = pl.aff_coef[(size_t)sl * (dim + 1)];
            for (int k = 0; k < dim; ++k) v += pl.aff_coef[(size_t)sl * (dim + 1) + 1 + k] * (double)latQ[(size_t)c * dim + k] / q;
            EXPECT(std::fabs(v - sigma[(size_t)e * nq + c]) < 1e-12);
          }
        } else {
          EXPECT(pl.mat_of[(size_t)sl] < pl.nmat);
        }
      }
      EXPECT(naff == pl.naffine);
      if (rq.pre_family) {
        for (int32_t ms : pl.mat_slots) EXPECT(ms >= 0 && ms < pl.nslots && pl.mat_of[(size_t)ms] >= 0);
        EXPECT((int)pl.mat_slots.size() + pl.naffine == (int)std::count_if(pl.slot.begin(), pl.slot.end(), [](int32_t v) { return v >= 0; }));
      }
      if (pl.naffine > 0) {
        EXPECT(pl.X.size() == (size_t)dim * nd * pl.W && pl.col.size() == (size_t)nd * pl.W && pl.item_slots.size() == pl.items.size() * gw);
        // X_k is multiplication by xi_k followed by the L2 projection: it maps the constant 1 to xi_k at the nodes
        std::vector<int> latP;
        lattice_points(dim, degree, latP, kind);
        for (int k = 0; k < dim; ++k)
          for (int a = 0; a < nd; ++a) {
            double row = 0.0, ell = 0.0;
            for (int b = 0; b < nd; ++b) row += pl.Xd[((size_t)k * nd + a) * nd + b];
            for (int j = 0; j < pl.W; ++j) ell += pl.X[((size_t)k * nd + a) * pl.W + j];
            EXPECT(std::fabs(row - (double)latP[(size_t)a * dim + k] / degree) < 1e-10 && std::fabs(row - ell) < 1e-12);
          }
        int seen = 0;
        for (int32_t s : pl.item_slots) seen += s >= 0;
        EXPECT(seen == pl.naffine);
      }
    }
  }
}

int main() {
  for (int cell_type : {0, 1})
    for (int dim = 1; dim <= 3; ++dim)
      for (int degree = 1; degree <= 4; ++degree) {
        if (cell_type == 1 && dim == 1) {     // no tensor-product cell in 1-D (an interval is a simplex): refused
          EXPECT(sg_reference_operator_cell(1, 1, degree, 0, 0, nullptr, 0) == SG_ERR_ARG);
          continue;
        }
        reference_operators(cell_type, dim, degree);
      }
  for (int dim = 1; dim <= 3; ++dim)
    for (int degree = 1; degree <= 4; ++degree)
      for (int diagonal : {0, 1, 2}) {
        if (diagonal == 2 && dim == 1) continue;
        mesh_tables(dim, degree, diagonal);
      }
  for (int degree = 1; degree <= 4; ++degree) {
    mfma_tables_3d(degree);
    tile_tables_2d(degree, KIND_SIMPLEX);
    tile_tables_2d(degree, KIND_TENSOR);
  }
  regions_and_coords();
  for (int dim = 1; dim <= 3; ++dim)
    for (int degree : {1, 2, 4})
      for (int kind : {KIND_SIMPLEX, KIND_TENSOR}) {
        if (kind == KIND_TENSOR && dim == 1) continue;
        if (kind == KIND_TENSOR && dim == 3 && degree == 4) continue;      // 125^3 sponge tensor: minutes under the sanitizers
        for (int q : {1, 4}) sponge_plans(dim, degree, kind, q);
      }
  // arguments the entry points must refuse
  EXPECT(sg_reference_operator_cell(7, 2, 2, 0, 0, nullptr, 0) == SG_ERR_ARG);
  EXPECT(sg_reference_operator_cell(0, 2, 2, 9, 0, nullptr, 0) == SG_ERR_ARG);
  EXPECT(sg_tabulate_cell(0, 4, 2, 1, nullptr, nullptr) == SG_ERR_ARG);
  EXPECT(sg_mesh_tables(2, 2, 0, nullptr, nullptr, nullptr, nullptr, nullptr) == SG_ERR_ARG);
  std::printf("host_asan_driver: %s (%d failed expectations)\n", nfail ? "FAILED" : "clean", nfail);
  return nfail ? 1 : 0;
}
