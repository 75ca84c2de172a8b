// CPU sanitizer driver (SURVEY 5 "sanitizers"; `make -C seigen_amd/csrc host-asan`): everything of libseigen_hip that
// needs no device - reference elements, mesh tables, MFMA fragment tables, the device-free C-ABI entry points - built with
// -fsanitize=address,undefined and walked over every (dim, degree, cell type, diagonal) the library accepts, plus the
// argument errors the entry points must refuse.  Exit code 0 and no sanitizer report = clean.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hostlogic.hpp"
#include "kernels.hpp"
#include "mfma_tables.hpp"

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
  // arguments the entry points must refuse
  EXPECT(sg_reference_operator_cell(7, 2, 2, 0, 0, nullptr, 0) == SG_ERR_ARG);
  EXPECT(sg_reference_operator_cell(0, 2, 2, 9, 0, nullptr, 0) == SG_ERR_ARG);
  EXPECT(sg_tabulate_cell(0, 4, 2, 1, nullptr, nullptr) == SG_ERR_ARG);
  EXPECT(sg_mesh_tables(2, 2, 0, nullptr, nullptr, nullptr, nullptr, nullptr) == SG_ERR_ARG);
  std::printf("host_asan_driver: %s (%d failed expectations)\n", nfail ? "FAILED" : "clean", nfail);
  return nfail ? 1 : 0;
}
