// The device-free part of the C-ABI (include/seigen_hip.h "device-free setup queries") and the host logic behind
// sg_create that involves no device: kernel-family choice, regions of a split stage, node coordinates, the exports of the
// reference-element operators and mesh tables.  No HIP header, no HIP call: this file, refelem.cpp, mesh_tables.cpp and
// mfma_tables.cpp are what `make host-asan` builds with -fsanitize=address,undefined and runs on the CPU.
#include <algorithm>
#include <cstdlib>

#include "hostlogic.hpp"
#include "kernels.hpp"

using namespace sg;

std::string g_create_err;

// Which kernel family runs a block (and with it the layout's group width gw: 16 cubes per 128-byte line for the
// MFMA and tile kernels, 64 for the lane kernels, 1 = host layout for the generic kernel).
KernelPath choose_kernel_path(const sg_config& cfg) {
  KernelPath kp;
  if (cfg.diagonal == SG_DIAGONAL_QUAD) {
    // quadrilateral cells: the MFMA tile kernels (DQ_4: two 16-row tiles), or on request the table-driven generic
    // kernels (host layout); same size threshold and SEIGEN_HIP_PATH overrides as for triangles
    const char* pe = std::getenv("SEIGEN_HIP_PATH");
    const bool fg = pe && std::strcmp(pe, "generic") == 0, ft = pe && std::strcmp(pe, "tile") == 0;
    kp.tile = cfg.dim == 2 && tile2d_supported_quad(cfg.degree) && !fg &&
              (ft || (int64_t)cfg.n[0] * cfg.n[1] >= SG_TILE2D_MIN_CELLS / 2);
    // hexahedra (DQ_1, DQ_2): the sum-factorised lane-per-cell kernels (kernels_lane.hip hex_stage) from
    // SG_HEX_LANE_MIN_CELLS(degree) cubes up (below that the thread-per-node generic kernel has more parallelism);
    // SEIGEN_HIP_PATH=lane / generic forces one or the other
    const bool fl = pe && std::strcmp(pe, "lane") == 0;
    kp.lane = cfg.dim == 3 && lane_supported_hex(cfg.dim, cfg.degree) && !fg &&
              (fl || (int64_t)cfg.n[0] * cfg.n[1] * cfg.n[2] >= SG_HEX_LANE_MIN_CELLS(cfg.degree));
    // hexahedra DQ_3 / DQ_4: lines in registers, x lines on the matrix pipe (kernels_hexm.hip) at every size;
    // SEIGEN_HIP_PATH=generic: the thread-per-node kernel
    kp.hexm = cfg.dim == 3 && hexm_supported(cfg.dim, cfg.degree) && !fg;
    kp.gw = (kp.tile || kp.hexm) ? 16 : (kp.lane ? 64 : 1);
    return kp;
  }
  const int ncls = cfg.dim == 1 ? 1 : (cfg.dim == 2 ? 2 : 6);
  // kernel path: MFMA kernels (interleaved layout) where they exist, unless SEIGEN_HIP_PATH=generic
  const char* path_env = std::getenv("SEIGEN_HIP_PATH");
  const bool force_generic = path_env && std::strcmp(path_env, "generic") == 0;
  // 3-D: the MFMA kernels at every degree (degrees 1 and 2 use 4x4x4 tiles only); measured with
  // tools/path_sweep.py they beat the lane and generic kernels everywhere except degree 1 on blocks
  // under 65536 cells (SEIGEN_HIP_PATH=mfma forces them)
  const bool force_mfma = path_env && std::strcmp(path_env, "mfma") == 0;
  const int64_t ncube_all = (int64_t)cfg.n[0] * (cfg.dim > 1 ? cfg.n[1] : 1) * (cfg.dim > 2 ? cfg.n[2] : 1);
  const int64_t ncells_all = ncube_all * ncls;
  kp.mfma = mfma_supported(cfg.dim, cfg.degree) && !force_generic &&
            !(path_env && std::strcmp(path_env, "lane") == 0) &&
            (cfg.degree >= 2 || ncells_all >= 65536 || force_mfma);
  // lane-per-cell kernels need enough 64-cell groups to fill the chip; below that the
  // thread-per-node generic kernel has more parallelism (SEIGEN_HIP_PATH=lane forces them)
  const bool force_lane = path_env && std::strcmp(path_env, "lane") == 0;
  kp.lane = !kp.mfma && lane_supported(cfg.dim, cfg.degree) && !force_generic &&
            (force_lane || ncells_all >= (cfg.degree == 1 ? 196608 : 120000));  // crossovers measured
                                                                 // (tools/path_sweep.py, profiles/r02/small_2d_configs_negative_results.txt)
  // 2-D: the MFMA tile kernels (16 cells per wave, operators in registers) from SG_TILE2D_MIN_CELLS cells up
  // (measured crossover against the generic kernel, tools/path_sweep.py); SEIGEN_HIP_PATH=tile forces them
  const bool force_tile = path_env && std::strcmp(path_env, "tile") == 0;
  kp.tile = tile2d_supported(cfg.dim, cfg.degree) && !force_generic && !force_lane &&
            (force_tile || ncells_all >= SG_TILE2D_MIN_CELLS);
  if (kp.tile) kp.lane = false;
  kp.gw = (kp.mfma || kp.tile) ? 16 : (kp.lane ? 64 : 1);
  return kp;
}

namespace sg {
// which (dim, degree) each kernel family is instantiated for (kernels_mfma.hip, kernels_lane.hip, kernels_tile2d.hip)
bool mfma_supported(int dim, int P) { return dim == 3 && P >= 1 && P <= 4; }
// 3-D: only P1/P2 fit a lane's registers (P3/P4 take the MFMA path)
bool lane_supported(int dim, int P) { return ((dim == 1 || dim == 2) && P >= 1 && P <= 4) || (dim == 3 && (P == 1 || P == 2)); }
// hexahedra: DQ_1 and DQ_2 (27 nodes) fit a lane's registers one component at a time
bool lane_supported_hex(int dim, int P) { return dim == 3 && (P == 1 || P == 2); }
bool hexm_supported(int dim, int P) { return dim == 3 && (P == 3 || P == 4); }
bool tile2d_supported(int dim, int P) { return dim == 2 && P >= 1 && P <= 4; }
bool tile2d_supported_quad(int P) { return P >= 1 && P <= 4; }   // DQ_4 has 25 rows: two row tiles, one after the other
}  // namespace sg

void region_boxes(int d, const int32_t n[3], const int32_t has_nbr[6], int region, std::vector<Box>& out, int xw) {
  out.clear();
  int lo[3] = {0, 0, 0}, hi[3];
  for (int a = 0; a < 3; ++a) hi[a] = n[a];
  if (region == SG_REGION_ALL) {
    out.push_back(Box{{0, 0, 0}, {hi[0], hi[1], hi[2]}});
    return;
  }
  // interior: peel one cube (along x: one layout group of xw cubes, handle.hpp shell_width_x) off every side
  // that has a neighbour block
  int ilo[3] = {0, 0, 0}, ihi[3] = {hi[0], hi[1], hi[2]};
  for (int a = 0; a < d; ++a) {
    const int w = (a == 0 && xw > 1) ? xw : 1;
    if (has_nbr[2 * a]) ilo[a] = w < hi[a] ? w : hi[a];
    if (has_nbr[2 * a + 1]) ihi[a] = hi[a] - w > 0 ? hi[a] - w : 0;
    if (ihi[a] < ilo[a]) ihi[a] = ilo[a];
  }
  if (region == SG_REGION_INTERIOR) {
    out.push_back(Box{{ilo[0], ilo[1], ilo[2]}, {ihi[0] - ilo[0], ihi[1] - ilo[1], ihi[2] - ilo[2]}});
    return;
  }
  // FIRST / SECOND: the interior cut in two along the slowest axis (whole runs of the layout)
  const int ax = d - 1, mid = ilo[ax] + (ihi[ax] - ilo[ax]) / 2;
  if (region == SG_REGION_SECOND) {
    Box b{{ilo[0], ilo[1], ilo[2]}, {ihi[0] - ilo[0], ihi[1] - ilo[1], ihi[2] - ilo[2]}};
    b.o[ax] = mid;
    b.n[ax] = ihi[ax] - mid;
    out.push_back(b);
    return;
  }
  if (region == SG_REGION_FIRST) {
    Box b{{ilo[0], ilo[1], ilo[2]}, {ihi[0] - ilo[0], ihi[1] - ilo[1], ihi[2] - ilo[2]}};
    b.n[ax] = mid - ilo[ax];
    out.push_back(b);
  }
  // boundary shell = all \ interior, as disjoint slabs: peel axis by axis
  int clo[3] = {lo[0], lo[1], lo[2]}, chi[3] = {hi[0], hi[1], hi[2]};
  for (int a = 0; a < d; ++a) {
    if (ilo[a] > clo[a]) {
      Box b;
      for (int k = 0; k < 3; ++k) {
        b.o[k] = clo[k];
        b.n[k] = chi[k] - clo[k];
      }
      b.n[a] = ilo[a] - clo[a];
      out.push_back(b);
      clo[a] = ilo[a];
    }
    if (ihi[a] < chi[a] && ihi[a] >= clo[a]) {
      Box b;
      for (int k = 0; k < 3; ++k) {
        b.o[k] = clo[k];
        b.n[k] = chi[k] - clo[k];
      }
      b.o[a] = ihi[a];
      b.n[a] = chi[a] - ihi[a];
      out.push_back(b);
      chi[a] = ihi[a];
    }
  }
}

extern "C" {

int sg_abi_version(void) { return SG_ABI_VERSION; }

int sg_block_node_coords(const sg_config* cfg, int degree, double* out, size_t nbytes) {
  if (!cfg || !out || degree < 1 || degree > 8 || cfg->dim < 1 || cfg->dim > 3) return SG_ERR_ARG;
  NodeGeom G;
  if (!G.init(cfg, degree)) return SG_ERR_ARG;
  const int d = G.d;
  int n[3] = {1, 1, 1};
  for (int a = 0; a < d; ++a) n[a] = cfg->n[a];
  if (nbytes != (size_t)n[0] * n[1] * n[2] * G.ncls * G.nq * d * sizeof(double)) return SG_ERR_ARG;
  size_t o = 0;
  for (int ck = 0; ck < n[2]; ++ck)
    for (int cj = 0; cj < n[1]; ++cj)
      for (int ci = 0; ci < n[0]; ++ci) {
        const int c[3] = {ci, cj, ck};
        for (int k = 0; k < G.ncls; ++k)
          for (int a = 0; a < G.nq; ++a) {
            double x[3];
            G.node(c, k, a, x);
            for (int i = 0; i < d; ++i) out[o++] = x[i];
          }
      }
  return SG_OK;
}

int64_t sg_reference_operator(int dim, int degree, int which, int q, double* out, size_t nbytes) {
  return sg_reference_operator_cell(0, dim, degree, which, q, out, nbytes);
}

int64_t sg_reference_operator_cell(int cell_type, int dim, int degree, int which, int q, double* out, size_t nbytes) {
  std::vector<double> v;
  if (cell_type != KIND_SIMPLEX && cell_type != KIND_TENSOR) return SG_ERR_ARG;
  try {
    if (which == 3) {
      if (q < 1 || q > 6 || dim < 1 || dim > 3 || degree < 1 || degree > 4) return SG_ERR_ARG;
      v = sponge_tensor(dim, degree, q, cell_type);
    } else {
      RefElem re = make_refelem(dim, degree, cell_type);
      if (which == 0) v = re.D;
      else if (which == 1) v = re.L;
      else if (which == 2) v = re.Mhat;
      else if (which == 4) v.assign(re.fnode.begin(), re.fnode.end());
      else return SG_ERR_ARG;
    }
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return SG_ERR_ARG;
  }
  if (out) {
    if (nbytes != v.size() * sizeof(double)) return SG_ERR_ARG;
    std::memcpy(out, v.data(), nbytes);
  }
  return (int64_t)v.size();
}

int sg_tabulate(int dim, int degree, int64_t npts, const double* xi, double* phi) {
  return sg_tabulate_cell(0, dim, degree, npts, xi, phi);
}

int sg_tabulate_cell(int cell_type, int dim, int degree, int64_t npts, const double* xi, double* phi) {
  if (dim < 1 || dim > 3 || degree < 1 || degree > 8 || npts < 0 || !xi || !phi) return SG_ERR_ARG;
  if (cell_type != KIND_SIMPLEX && cell_type != KIND_TENSOR) return SG_ERR_ARG;
  tabulate(dim, degree, (int)npts, xi, phi, cell_type);
  return SG_OK;
}

int sg_mesh_tables(int dim, int degree, int diagonal, const double* h, int32_t* nb, int32_t* nb_node, double* cn,
                   double* jinv) {
  if (!h || !nb || !nb_node || !cn || !jinv) return SG_ERR_ARG;
  try {
    RefElem re = make_refelem(dim, degree, diagonal == SG_DIAGONAL_QUAD ? KIND_TENSOR : KIND_SIMPLEX);
    MeshDev md;
    std::memset(&md, 0, sizeof(md));
    md.nd = re.nd;
    md.nf = re.nf;
    double hh[3] = {1, 1, 1};
    for (int a = 0; a < dim; ++a) hh[a] = h[a];
    build_mesh_tables(dim, degree, diagonal, hh, re.fnode.data(), re.lattice.data(), md);
    for (int c = 0; c < md.ncls; ++c) {
      for (int f = 0; f < md.nfaces; ++f) {
        int32_t* o = nb + ((size_t)c * md.nfaces + f) * 5;
        o[0] = md.nb_axis[c][f];
        o[1] = md.nb_dir[c][f];
        o[2] = md.nb_cls[c][f];
        o[3] = md.nb_face[c][f];
        o[4] = md.face_ord[c][f];
        for (int b = 0; b < md.nf; ++b) nb_node[((size_t)c * md.nfaces + f) * md.nf + b] = md.nb_node[c][f][b];
        for (int j = 0; j < 3; ++j) cn[((size_t)c * md.nfaces + f) * 3 + j] = md.cn[c][f][j];
      }
      for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 3; ++j) jinv[((size_t)c * 3 + r) * 3 + j] = md.Jinv[c][r][j];
    }
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return SG_ERR_ARG;
  }
  return SG_OK;
}

int sg_region_boxes(const sg_config* cfg, int region, int32_t* boxes, int max_boxes) {
  if (!cfg || !boxes || cfg->dim < 1 || cfg->dim > 3 || region < 0 || region > 4 || max_boxes < 0) return SG_ERR_ARG;
  int32_t n[3] = {1, 1, 1}, has_nbr[6] = {0, 0, 0, 0, 0, 0};
  for (int a = 0; a < cfg->dim; ++a) n[a] = cfg->n[a];
  for (int s2 = 0; s2 < 2 * cfg->dim; ++s2) has_nbr[s2] = (cfg->nbr_mask >> s2) & 1;
  std::vector<Box> out;
  region_boxes(cfg->dim, n, has_nbr, region, out, shell_width_x(choose_kernel_path(*cfg).gw, n[0], has_nbr[0] != 0, has_nbr[1] != 0));
  int cnt = 0;
  for (const Box& b : out) {
    if (b.n[0] <= 0 || b.n[1] <= 0 || b.n[2] <= 0) continue;
    if (cnt < max_boxes)
      for (int k = 0; k < 3; ++k) {
        boxes[6 * cnt + k] = b.o[k];
        boxes[6 * cnt + 3 + k] = b.n[k];
      }
    cnt += 1;
  }
  return cnt;
}

}  // extern "C"
