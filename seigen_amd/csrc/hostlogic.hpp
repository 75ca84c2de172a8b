// Host logic of libseigen_hip that needs no device and no HIP header: which kernel family runs a block, the boxes of
// the regions of a split stage, node coordinates of a block.  Defined in hostapi.cpp, which - with refelem.cpp,
// mesh_tables.cpp and mfma_tables.cpp - also builds on its own for the CPU sanitizer target (`make host-asan`).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/seigen_hip.h"
#include "mesh_tables.hpp"
#include "refelem.hpp"

extern std::string g_create_err;   // message of the last failed sg_create / device-free query

// smallest 2-D block (cells) that takes the MFMA tile kernels instead of the generic kernel: they win at every
// size measured, 40 x 40 squares included (tools/path_sweep2d.py, profiles/r02/path_sweep2d_tile_v2.txt)
static constexpr int64_t SG_TILE2D_MIN_CELLS = 0;

struct Box {
  int o[3], n[3];
};
struct KernelPath {
  bool mfma = false, lane = false, tile = false, hexm = false;
  int gw = 1;
};
KernelPath choose_kernel_path(const sg_config& cfg);
// shell thickness along x: the interleaved layouts put gw consecutive cubes of an x-row on the lanes of one item, so a
// one-cube shell next to an x side would use one lane in gw of every item it touches AND make the launch that owns the
// other gw - 1 lanes run the same item again.  With whole groups in the shell no item is cut (2 x 2 x 2).
// ... unless that would leave the launch that runs beside the exchange less than half of the block's rows to work on
// (a block with neighbours on both x sides and n[0] <= 2 gw had an EMPTY interior: nothing overlapped the exchange), or
// the layout is the lane kernels' (64 cubes per item: the shell would swallow blocks up to 128 cubes wide).  Then the
// shell is one cube thick again and the kernels mask the lanes of the groups it cuts (region_whole = false).
inline int shell_width_x(int gw, int n0, bool nbr_lo, bool nbr_hi) {
  if (gw <= 1 || gw >= 64) return 1;
  const int sides = (nbr_lo ? 1 : 0) + (nbr_hi ? 1 : 0);
  return 2 * (n0 - sides * gw) >= n0 ? gw : 1;
}
void region_boxes(int d, const int32_t n[3], const int32_t has_nbr[6], int region, std::vector<Box>& out, int xw);

// Node coordinates of a block: the affine image of the reference lattice under every cell's vertex map (one
// arithmetic, shared by sg_block_node_coords and the source box test of sg_set_source_box_ricker).
struct NodeGeom {
  int d = 0, degree = 0, nq = 0, ncls = 0;
  std::vector<int> lat;
  int off[sg::MAX_CLS][4][3];
  const sg_config* cfg = nullptr;
  bool init(const sg_config* c, int deg) {
    cfg = c;
    d = c->dim;
    degree = deg;
    const bool quad = c->diagonal == sg::SG_DIAGONAL_QUAD;
    if (quad && d != 2 && d != 3) return false;
    const int kind = quad ? sg::KIND_TENSOR : sg::KIND_SIMPLEX;
    sg::lattice_points(d, degree, lat, kind);
    nq = sg::num_nodes(d, degree, kind);
    if (quad) {   // vertex 0 the low corner, vertex 1 / 2 / 3 one cell along x / y / z: the affine map of the unit square / cube
      std::memset(off, 0, sizeof(off));
      ncls = 1;
      for (int m = 0; m < d; ++m) off[0][m + 1][m] = 1;
    } else {
      sg::class_vertices(d, c->diagonal, ncls, off);
    }
    return true;
  }
  // coordinates of node a of the cell of class k in cube c
  void node(const int c[3], int k, int a, double x[3]) const {
    double X[4][3];
    for (int v = 0; v <= d; ++v)
      for (int i = 0; i < d; ++i) X[v][i] = cfg->origin[i] + (double)(cfg->cube0[i] + c[i] + off[k][v][i]) * cfg->h[i];
    for (int i = 0; i < d; ++i) {
      double xv = X[0][i];
      for (int m = 0; m < d; ++m) xv += (X[m + 1][i] - X[0][i]) * ((double)lat[a * d + m] / (double)degree);
      x[i] = xv;
    }
  }
};

