// Translation-invariant tables of the structured simplicial mesh.
//
// [upstream] UnitSquareMesh / RectangleMesh / UnitCubeMesh (tests/eigenmode/
// eigenmode_2d.py:11, eigenmode_3d.py:11, tests/explosive_source/
// explosive_source_lf4.py:10) cut every square into 2 triangles and every cube
// into 6 Kuhn tetrahedra.  Each simplex "class" has a constant Jacobian, and the
// neighbour across each facet is a pure function of (cube index, class, facet):
// no connectivity arrays live in HBM.  This replaces PyOP2's cell->node /
// interior-facet maps and DMPlex topology for the hot path.
#pragma once
#include <cstdint>

namespace sg {

constexpr int SG_DIAGONAL_QUAD = 2;   // sg_config::diagonal: the squares are the cells (tensor-product element)
constexpr int MAX_CLS = 6;
constexpr int MAX_FACES = 6;     // hexahedra; simplices use dim + 1, quadrilaterals 4
constexpr int MAX_NF = 27;     // facet nodes: 15 on a P4 triangle, 25 on a DQ_4 square (hexahedra); tables hold MAX_NF + 1

// POD mirrored in device memory (copied to LDS at workgroup start).
struct MeshDev {
  int32_t dim, P, nd, nf, nfaces, ncls;
  int32_t n[3];          // cubes per axis of this block
  int32_t has_nbr[6];    // side (2*axis+hi) touches another block
  int32_t halo_per_cube; // cell-facets per boundary cube on one side (1, 1, 2 for dim 1, 2, 3)
  // HBM layout of a field: cubes are grouped `gw` at a time along the linear cube index and
  // the group's cells are interleaved:
  //   offset(cube c, class k, node b, comp) = ((((c/gw)*ncls + k)*nd + b)*ncomp + comp)*gw + c%gw
  // gw = 1 is the host (reference) layout [cell][node][comp]; gw = 16 makes 16 cells' values
  // of one (node, comp) a 128-byte line = one MFMA B-operand row / accumulator row.
  int32_t gw;
  int64_t ncube;      // cubes in this block
  int64_t ncube_pad;  // rounded up to a multiple of gw
  double Jinv[MAX_CLS][3][3];        // [cls][r][j] = d xi_r / d x_j
  double cn[MAX_CLS][MAX_FACES][3];  // (|F|/|detJ|) * outward normal
  int32_t nb_axis[MAX_CLS][MAX_FACES];  // axis crossed by the facet, -1: neighbour in the same cube
  int32_t nb_dir[MAX_CLS][MAX_FACES];   // +1 / -1
  int32_t nb_cls[MAX_CLS][MAX_FACES];   // neighbour's class
  int32_t nb_face[MAX_CLS][MAX_FACES];  // neighbour's local facet
  int32_t face_ord[MAX_CLS][MAX_FACES]; // ordinal of this facet among the cube's facets on that side
  int8_t side_cls[6][2], side_face[6][2]; // inverse: the (class, facet) with ordinal o on cube side s
  uint8_t nb_node[MAX_CLS][MAX_FACES][MAX_NF + 1];  // neighbour ELEMENT node matching my facet node b
  uint8_t nb_fnode[MAX_CLS][MAX_FACES][MAX_NF + 1]; // same, as position in the neighbour's facet list
  uint8_t fnode[MAX_FACES][MAX_NF + 1];             // my element node of facet node b
};

// class vertex offsets, in cells: off[cls][vertex][axis]
void class_vertices(int dim, int diagonal, int& ncls, int off[MAX_CLS][4][3]);

// fills everything except n / has_nbr
void build_mesh_tables(int dim, int P, int diagonal, const double h[3], const int* fnode /*[nfaces][nf]*/,
                       const int* lattice /*[nd][dim]*/, MeshDev& md);

}  // namespace sg
