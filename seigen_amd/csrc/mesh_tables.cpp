#include "mesh_tables.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace sg {

void class_vertices(int dim, int diagonal, int& ncls, int off[MAX_CLS][4][3]) {
  std::memset(off, 0, sizeof(int) * MAX_CLS * 4 * 3);
  if (dim == 1) {
    ncls = 1;
    off[0][1][0] = 1;
  } else if (dim == 2) {
    ncls = 2;
    // "left": the diagonal runs from (i, j+1) to (i+1, j) (Firedrake's default);
    // "right": from (i, j) to (i+1, j+1).
    static const int left[2][3][2] = {{{0, 0}, {1, 0}, {0, 1}}, {{1, 1}, {0, 1}, {1, 0}}};
    static const int right[2][3][2] = {{{0, 0}, {1, 0}, {1, 1}}, {{0, 0}, {1, 1}, {0, 1}}};
    for (int c = 0; c < 2; ++c)
      for (int v = 0; v < 3; ++v)
        for (int a = 0; a < 2; ++a) off[c][v][a] = diagonal ? right[c][v][a] : left[c][v][a];
  } else {
    ncls = 6;
    // Kuhn split: one tetrahedron per permutation of the axes (lexicographic order),
    // vertices 0, e_p0, e_p0+e_p1, (1,1,1): all six share the cube's main diagonal.
    int perm[3] = {0, 1, 2};
    int c = 0;
    do {
      int cur[3] = {0, 0, 0};
      for (int v = 1; v <= 3; ++v) {
        cur[perm[v - 1]] += 1;
        for (int a = 0; a < 3; ++a) off[c][v][a] = cur[a];
      }
      ++c;
    } while (std::next_permutation(perm, perm + 3));
  }
}

static void invert_small(int d, const double J[3][3], double Ji[3][3], double& det) {
  if (d == 1) {
    det = J[0][0];
    Ji[0][0] = 1.0 / J[0][0];
  } else if (d == 2) {
    det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    Ji[0][0] = J[1][1] / det;
    Ji[0][1] = -J[0][1] / det;
    Ji[1][0] = -J[1][0] / det;
    Ji[1][1] = J[0][0] / det;
  } else {
    det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
          J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
    Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
    Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
    Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
    Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
    Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
    Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
    Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
    Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  }
}

// Tensor-product cells (diagonal == SG_DIAGONAL_QUAD): the squares / cubes themselves, one class, 2 dim facets
// (0: x = 0, 1: x = 1, 2: y = 0, 3: y = 1, 4: z = 0, 5: z = 1, refelem.hpp); facet nodes count the transverse lattice
// coordinates on both sides of a facet, so facet node b meets the neighbour's facet node b.
static void build_quad_tables(int dim, int P, const double h[3], const int* fnode, MeshDev& md) {
  md.dim = dim;
  md.P = P;
  md.ncls = 1;
  md.nfaces = 2 * dim;
  md.halo_per_cube = 1;
  const int nf = md.nf;
  for (int f = 0; f < md.nfaces; ++f)
    for (int b = 0; b < nf; ++b) md.fnode[f][b] = (uint8_t)fnode[f * nf + b];
  for (int r = 0; r < 3; ++r)
    for (int j = 0; j < 3; ++j) md.Jinv[0][r][j] = (r == j && r < dim) ? 1.0 / h[r] : 0.0;
  for (int f = 0; f < md.nfaces; ++f) {
    const int axis = f / 2, dir = (f % 2) ? 1 : -1;
    for (int j = 0; j < 3; ++j) md.cn[0][f][j] = (j == axis) ? dir / h[axis] : 0.0;   // |F| / |K| = 1 / h_axis
    md.nb_axis[0][f] = axis;
    md.nb_dir[0][f] = dir;
    md.nb_cls[0][f] = 0;
    md.nb_face[0][f] = f ^ 1;
    md.face_ord[0][f] = 0;
    md.side_cls[f][0] = 0;      // block side 2 * axis + (dir > 0) = f
    md.side_face[f][0] = (int8_t)f;
    for (int b = 0; b < nf; ++b) {
      md.nb_node[0][f][b] = md.fnode[f ^ 1][b];
      md.nb_fnode[0][f][b] = (uint8_t)b;
    }
  }
}

void build_mesh_tables(int dim, int P, int diagonal, const double h[3], const int* fnode, const int* lattice,
                       MeshDev& md) {
  if (diagonal == SG_DIAGONAL_QUAD) {
    if (dim != 2 && dim != 3) throw std::runtime_error("tensor-product cells: quadrilaterals (2-D) and hexahedra (3-D)");
    if (md.nf > MAX_NF) throw std::runtime_error("tensor-product cells: too many facet nodes for the mesh tables");
    build_quad_tables(dim, P, h, fnode, md);
    return;
  }
  int off[MAX_CLS][4][3];
  int ncls;
  class_vertices(dim, diagonal, ncls, off);
  const int nfaces = dim + 1;
  md.dim = dim;
  md.P = P;
  md.ncls = ncls;
  md.nfaces = nfaces;
  const int nd = md.nd, nf = md.nf;
  double fact = 1.0;
  for (int i = 2; i <= dim - 1; ++i) fact *= i;
  md.halo_per_cube = (dim == 3) ? 2 : 1;

  for (int f = 0; f < nfaces; ++f)
    for (int b = 0; b < nf; ++b) md.fnode[f][b] = (uint8_t)fnode[f * nf + b];

  // geometry per class
  for (int c = 0; c < ncls; ++c) {
    double J[3][3] = {{0}}, Ji[3][3] = {{0}}, det;
    for (int i = 0; i < dim; ++i)
      for (int m = 0; m < dim; ++m) J[i][m] = (off[c][m + 1][i] - off[c][0][i]) * h[i];
    invert_small(dim, J, Ji, det);
    for (int r = 0; r < 3; ++r)
      for (int j = 0; j < 3; ++j) md.Jinv[c][r][j] = (r < dim && j < dim) ? Ji[r][j] : 0.0;
    for (int f = 0; f < nfaces; ++f) {
      // grad lambda_f: rows of Jinv for f>=1, minus their sum for f=0;
      // outward normal = -grad lambda_f / |grad lambda_f|, |F|/|detJ| = |grad lambda_f|/(dim-1)!
      double gl[3] = {0, 0, 0};
      for (int j = 0; j < dim; ++j) {
        if (f == 0)
          for (int r = 0; r < dim; ++r) gl[j] -= Ji[r][j];
        else
          gl[j] = Ji[f - 1][j];
      }
      for (int j = 0; j < 3; ++j) md.cn[c][f][j] = (j < dim) ? -gl[j] / fact : 0.0;
    }
  }

  // integer positions (units of h/P) of every node of every class, relative to the cube corner
  auto node_pos = [&](int c, int a, int pos[3]) {
    for (int i = 0; i < 3; ++i) pos[i] = 0;
    for (int i = 0; i < dim; ++i) {
      int p = P * off[c][0][i];
      for (int m = 0; m < dim; ++m) p += lattice[a * dim + m] * (off[c][m + 1][i] - off[c][0][i]);
      pos[i] = p;
    }
  };

  int ord_count[6] = {0, 0, 0, 0, 0, 0};
  for (int c = 0; c < ncls; ++c)
    for (int f = 0; f < nfaces; ++f) {
      // facet vertex set
      std::vector<std::array<int, 3>> fv;
      for (int v = 0; v <= dim; ++v)
        if (v != f) fv.push_back({off[c][v][0], off[c][v][1], off[c][v][2]});
      std::sort(fv.begin(), fv.end());
      bool found = false;
      int dcmax = 1;
      for (int dz = (dim > 2 ? -dcmax : 0); dz <= (dim > 2 ? dcmax : 0) && !found; ++dz)
        for (int dy = (dim > 1 ? -dcmax : 0); dy <= (dim > 1 ? dcmax : 0) && !found; ++dy)
          for (int dx = -dcmax; dx <= dcmax && !found; ++dx)
            for (int c2 = 0; c2 < ncls && !found; ++c2) {
              if (dx == 0 && dy == 0 && dz == 0 && c2 == c) continue;
              int dc[3] = {dx, dy, dz};
              for (int f2 = 0; f2 < nfaces && !found; ++f2) {
                std::vector<std::array<int, 3>> gv;
                for (int v = 0; v <= dim; ++v)
                  if (v != f2) gv.push_back({off[c2][v][0] + dc[0], off[c2][v][1] + dc[1], off[c2][v][2] + dc[2]});
                std::sort(gv.begin(), gv.end());
                if (gv != fv) continue;
                found = true;
                int nnz = (dx != 0) + (dy != 0) + (dz != 0);
                if (nnz > 1) throw std::runtime_error("facet neighbour crosses more than one axis");
                int axis = -1, dir = 0;
                for (int a = 0; a < 3; ++a)
                  if (dc[a] != 0) {
                    axis = a;
                    dir = dc[a];
                  }
                md.nb_axis[c][f] = axis;
                md.nb_dir[c][f] = dir;
                md.nb_cls[c][f] = c2;
                md.nb_face[c][f] = f2;
                // node matching
                for (int b = 0; b < nf; ++b) {
                  int pa[3];
                  node_pos(c, md.fnode[f][b], pa);
                  int match = -1, matchf = -1;
                  for (int b2 = 0; b2 < nf; ++b2) {
                    int pb[3];
                    node_pos(c2, md.fnode[f2][b2], pb);
                    if (pb[0] + P * dc[0] == pa[0] && pb[1] + P * dc[1] == pa[1] && pb[2] + P * dc[2] == pa[2]) {
                      match = md.fnode[f2][b2];
                      matchf = b2;
                    }
                  }
                  if (match < 0) throw std::runtime_error("facet node matching failed");
                  md.nb_node[c][f][b] = (uint8_t)match;
                  md.nb_fnode[c][f][b] = (uint8_t)matchf;
                }
              }
            }
      if (!found) throw std::runtime_error("facet neighbour not found");
      if (md.nb_axis[c][f] >= 0) {
        int side = 2 * md.nb_axis[c][f] + (md.nb_dir[c][f] > 0 ? 1 : 0);
        if (ord_count[side] < 2) {
          md.side_cls[side][ord_count[side]] = (int8_t)c;
          md.side_face[side][ord_count[side]] = (int8_t)f;
        }
        md.face_ord[c][f] = ord_count[side]++;
      } else {
        md.face_ord[c][f] = -1;
      }
    }
  for (int s = 0; s < 2 * dim; ++s)
    if (ord_count[s] != md.halo_per_cube) throw std::runtime_error("unexpected facet count on cube side");
  (void)nd;
}

}  // namespace sg
