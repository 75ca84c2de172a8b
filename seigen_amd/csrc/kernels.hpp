// Launch interface between the C-ABI host code (api.cpp, transfer.cpp, stages.cpp) and the HIP kernels.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mesh_tables.hpp"

namespace sg {

// SG_REGION_FIRST of a 3-D block with neighbours on all six sides = half of the interior + six shell slabs
constexpr int SG_MAX_BOXES = 7;

// Node / facet counts of the P_k simplex element, shared by the three kernel families
// (kernels.hip `Geo`, kernels_lane.hip `LG`, kernels_mfma.hip `MG`).
// TP = 1: the tensor-product element on quadrilaterals (DIM = 2; generic kernels only).
template <int DIM, int P, int TP = 0>
struct ElemDims {
  static constexpr int ND = TP ? (DIM == 3 ? (P + 1) * (P + 1) * (P + 1) : (P + 1) * (P + 1))
                               : (DIM == 1) ? (P + 1) : (DIM == 2) ? (P + 1) * (P + 2) / 2 : (P + 1) * (P + 2) * (P + 3) / 6;
  static constexpr int NF = (DIM == 1) ? 1 : (DIM == 2) ? (P + 1) : (TP ? (P + 1) * (P + 1) : (P + 1) * (P + 2) / 2);
  static constexpr int NFACES = TP ? 2 * DIM : DIM + 1;
  static constexpr int NCLS = TP ? 1 : (DIM == 1) ? 1 : (DIM == 2) ? 2 : 6;
};

// What the MFMA stage kernels need to know about the mesh, laid out for SCALAR loads: everything is uniform over
// a wave (an item is 16 cells of one class), the per-class part is contiguous (one batch of s_load's at an offset
// that depends only on the class), and the per-lane node tables are packed four byte entries to a word - lane
// group q = lane >> 4 extracts row 4 ks + q with one v_bfe.  The first MFMA kernels read these from a copy of
// MeshDev in LDS: about 45 DEPENDENT ds_read / s_waitcnt round trips per item in front of the first trace load
// (tools/wave_sim.py, profiles/r03/kernel_experiments.txt) - the lesson of the 2-D tile kernels' T2Const.
constexpr int MK_KSF = 4;     // facet k-steps at degree 4 (15 facet nodes)
constexpr int MK_KS = 9;      // volume k-steps at degree 4 (35 nodes)
#ifndef SG_GQ_FROM_DEGREE
#define SG_GQ_FROM_DEGREE 4   // G stages with the factorised volume term (mfma_stage_G<.., FACT = 1>) from this degree on, unless SEIGEN_HIP_GQ says otherwise
#endif
struct MfmaClassConst {
  int32_t nb_axis[4], nb_dir[4], nb_cls[4];
  int32_t slot_ord[4];        // ordinal of the matching facet among the neighbour cube's facets on that side
  uint32_t nbw[4][MK_KSF];    // [f][ks] byte q: neighbour ELEMENT node matching my facet node 4 ks + q (MeshDev::nb_node)
  uint32_t nfw[4][MK_KSF];    // same, as position in the neighbour's facet list (MeshDev::nb_fnode)
  int32_t nb_face[4];         // neighbour's local facet (MeshDev::nb_face)
};
struct MfmaConst {
  int32_t n[3];
  int32_t halo_per_cube;
  int32_t has_nbr[6];
  int32_t pad_[2];
  int64_t ncube, ncube_pad;
  uint32_t fw[4][MK_KSF];     // [f][ks] byte q: my element node of facet node 4 ks + q (MeshDev::fnode)
  MfmaClassConst cls[6];
};
MfmaConst mfma_const(const MeshDev& md_host);
// [variant 0: neighbour inside the block, 1: packed remote record, 2: domain boundary (own cell)][class][facet][lane group q][k-step]:
// offset (in values of the field type, before the component) of the neighbour's node that matches my facet node 4 ks + q -
// node * ncomp * 16 in a field, position in the facet list * 3 in a remote record; 3 * 6 * 4 * 4 * 4 entries
void mfma_trace_offsets(const MeshDev& md_host, int ncomp, std::vector<int32_t>& tab);
// [item = cell group * 6 + class][lane 0..15][facet 0..3] (see StageArgs::nbr_tab); (ncube_pad / 16) * 6 * 64 entries
void build_nbr_table(const MeshDev& md_host, std::vector<int32_t>& tab);

// scale * value rounded to double BEFORE anything is added to it: a separable source (weight x pattern) then gives
// bit for bit what a table of the products gives.  (HIP's __dmul_rn is a plain `x * y`, which hipcc contracts with
// a following add into one FMA; an instruction the optimiser cannot see into is not contracted.)
#if defined(__HIPCC__)
__host__ __device__ __forceinline__ double sg_mul_rounded(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double p;
  asm("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
  return p;
#else
  return a * b;   // host pass of the compiler only; never executed
#endif
}
#endif

// A source whose time slice is chosen ON THE DEVICE: a captured hipGraph freezes kernel arguments, so a launch that
// is to be replayed step after step reads the step index from a device word (bumped by a one-thread launch at the end
// of every step) instead of getting this step's slice and weight from the host (elastic.py:285-288 re-interpolates
// the source Expression before every step).
struct SrcStep {
  const int64_t* ctr;      // device word: index of the current step; null: the host chose slice and scale
  int64_t nsteps;          // steps the source covers (a step beyond them adds nothing)
  int64_t stride;          // doubles between two time slices of the value table (0: one slice for every step)
  const double* weights;   // separable source: weight per step (device), else null
  int32_t is_static;       // one slice that holds at every step
  int32_t pad_;
};

struct StageArgs {
  const double* in;    // stress for F, velocity for G           [cell][node][comp]
  double* out;         // result, or in-place target of a fused combine
  const double* aux;   // fused combine: second operand (uh1 / sh1)
  const double* uabs;  // F: velocity multiplied by the sponge (elastic.py:207-208)
  const double* ghost[6];  // packed remote facet traces of `in` per block side, or null
  const double* Dt;        // [dim][nd(b)][nd(a)]  transposed Mhat^-1 Shat_r
  const double* Lt;        // [nfaces][nf(b')][nd(a)] transposed facet lifts
  const MeshDev* md;       // device copy
  const MfmaConst* mk;     // MFMA path: device copy of mfma_const(md)
  const int32_t* ftab;     // MFMA path, F stages: trace offsets per (variant, class, facet, lane group) x 4 k-steps (mfma_trace_offsets)
  // MFMA path: where every cell finds its four facet neighbours, tabulated once per block (mfma_tables.cpp
  // build_nbr_table): nbr_tab[(item * 16 + lane) * 4 + f] = the neighbour's cell slot (cell group * 16 * 6 ... see
  // there), -1 on the domain boundary, -2 - s for slot s of the packed remote trace of that block side.  Replaces
  // the per-item recomputation (cube coordinates by division, neighbour class / axis / direction look-ups, bounds
  // tests): 256 B per item read with one 16-byte load per lane, against 76 KB of cell data.
  const int32_t* nbr_tab;
  int32_t all_active;      // the launch covers the whole block (no region boxes to test)
  int32_t tensor;          // generic path: quadrilateral cells (ElemDims<2, P, 1>) / hexahedra
  const double* fragV;     // MFMA path: volume operator fragments (mfma_tables.hpp), else null
  const double* fragL;     // MFMA path: facet-lift operator fragments
  const double* fragQ;     // MFMA path, G stages with the factorised volume term (mfma_stage_G<.., FACT = 1>): the Q tiles; fragV = the P_r tiles
  unsigned long long* dbg; // diagnostic builds (-DSG_STAMPS): per-phase cycle sums, else null
  const int32_t* sponge_slot;  // [cell] -> slot or -1 (null: no sponge)
  const double* sponge_B;      // [slot][nd(a)][nd(b)]
  const double* sponge_sigma;  // 2-D tile and 3-D MFMA kernels: [cell] 0 = no sponge; a value = sigma is CONSTANT over the cell, the sponge term is
                               // sigma u_i at the node itself and the cell has no slot; NaN = sigma varies: sponge_slot / sponge_B
  const void* sponge_pre;      // 3-D MFMA kernels: [slot][nd][dim] = B_slot u_abs of the cell, computed before the stage (launch_sponge_pre)
  const double* lam;           // per-cell (per_cell=1) or null
  const double* mu;
  double lam0, mu0;
  double c_self, c_aux, c_new;  // mode 1, F stages: out = c_self*out + c_aux*aux + c_new*rhs;  G stages: out = c_self*out + c_new*rhs
                                // (no second operand: the one fused G stage, S1, gets dt sh1 + dt^3/24 sh2 as ONE G, stages.cpp)
  // F, mode 1, per-cell density: rho2[cell][2] = {factor replacing c_self, factor multiplying c_aux and c_new}
  // (reference update: {rho, 1}; physical update: {1, 1/rho}); null: the scalars above
  const double* rho2;
  int32_t mode;                 // 0: out = rhs; 1: fused (see c_self ..); 2 (F stages): fused without the self term, out = c_aux*aux + c_new*rhs
  int32_t per_cell;
  int32_t box_o[3], box_n[3];   // region of cubes covered by this launch (generic kernel: one box per launch)
  // MFMA / lane kernels: one launch covers up to SG_MAX_BOXES disjoint boxes (a boundary shell + half an interior); `spread` deals
  // the items round-robin over all waves instead of one contiguous range per XCD, because a shell's
  // active items are a few contiguous runs that would otherwise land on one XCD
  int32_t nbox;
  int32_t boxes_o[SG_MAX_BOXES][3], boxes_n[SG_MAX_BOXES][3];
  int32_t spread;
  const int32_t* item_list;  // a region of a split stage: its active items (cell group * ncls + class), else null
  int32_t nlist;
  int32_t sym;                  // MFMA path: stress fields are symmetric, touch only the i <= j lines
  int32_t grid_blocks;          // MFMA path: size of the persistent grid (a multiple of 8)
  int32_t f32;                  // MFMA path: fields, halo buffers and operator tables are float (sg_config.dtype = 1)
  // MFMA path, whole-block launches: items are dealt to the XCDs in chunks of this many consecutive items, round-robin
  // (0: one contiguous eighth of the items per XCD).  With a chunk of 1/8 of a z-layer all XCDs sweep the block layer by
  // layer together, so the z-neighbour traces and the own rows of the next layer meet in the Infinity Cache.
  int32_t order_chunk;
  int32_t nitems;               // MFMA path: items of this launch (the item list's length, or cell groups x classes)
  // Host side only (sg_stage_kernel_name): when set, the launch functions below launch NOTHING and write the name of the
  // kernel instantiation they would have launched - taken from the same function pointer the launch uses (SG_LAUNCH)
  std::string* name_out;
  // 2-D tile path, G stages: the sparse nodal source (elastic.py:217-218) added inside the stage kernel instead of
  // by a launch of its own.  src_slot[item] = slot of an item (16 cells of one class) that holds source nodes, or -1;
  // src_idx[slot][node][cell] = row of that node in this step's value table src_vals[row][dim*dim], or -1.
  const int32_t* src_slot;
  const int32_t* src_idx;
  const double* src_vals;
  double src_scale;          // factor on src_vals (a separable source's weight of this step, else 1)
  double src_coef;           // ... and on the scaled, rounded value as it enters the right-hand side (stage S1: dt + dt^3/24, else 1)
  SrcStep src_step;          // graph replay: slice and weight from the device-side step counter (ctr != null)
  int32_t src_bump;          // 2-D tile path, first stage of a step (an F stage): bump that counter (one thread)
};

// Name of a kernel as rocprofv3 prints it ("void sg::mfma_stage_G<double, 4, 0, 1, 1>(sg::StageArgs)"), from the host-side
// handle of the instantiation (kernels.hip: dladdr + demangling) - the library names its kernels itself, nobody re-derives
// template arguments from switches.
std::string kernel_name_of(const void* host_fn);
#if defined(__HIPCC__)
// every stage launch goes through here: launch, or (StageArgs::name_out) only name what would be launched
#define SG_LAUNCH(kernel, grid, block, stream, A, ...)                                           \
  do {                                                                                           \
    if ((A).name_out)                                                                            \
      *(A).name_out = ::sg::kernel_name_of(reinterpret_cast<const void*>(&kernel));              \
    else                                                                                         \
      hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                           \
  } while (0)
#endif

// kind: 0 = F (velocity RHS), 1 = G (stress RHS)
int launch_stage(int kind, int dim, int P, const StageArgs& a, void* stream);

// MFMA path (3-D, degree >= 3; fields in the gw = 16 interleaved layout)
bool mfma_supported(int dim, int P);
int mfma_blocks_per_cu(int P, int f32);
int launch_stage_mfma(int kind, int P, const StageArgs& a, void* stream);

// lane-per-cell path (1-D / 2-D; fields in the gw = 64 interleaved layout; a.Dt = E[r][a][b],
// a.Lt = L[f][a][b'] row-major)
bool lane_supported(int dim, int P);
// measured crossover of the (sum-factorised) generic and the lane kernels on hexahedra (tools/experiments/
// hex_crossover.py, profiles/r04/hexahedra.txt): between 24^3 and 32^3 cubes at both degrees
#ifndef SG_HEX_LANE_MIN_CELLS
#define SG_HEX_LANE_MIN_CELLS(degree) 24000
#endif
bool lane_supported_hex(int dim, int P);   // a.tensor: hexahedra (sum-factorised; a.Dt = {D1, lift1})
// hexahedra DQ_3 / DQ_4 with the lines of a cube in registers and the x lines on the matrix pipe (kernels_hexm.hip; fields
// in the gw = 16 interleaved layout; a.Dt = hexm_table(): line operators E_k, trace lifts, x-pass A operands)
bool hexm_supported(int dim, int P);
int hexm_blocks_per_cu(int P);
std::vector<double> hexm_table(int P, const double* D1, const double* lift1, const MeshDev& md_host);
int launch_stage_hexm(int kind, int P, const StageArgs& a, long nitems, void* stream);
int launch_stage_lane(int kind, int dim, int P, const StageArgs& a, long nitems, void* stream);

// 2-D MFMA tile path (P1..P4; fields in the gw = 16 interleaved layout; a.fragV / a.fragL = tile2d_frags_*).
// T2Const is the part of MeshDev these kernels use, passed by value in the kernarg segment so that no load
// depends on another one (kernels_tile2d.hip): sizes, class constants, neighbour rules, and the node
// permutations packed four byte entries to a word (entry q of word [ks] = row 4 ks + q).
// (facet arrays hold four entries: quadrilateral cells have four facets, triangles use the first three)
struct T2Class {            // everything that depends on the triangle class, contiguous: one batch of scalar loads
  double Jinv[2][2], cn[4][2];
  int32_t nb_axis[4], nb_dir[4], nb_cls[4];
  int32_t slot_ord[4];      // ordinal of the matching facet among the neighbour cube's facets on that side
  uint32_t tfw[4][2];       // neighbour ELEMENT node matching my facet node (MeshDev::nb_node)
  uint32_t tgw[4][2];       // same, as position in the neighbour's facet list (MeshDev::nb_fnode)
};
struct T2Const {
  int32_t n0, n1, ncube, ngroups;
  int32_t has_nbr[4];
  int32_t halo_per_cube;
  int32_t gpr;              // groups of 16 squares that cover at least one row of the block (+1: a group may straddle rows)
  double inv_n0;            // 1 / n0
  uint32_t tpw[4][2];       // my element node of facet node (MeshDev::fnode)
  T2Class cls[2];
};
T2Const tile2d_const(const MeshDev& md_host);
bool tile2d_supported(int dim, int P);
bool tile2d_supported_quad(int P);   // quadrilateral cells (StageArgs::tensor): DQ_1..4
int launch_stage_tile2d(int kind, int P, const StageArgs& a, const T2Const& c, long nitems, void* stream);

// host layout [cell][node][comp] <-> device layout (MeshDev::gw) for `ncells` cells from `cell0`
// dir = 0: staging -> field, 1: field -> staging
// sym = 1 (stress field in symmetric mode): a download mirrors the lower triangle from the upper;
// flag (device int, may be null): set to 1 by an upload whose tensors are not exactly symmetric
// f32 = 1: the field is float (sg_config.dtype = 1); the staging side is double either way
int launch_layout(const MeshDev& md_host, int ncomp, int dir, void* field, double* staging, int64_t cell0,
                  int64_t ncells, int sym, int* flag, int f32, void* stream);
// copy the upper triangle over the lower one in a whole stress field (leaving symmetric mode)
int launch_mirror(const MeshDev& md_host, void* field, int f32, void* stream);

// facet traces of a field on `nside` block sides -> one packed device buffer per side, one launch
// (sym = 1: a stress field stored in symmetric mode; lower-triangle values come from their mirrors)
int launch_pack(const MeshDev* md_dev, const MeshDev& md_host, const void* field, int ncomp, int nside,
                const int* sides, void* const* outs, int sym, int f32, void* stream);

// field[off[k] + c*gw] += coef * values[k][c] at the sparse source nodes (off = device offset of comp 0)
// (values are scaled by `scale` and rounded first: a separable source's weight of this step)
int launch_source(void* field, int ncomp, int gw, int64_t nnz, const int64_t* offs, const double* values, double coef,
                  double scale, const SrcStep& ss, int f32, void* stream);
// sp[slot][a][i] = sum_b B[slot][a][b] u_abs[cell of slot][b][i] for the cells with a sponge matrix of their own (one block per
// cell), queued before the F stage's launches (StageArgs::sponge_pre)
// (slots: the list of slots to work on, or null = slots 0 .. nslots-1)
// (lines: sp in the layout of the fields, slot = item * gw + w - the 3-D MFMA family)
int launch_sponge_pre(const void* uabs, const double* B, const int32_t* cells, const int32_t* mats, const int32_t* slots, void* sp,
                      int32_t nslots, int nd, int dim, int ncls, int gw, int lines, int f32, void* stream);
// the same sp[slot][a][i] for the cells whose sigma is affine in the reference coordinates: B_e = s_0 I + sum_k s_k X_k with the
// element-constant X[k][a][j] (row a in ELL form: column col[a][j], or j itself where col is null), coef[slot][dim + 1] = s;
// items[n] = (cube group) * ncls + class of the n-th item that holds such a cell, item_slots[n][gw] its cells' slots (-1: none)
int launch_sponge_pre_affine(const void* uabs, const double* X, const int32_t* col, int W, const int32_t* items, const int32_t* item_slots,
                             const double* coef, void* sp, int32_t nitems, int nd, int dim, int gw, int lines, int f32, void* stream);
size_t sponge_pre_affine_lds(int W, int has_col, int nd, int dim, int gw);
int prepare_sponge_pre_affine(int dim, int f32, size_t lds);   // once, outside any stream capture: allow that much dynamic LDS
// ... on the matrix pipe for the 3-D MFMA family in double (kernels_mfma.hip): fragX = mfma_frags_dense of the three X_k
int prepare_sponge_affine_mfma(int P);   // once, at set-up: the occupancy query behind the launch's grid size
int launch_sponge_affine_mfma(int P, const void* uabs, const double* fragX, const int32_t* items, const int32_t* item_slots,
                              const double* coef, void* sp, int32_t nitems, void* stream);
// the device-side step counter of SrcStep: *ctr = value (add = 0) or *ctr += value (add = 1), one thread
int launch_step_counter(int64_t* ctr, int64_t value, int add, void* stream);

}  // namespace sg
