// Lane-per-cell stage kernels for small elements (1-D and 2-D P1..P4, 3-D P1/P2; gfx950, FP64).
//
// With at most 15 nodes per cell the whole cell fits in one lane's registers, so the wavefront
// is used as 64 independent cells: lane l owns cell (cube 64g + l, class k) of a group of 64
// consecutive cubes.  Fields use the gw = 64 interleaved layout (mesh_tables.hpp): the 64 values
// of one (node, component) are one contiguous 512-byte run, so every load and store of the
// kernel is a full-width coalesced wave access, all of a cell's operands are requested up front
// (maximum memory-level parallelism), and there is no LDS traffic for cell data at all.
//
// The reference operators are the same for all 64 lanes: they are read through the scalar
// cache (s_load) and enter the FMAs as scalar operands.  E_r = D_r - C_r folds the own-trace
// half of the central flux into the volume operator (mfma_tables.cpp), so the facet lifts carry
// only the neighbour's half.  The path is HBM-bound: ~0.1 kflop per 64 bytes at 2-D P2.
//
//   F:  uh_i  = -sum_r E_r (Jinv_rj T_ij) + sum_f L_f [ w_f (c n)_f,j T(nbr)_ij ]  - sponge
//   G:  W_ik  = -Jinv_rk (E_r u_i) + sum_f (c n)_f,k L_f [ 1/2 u(nbr)_i ];  sh = lam tr(W) I + mu (W + W^T)
// (seigen/elastic.py:204-219 with the element mass inverse of :358-367 folded in.)
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sg {

template <int DIM, int P>
struct LG : ElemDims<DIM, P> {};

struct LaneCell {
  long c;
  int cc[3];
  bool valid, active;
};

__device__ __forceinline__ LaneCell lane_cell(const MeshDev* md, const StageArgs& A, long g, int lane) {
  LaneCell L;
  L.c = g * 64 + lane;
  L.valid = L.c < md->ncube;
  long cl = L.valid ? L.c : 0;
  L.cc[0] = (int)(cl % md->n[0]);
  long t = cl / md->n[0];
  L.cc[1] = (int)(t % md->n[1]);
  L.cc[2] = (int)(t / md->n[1]);
  bool in = false;
  for (int bx = 0; bx < A.nbox; ++bx) {
    bool ib = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) ib = ib && (L.cc[a] >= A.boxes_o[bx][a]) && (L.cc[a] < A.boxes_o[bx][a] + A.boxes_n[bx][a]);
    in = in || ib;
  }
  L.active = L.valid && in;
  return L;
}

struct LaneNbr {
  const double* p;
  int cstride;    // 64 in a field, 1 in a packed ghost buffer
  bool ghost, physical;
};

template <int DIM, int ND, int NF, int NC, int NCLS>
__device__ __forceinline__ LaneNbr lane_nbr(const MeshDev* md, const StageArgs& A, const LaneCell& L, long g, int k, int f,
                                            int lane, const double* own) {
  LaneNbr R;
  R.p = own;
  R.cstride = 64;
  R.ghost = false;
  R.physical = false;
  const int axis = md->nb_axis[k][f];
  const int kn = md->nb_cls[k][f];
  if (axis < 0) {
    R.p = A.in + ((g * NCLS + kn) * (long)ND) * NC * 64 + lane;
    return R;
  }
  const int dir = md->nb_dir[k][f];
  const int cn = L.cc[axis] + dir;
  const bool inside = L.valid && cn >= 0 && cn < md->n[axis];
  const long stride = (axis == 0) ? 1 : (axis == 1) ? md->n[0] : (long)md->n[0] * md->n[1];
  const long nc = inside ? L.c + dir * stride : L.c;
  const double* pin = A.in + (((nc >> 6) * NCLS + (inside ? kn : k)) * (long)ND) * NC * 64 + (nc & 63);
  const int side = 2 * axis + (dir > 0 ? 1 : 0);
  if (!inside && L.valid && md->has_nbr[side]) {
    long c2 = (axis == 0) ? (L.cc[1] + (long)md->n[1] * L.cc[2])
                          : (axis == 1) ? (L.cc[0] + (long)md->n[0] * L.cc[2]) : (L.cc[0] + (long)md->n[0] * L.cc[1]);
    long slot = c2 * md->halo_per_cube + md->face_ord[kn][md->nb_face[k][f]];
    R.p = A.ghost[side] + slot * NF * DIM;  // packed trace: DIM comps per facet node (velocity, or T_i,axis)
    R.cstride = 1;
    R.ghost = true;
    return R;
  }
  R.p = pin;
  R.physical = !inside;
  return R;
}

// SYM = 1: symmetric-stress mode (kernels_mfma.hip): only the i <= j lines of a stress field are
// read or written; component (i, j) with i > j is taken from its mirror (j, i).
template <int DIM, int SYM>
__device__ __forceinline__ constexpr int cidx(int i, int j) {
  return (SYM && i > j) ? j * DIM + i : i * DIM + j;
}

template <int DIM, int P, int KIND, int MODE, int SYM>
__global__ __launch_bounds__(256) void lane_stage(StageArgs A) {
  using G = LG<DIM, P>;
  constexpr int ND = G::ND, NF = G::NF, NFACES = G::NFACES, NCLS = G::NCLS;
  constexpr int NC = (KIND == 0) ? DIM * DIM : DIM;  // input components
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const MeshDev* md = A.md;                 // uniform: scalar loads
  // operator tables are read-only and indexed uniformly: constant address space => s_load, and the
  // values enter the FMAs as scalar operands (a plain global pointer out of the by-value argument
  // struct carries no noalias/readonly information and hipcc would use per-lane vector loads)
  typedef __attribute__((address_space(4))) const double cdouble;
  const cdouble* Eop = (const cdouble*)(unsigned long long)A.Dt;  // [r][a][b]  E_r (row-major; lane path)
  const cdouble* Lop = (const cdouble*)(unsigned long long)A.Lt;  // [f][a][b']
  const double* __restrict__ in = A.in;
  const double* __restrict__ aux = A.aux;
  double* __restrict__ out = A.out;

  const long ngroups = md->ncube_pad >> 6;
  const bool listed = A.item_list != nullptr;  // a region of a split stage: its active items (stages.cpp)
  const long nitems = listed ? (long)A.nlist : ngroups * NCLS;
  // one contiguous item range per XCD label (blocks with equal blockIdx % 8 share an L2)
  const long nblk = gridDim.x, xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const long blocks_here = (nblk - xcd + 7) / 8, ipx = (nitems + 7) / 8;
  const long lo = xcd * ipx, hi = (xcd + 1) * ipx < nitems ? (xcd + 1) * ipx : nitems;

  const long i0 = A.spread ? (long)blockIdx.x * 4 + wave : lo + slot * 4 + wave;
  const long i1 = A.spread ? nitems : hi;
  const long istep = A.spread ? (long)gridDim.x * 4 : blocks_here * 4;
  for (long it = i0; it < i1; it += istep) {
    const long item = listed ? (long)A.item_list[it] : it;
    const long g = item / NCLS;
    const int k = (int)(item - g * NCLS);
    const LaneCell L = lane_cell(md, A, g, lane);
    if (!__any(L.active)) continue;
    const double* own = in + ((g * NCLS + k) * (long)ND) * NC * 64 + lane;

    double Jv[DIM][DIM], cnv[NFACES][DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int j = 0; j < DIM; ++j) Jv[r][j] = md->Jinv[k][r][j];
#pragma unroll
    for (int f = 0; f < NFACES; ++f)
#pragma unroll
      for (int j = 0; j < DIM; ++j) cnv[f][j] = md->cn[k][f][j];

    // where each facet's neighbour trace lives (a boundary lane's neighbour is its own cell)
    const double* np[NFACES];
    int nst[NFACES];
    double wf[NFACES];
    int nnode[NFACES][NF];
    bool gh[NFACES];   // F: this lane reads the facet from a packed remote trace (T_i,axis only)
#pragma unroll
    for (int f = 0; f < NFACES; ++f) {
      const LaneNbr R = lane_nbr<DIM, ND, NF, NC, NCLS>(md, A, L, g, k, f, lane, own);
      np[f] = R.p;
      nst[f] = R.cstride;
      // F: +1/2 neighbour flux inside, -1/2 own flux on the boundary (cancels the folded half: T.n = 0);
      // G: 1/2 of the neighbour, or the missing 1/2 of the own trace on the boundary
      wf[f] = (KIND == 0 && R.physical) ? -0.5 : 0.5;
      gh[f] = R.ghost;
#pragma unroll
      for (int bp = 0; bp < NF; ++bp) {
        const int on = md->fnode[f][bp];
        nnode[f][bp] = R.ghost ? md->nb_fnode[k][f][bp] * DIM : (R.physical ? on : md->nb_node[k][f][bp]) * NC * R.cstride;
      }
    }
    const long e = (L.valid ? L.c : 0) * NCLS + k;

    if (KIND == 0) {
      // ---- F, one velocity component i at a time: only row i of the stress is needed, which
      //      halves (2-D) the live registers and doubles the resident waves
      // sponge of this lane's cell: a sigma that is one value on all its nodes is applied as sigma u_abs at the node, a varying
      // one through B_e u_abs computed before the stage by a launch of its own (StageArgs::sponge_sigma / sponge_pre): one
      // load per value either way - another base, stride and factor - instead of an nd x nd matrix per lane
      double sig = 0.0;
      int sslot = -1;
      if (A.sponge_sigma != nullptr && L.active) {
        sig = A.sponge_sigma[e];
        if (sig != sig) sslot = A.sponge_slot[e];
      }
      const long ubase = ((g * NCLS + k) * (long)ND) * DIM * 64 + lane;
      double cs = A.c_self, ca = A.c_aux, cn = A.c_new;
      const bool self = A.c_self != 0.0 || A.rho2 != nullptr;     // uniform
      if (MODE == 1 && A.rho2 != nullptr) {  // per-cell density (kernels.hpp)
        cs = A.rho2[2 * e];
        ca *= A.rho2[2 * e + 1];
        cn *= A.rho2[2 * e + 1];
      }
      // in the fused stages u_abs IS one of the combine's operands (`out` in stage U1, `aux` in stage UTEMP): a constant
      // sigma then changes that operand's coefficient - no load, no arithmetic of its own
      if (MODE == 1 && sslot < 0 && sig != 0.0) {
        if (self && A.uabs == A.out) {
          cs -= cn * sig;
          sig = 0.0;
        } else if (A.uabs == A.aux) {
          ca -= cn * sig;
          sig = 0.0;
        }
      }
      const bool any_sponge = __any(sig != 0.0);     // (NaN != 0: the lanes with a matrix count)
      const double* sp_pb = sslot >= 0 ? reinterpret_cast<const double*>(A.sponge_pre) + (long)sslot * ND * DIM : A.uabs + ubase;
      const int sp_es = sslot >= 0 ? 1 : 64;
      const double sp_sc = sslot >= 0 ? 1.0 : sig;
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        double q[ND][DIM];  // T_ij (j = 0..DIM-1) at every node, all requested at once
#pragma unroll
        for (int b = 0; b < ND; ++b)
#pragma unroll
          for (int j = 0; j < DIM; ++j) q[b][j] = own[(b * NC + cidx<DIM, SYM>(i, j)) * 64];
        double fl[NFACES][NF];  // w_f (c n)_f,j T(nbr)_ij
#pragma unroll
        for (int f = 0; f < NFACES; ++f)
#pragma unroll
          for (int bp = 0; bp < NF; ++bp) {
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
              // a packed remote trace holds g_i = T_i,axis only: such a lane reads g_i for every j (one
              // unconditional load, selected offset); the columns j != axis meet cn_j = 0 there
              sacc += cnv[f][j] * np[f][nnode[f][bp] + (gh[f] ? i : cidx<DIM, SYM>(i, j) * 64)];
            }
            fl[f][bp] = wf[f] * sacc;
          }
        // sponge operand and in-place combine operands, read before this cell's u_i is written
        double ua[ND];
        if (any_sponge) {
#pragma unroll
          for (int b = 0; b < ND; ++b) ua[b] = sig != 0.0 ? sp_sc * sp_pb[(b * DIM + i) * sp_es] : 0.0;
        }
        double po[ND], pa[ND];
        if (MODE == 1) {     // a.mode 2 (UTEMP: no self term) arrives here with c_self = 0 and `out` is not read
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            po[a] = self ? out[ubase + (a * DIM + i) * 64] : 0.0;
            pa[a] = aux[ubase + (a * DIM + i) * 64];
          }
        }
        // T~_ir = Jinv_rj T_ij
#pragma unroll
        for (int b = 0; b < ND; ++b) {
          double t[DIM];
#pragma unroll
          for (int r = 0; r < DIM; ++r) {
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < DIM; ++j) sacc += Jv[r][j] * q[b][j];
            t[r] = sacc;
          }
#pragma unroll
          for (int r = 0; r < DIM; ++r) q[b][r] = t[r];
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int b = 0; b < ND; ++b) acc -= Eop[(r * ND + a) * ND + b] * q[b][r];
#pragma unroll
          for (int f = 0; f < NFACES; ++f)
#pragma unroll
            for (int bp = 0; bp < NF; ++bp) acc += Lop[(f * ND + a) * NF + bp] * fl[f][bp];
          if (any_sponge) acc -= ua[a];
          if (MODE == 1) acc = cs * po[a] + ca * pa[a] + cn * acc;
          if (L.active) out[ubase + (a * DIM + i) * 64] = acc;
        }
      }
    } else {
      // ---- G: the cell's velocity and the halved neighbour traces, all requested at once
      double q[ND][DIM];
#pragma unroll
      for (int b = 0; b < ND; ++b)
#pragma unroll
        for (int c = 0; c < DIM; ++c) q[b][c] = own[(b * NC + c) * 64];
      double fl[NFACES][NF][DIM];
#pragma unroll
      for (int f = 0; f < NFACES; ++f)
#pragma unroll
        for (int bp = 0; bp < NF; ++bp)
#pragma unroll
          for (int i = 0; i < DIM; ++i) fl[f][bp][i] = wf[f] * np[f][nnode[f][bp] + i * nst[f]];
      const double lam = A.per_cell ? A.lam[e] : A.lam0;
      const double mu = A.per_cell ? A.mu[e] : A.mu0;
      const long sbase = ((g * NCLS + k) * (long)ND) * DIM * DIM * 64 + lane;
      double po[DIM * DIM];   // in-place combine operand, one node ahead (a fused G stage: out = c_self out + c_new rhs)
      // Fused stage: loads and stores share one counter and complete out of order with each
      // other, so a wait for a load with stores in flight waits for every store's acknowledgement
      // (kernels_mfma.hip).  Where the cell's results fit in registers they are all kept until the
      // last old value has been read, and stored together at the end.
      constexpr bool DEFER = (MODE == 1) && (ND * DIM * DIM <= 24);
      double res[DEFER ? ND : 1][DIM * DIM];
      if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < DIM * DIM; ++c)
          if (!SYM || (c / DIM) <= (c % DIM)) po[c] = out[sbase + c * 64];
      }
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        double R[DIM][DIM];  // (E_r u_i)_a
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int r = 0; r < DIM; ++r) R[i][r] = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
          for (int b = 0; b < ND; ++b) {
            const double ev = Eop[(r * ND + a) * ND + b];
#pragma unroll
            for (int i = 0; i < DIM; ++i) R[i][r] += ev * q[b][i];
          }
        double W[DIM][DIM];
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int kk = 0; kk < DIM; ++kk) {
            double sacc = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) sacc -= Jv[r][kk] * R[i][r];
            W[i][kk] = sacc;
          }
#pragma unroll
        for (int f = 0; f < NFACES; ++f) {
          double lu[DIM];
#pragma unroll
          for (int i = 0; i < DIM; ++i) lu[i] = 0.0;
#pragma unroll
          for (int bp = 0; bp < NF; ++bp) {
            const double lv = Lop[(f * ND + a) * NF + bp];
#pragma unroll
            for (int i = 0; i < DIM; ++i) lu[i] += lv * fl[f][bp][i];
          }
#pragma unroll
          for (int i = 0; i < DIM; ++i)
#pragma unroll
            for (int kk = 0; kk < DIM; ++kk) W[i][kk] += cnv[f][kk] * lu[i];
        }
        double trc = 0.0;
#pragma unroll
        for (int kk = 0; kk < DIM; ++kk) trc += W[kk][kk];
        double v[DIM * DIM];
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int j = 0; j < DIM; ++j) {
            v[i * DIM + j] = mu * (W[i][j] + W[j][i]) + ((i == j) ? lam * trc : 0.0);
            if (MODE == 1 && (!SYM || i <= j))
              v[i * DIM + j] = A.c_self * po[i * DIM + j] + A.c_new * v[i * DIM + j];
          }
        if (MODE == 1 && a + 1 < ND) {
#pragma unroll
          for (int c = 0; c < DIM * DIM; ++c)
            if (!SYM || (c / DIM) <= (c % DIM)) po[c] = out[sbase + ((a + 1) * DIM * DIM + c) * 64];
        }
        if (DEFER) {
#pragma unroll
          for (int c = 0; c < DIM * DIM; ++c) res[DEFER ? a : 0][c] = v[c];
        } else if (L.active) {
#pragma unroll
          for (int c = 0; c < DIM * DIM; ++c)
            if (!SYM || (c / DIM) <= (c % DIM)) out[sbase + (a * DIM * DIM + c) * 64] = v[c];
        }
      }
      if (DEFER && L.active) {
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
          for (int c = 0; c < DIM * DIM; ++c)
            if (!SYM || (c / DIM) <= (c % DIM)) out[sbase + (a * DIM * DIM + c) * 64] = res[DEFER ? a : 0][c];
      }
    }
  }
}

// ---- hexahedral cells: sum-factorised lane-per-cell kernels (DQ_1, DQ_2) --------------------------------------------------
// On the cubes of a structured block the tensor-product element factorises completely:
//   D_r = I x .. x D1 x .. x I   (D1 = M1^-1 S1 of the interval element, along axis r only),
//   L_f (facet 2m + s) = lift1[s][a_m] on the facet node with the same transverse indices,
//   Jinv = diag(1 / h), (c n)_f has the single component -+ 1 / h_m,
// so a node's right-hand side takes (P + 1) FMAs per direction and one per facet instead of a dense row of nd + 6 nf
// entries: per 64 bytes moved the stage needs ~15 FMAs, and the path is bound by the memory system alone (measured:
// 5.5 - 6.1 TB/s of fabric traffic, profiles/r04/hexahedra.txt).  Lane l owns cube 64 g + l as in lane_stage; the fields
// use the same gw = 64 layout.  F works one stress component (i, j) at a time: its 27 nodal values, the two opposite
// facet traces across axis j and one set of 27 accumulators are live together (two waves per SIMD).  G reads every velocity
// component twice (two sweeps, see there) with one wave per SIMD at DQ_2.
//   a.Dt = { D1 [P+1][P+1] row-major, lift1 [2][P+1] }  (api.cpp, checked there against the full D_r, L_f)
#ifndef SG_HEX_WAVES
#define SG_HEX_WAVES 2
#endif
#ifndef SG_HEX_HOLD
#define SG_HEX_HOLD 1
#endif
#ifndef SG_HEX_WAVES_G2
#define SG_HEX_WAVES_G2 1      // the G stages of DQ_2: three sets of nd results and the next operands live together
#endif
#define SG_HEX_WPE(P, KIND) (((P) == 2 && (KIND) == 1) ? SG_HEX_WAVES_G2 : SG_HEX_WAVES)
template <int P, int KIND, int MODE, int SYM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SG_HEX_WPE(P, KIND), SG_HEX_WPE(P, KIND)))) void hex_stage(StageArgs A) {
  constexpr int DIM = 3, N1 = P + 1, ND = N1 * N1 * N1, NF = N1 * N1, NFACES = 6;
  constexpr int NC = (KIND == 0) ? 9 : 3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const MeshDev* md = A.md;
  typedef __attribute__((address_space(4))) const double cdouble;
  const cdouble* D1 = (const cdouble*)(unsigned long long)A.Dt;
  const cdouble* lift1 = D1 + N1 * N1;
  const double* __restrict__ in = A.in;
  const double* __restrict__ aux = A.aux;
  double* __restrict__ out = A.out;

  const long ngroups = md->ncube_pad >> 6;
  const bool listed = A.item_list != nullptr;
  const long nitems = listed ? (long)A.nlist : ngroups;
  const long nblk = gridDim.x, xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const long blocks_here = (nblk - xcd + 7) / 8, ipx = (nitems + 7) / 8;
  const long lo = xcd * ipx, hi = (xcd + 1) * ipx < nitems ? (xcd + 1) * ipx : nitems;
  const long i0 = A.spread ? (long)blockIdx.x * 4 + wave : lo + slot * 4 + wave;
  const long i1 = A.spread ? nitems : hi;
  const long istep = A.spread ? (long)gridDim.x * 4 : blocks_here * 4;

  // node of a line: axis m, transverse indices (t0, t1) with the lower axis first - also the facet-node index bp
  auto node_of = [](int m, int t0, int t1, int am) -> int {
    return m == 0 ? am + N1 * (t0 + N1 * t1) : m == 1 ? t0 + N1 * (am + N1 * t1) : t0 + N1 * (t1 + N1 * am);
  };

  for (long it = i0; it < i1; it += istep) {
    const long g = listed ? (long)A.item_list[it] : it;
    const LaneCell L = lane_cell(md, A, g, lane);
    if (!__any(L.active)) continue;
    const double* own = in + (g * (long)ND) * NC * 64 + lane;
    const long e = L.valid ? L.c : 0;

    double ih[DIM], cnf[NFACES];
#pragma unroll
    for (int r = 0; r < DIM; ++r) ih[r] = md->Jinv[0][r][r];
#pragma unroll
    for (int f = 0; f < NFACES; ++f) cnf[f] = md->cn[0][f][f / 2];

    const double* np[NFACES];
    bool gh[NFACES], ph[NFACES];
#pragma unroll
    for (int f = 0; f < NFACES; ++f) {
      const LaneNbr R = lane_nbr<DIM, ND, NF, NC, 1>(md, A, L, g, 0, f, lane, own);
      np[f] = R.p;
      gh[f] = R.ghost;
      ph[f] = R.physical;
    }
    // offset of facet node bp = t0 + N1 t1 of facet f, component c, in that facet's source: a packed remote trace
    // ([slot][bp][DIM]), the own cell (domain boundary) or the neighbour cell, whose matching node is the one across
    // the cube (the tables MeshDev::fnode / nb_node / nb_fnode say the same: checked at create, api.cpp)
    auto noff = [&](int f, int t0, int t1, int cfield, int cghost) -> int {
      const int ngh = (t0 + N1 * t1) * DIM + cghost;
      const int nph = (node_of(f / 2, t0, t1, (f & 1) ? P : 0) * NC + cfield) * 64;
      const int nin = (node_of(f / 2, t0, t1, (f & 1) ? 0 : P) * NC + cfield) * 64;
      return gh[f] ? ngh : (ph[f] ? nph : nin);
    };

    if (KIND == 0) {
      // sponge of this lane's cell: a sigma that is one value on all its nodes is applied as sigma u_abs at the node, a varying
      // one through B_e u_abs computed before the stage by a launch of its own (StageArgs::sponge_sigma / sponge_pre): one
      // load per value either way - another base, stride and factor - instead of an nd x nd matrix per lane
      double sig = 0.0;
      int sslot = -1;
      if (A.sponge_sigma != nullptr && L.active) {
        sig = A.sponge_sigma[e];
        if (sig != sig) sslot = A.sponge_slot[e];
      }
      const long ubase = (g * (long)ND) * DIM * 64 + lane;
      double cs = A.c_self, ca = A.c_aux, cn = A.c_new;
      const bool self = A.c_self != 0.0 || A.rho2 != nullptr;     // uniform
      if (MODE == 1 && A.rho2 != nullptr) {
        cs = A.rho2[2 * e];
        ca *= A.rho2[2 * e + 1];
        cn *= A.rho2[2 * e + 1];
      }
      // in the fused stages u_abs IS one of the combine's operands (`out` in stage U1, `aux` in stage UTEMP): a constant
      // sigma then changes that operand's coefficient - no load, no arithmetic of its own
      if (MODE == 1 && sslot < 0 && sig != 0.0) {
        if (self && A.uabs == A.out) {
          cs -= cn * sig;
          sig = 0.0;
        } else if (A.uabs == A.aux) {
          ca -= cn * sig;
          sig = 0.0;
        }
      }
      const bool any_sponge = __any(sig != 0.0);     // (NaN != 0: the lanes with a matrix count)
      const double* sp_pb = sslot >= 0 ? reinterpret_cast<const double*>(A.sponge_pre) + (long)sslot * ND * DIM : A.uabs + ubase;
      const int sp_es = sslot >= 0 ? 1 : 64;
      const double sp_sc = sslot >= 0 ? 1.0 : sig;
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        double acc[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) acc[a] = 0.0;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
          // T_ij at the cell's nodes and on the two facets across axis j
          double q[ND], tn[2][NF];
#pragma unroll
          for (int b = 0; b < ND; ++b) q[b] = own[(b * NC + cidx<DIM, SYM>(i, j)) * 64];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int bp = 0; bp < NF; ++bp)
              tn[s2][bp] = np[2 * j + s2][noff(2 * j + s2, bp % N1, bp / N1, cidx<DIM, SYM>(i, j), i)];
          // {T} = 1/2 (own + neighbour) on interior facets, no facet term on the domain boundary (elastic.py:204-206)
          const double w0 = ph[2 * j] ? 0.0 : 0.5 * cnf[2 * j], w1 = ph[2 * j + 1] ? 0.0 : 0.5 * cnf[2 * j + 1];
#pragma unroll
          for (int t1 = 0; t1 < N1; ++t1)
#pragma unroll
            for (int t0 = 0; t0 < N1; ++t0) {
              const int bp = t0 + N1 * t1;
              const double f0 = w0 * (q[node_of(j, t0, t1, 0)] + tn[0][bp]);
              const double f1 = w1 * (q[node_of(j, t0, t1, P)] + tn[1][bp]);
#pragma unroll
              for (int am = 0; am < N1; ++am) {
                double v = 0.0;
#pragma unroll
                for (int m = 0; m < N1; ++m) v += D1[am * N1 + m] * q[node_of(j, t0, t1, m)];
                acc[node_of(j, t0, t1, am)] += lift1[am] * f0 + lift1[N1 + am] * f1 - ih[j] * v;
              }
            }
        }
        if (any_sponge) {
          if (sig != 0.0) {
            double ua[ND];
#pragma unroll
            for (int b = 0; b < ND; ++b) ua[b] = sp_pb[(b * DIM + i) * sp_es];
#pragma unroll
            for (int a = 0; a < ND; ++a) acc[a] -= sp_sc * ua[a];
          }
        }
        if (MODE == 1) {
          double po[ND], pa[ND];     // a.mode 2 (UTEMP: no self term) arrives here with c_self = 0 and `out` is not read
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            po[a] = self ? out[ubase + (a * DIM + i) * 64] : 0.0;
            pa[a] = aux[ubase + (a * DIM + i) * 64];
          }
#pragma unroll
          for (int a = 0; a < ND; ++a) acc[a] = cs * po[a] + ca * pa[a] + cn * acc[a];
        }
        if (L.active) {
#pragma unroll
          for (int a = 0; a < ND; ++a) out[ubase + (a * DIM + i) * 64] = acc[a];
        }
      }
    } else {
      const double lam = A.per_cell ? A.lam[e] : A.lam0;
      const double mu = A.per_cell ? A.mu[e] : A.mu0;
      const long sbase = (g * (long)ND) * DIM * DIM * 64 + lane;
      // Two sweeps over the velocity, so that each component is read twice and not once per result that needs it
      // (four times: re-reads of a 41 KB cell group miss the L2 with 2000 waves in flight):
      //   A: W_kk of the three components -> trace and the diagonal entries;
      //   B: component i gives W_ij and W_ik along the other two axes, summed into the three pair accumulators.
      // The operands of sweep B's first component are requested before sweep A's results are stored (a load requested
      // behind a store waits for the store's acknowledgement).
      auto load_comp = [&](int i, double* q) __attribute__((always_inline)) {
        const double* o2 = own;
        asm volatile("" : "+v"(o2));        // a second read of the same component really is one (see above)
#pragma unroll
        for (int b = 0; b < ND; ++b) q[b] = o2[(b * NC + i) * 64];
      };
      auto load_trace = [&](int i, int k, double (&tn)[2][NF]) __attribute__((always_inline)) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int bp = 0; bp < NF; ++bp) tn[s2][bp] = np[2 * k + s2][noff(2 * k + s2, bp % N1, bp / N1, i, i)];
      };
      // acc[a] += sc W_ik(a),  W_ik = -(1/h_k) (D1 u_i along k) + sum_s (c n)_{2k+s} lift1[s] u^_i:  u^ = 1/2 (own + neighbour),
      // the own trace on the domain boundary (a boundary lane's "neighbour" is its own cell: elastic.py:214-216)
      auto add_W = [&](const double* q, const double (&tn)[2][NF], int k, double sc, double (&acc)[ND]) __attribute__((always_inline)) {
        const double w0 = 0.5 * sc * cnf[2 * k], w1 = 0.5 * sc * cnf[2 * k + 1], hk = sc * ih[k];
#pragma unroll
        for (int t1 = 0; t1 < N1; ++t1)
#pragma unroll
          for (int t0 = 0; t0 < N1; ++t0) {
            const int bp = t0 + N1 * t1;
            const double f0 = w0 * (q[node_of(k, t0, t1, 0)] + tn[0][bp]);
            const double f1 = w1 * (q[node_of(k, t0, t1, P)] + tn[1][bp]);
#pragma unroll
            for (int am = 0; am < N1; ++am) {
              double v = 0.0;
#pragma unroll
              for (int m = 0; m < N1; ++m) v += D1[am * N1 + m] * q[node_of(k, t0, t1, m)];
              acc[node_of(k, t0, t1, am)] += lift1[am] * f0 + lift1[N1 + am] * f1 - hk * v;
            }
          }
      };
      // (a fused G stage: out = c_self out + c_new rhs, no second operand - stages.cpp)
      auto load_old = [&](int c, double (&po)[ND]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < ND; ++a) po[a] = out[sbase + (a * DIM * DIM + c) * 64];
      };
      auto combine = [&](double (&v)[ND], const double (&po)[ND]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < ND; ++a) v[a] = A.c_self * po[a] + A.c_new * v[a];
      };
      auto store = [&](int c, const double (&v)[ND]) __attribute__((always_inline)) {
        if (L.active) {
#pragma unroll
          for (int a = 0; a < ND; ++a) out[sbase + (a * DIM * DIM + c) * 64] = v[a];
        }
      };
      // the arithmetic that produced v is complete HERE: sched_barrier orders the machine scheduler only, and the
      // optimiser would otherwise sink pure arithmetic to its first use, past the requests that follow (everything
      // requested so far then stays live until that point)
      auto pin = [&](double (&v)[ND]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < ND; ++a) asm volatile("" : "+v"(v[a]));
      };
      // Three finished results -> components c0, c1, c2 of a fused stage: all old values are requested before the first
      // store (no load behind a store): one exposed load latency.
      auto combine3 = [&](int c0, int c1, int c2, double (&v0)[ND], double (&v1)[ND], double (&v2)[ND]) __attribute__((always_inline)) {
        double po[ND], po1[ND], po2[ND];
        load_old(c0, po);
        load_old(c1, po1);
        load_old(c2, po2);
        combine(v0, po);
        combine(v1, po1);
        combine(v2, po2);
        pin(v0);
        pin(v1);
        pin(v2);
        __builtin_amdgcn_sched_barrier(0);
        store(c0, v0);
        store(c1, v1);
        store(c2, v2);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto store3 = [&](int c0, int c1, int c2, const double (&v0)[ND], const double (&v1)[ND], const double (&v2)[ND]) __attribute__((always_inline)) {
        store(c0, v0);
        store(c1, v1);
        store(c2, v2);
        __builtin_amdgcn_sched_barrier(0);
      };

      // ---- sweep A
      double w[DIM][ND];
      // a stage that is not fused keeps the three components for sweep B (HOLD): one read of the velocity instead of two;
      // a fused stage needs those registers for the old values and reads the components again
      constexpr bool HOLD = MODE == 0 && SG_HEX_HOLD;
      double q[DIM][ND];
      {
        double tn[DIM][2][NF];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          load_comp(k, q[k]);
          load_trace(k, k, tn[k]);
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
#pragma unroll
          for (int a = 0; a < ND; ++a) w[k][a] = 0.0;
          add_W(q[k], tn[k], k, 1.0, w[k]);
        }
      }
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        const double tr = lam * (w[0][a] + w[1][a] + w[2][a]);
#pragma unroll
        for (int k = 0; k < DIM; ++k) w[k][a] = 2.0 * mu * w[k][a] + tr;
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k) pin(w[k]);
      __builtin_amdgcn_sched_barrier(0);
      // ---- sweep B: component i's operands are requested while component i - 1 is evaluated (the first one ahead of
      //      sweep A's stores where the stage is not fused)
      double qb[HOLD ? 1 : 2][HOLD ? 1 : ND], tb[2][2][2][NF];
      if (MODE == 1) combine3(0, DIM + 1, 2 * DIM + 2, w[0], w[1], w[2]);     // (registers: the old values instead of the operands)
      if (!HOLD) load_comp(0, qb[0]);
      load_trace(0, 1, tb[0][0]);
      load_trace(0, 2, tb[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 0) store3(0, DIM + 1, 2 * DIM + 2, w[0], w[1], w[2]);
      double pr[DIM][ND];     // the pairs (0,1), (0,2), (1,2): sh_ij = sh_ji = mu (W_ij + W_ji)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int a = 0; a < ND; ++a) pr[p][a] = 0.0;
      if (!HOLD) load_comp(1, qb[HOLD ? 0 : 1]);
      load_trace(1, 0, tb[1][0]);
      load_trace(1, 2, tb[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      add_W(HOLD ? q[0] : qb[0], tb[0][0], 1, mu, pr[0]);       // W_01
      add_W(HOLD ? q[0] : qb[0], tb[0][1], 2, mu, pr[1]);       // W_02
      pin(pr[0]);
      pin(pr[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (!HOLD) load_comp(2, qb[0]);
      load_trace(2, 0, tb[0][0]);
      load_trace(2, 1, tb[0][1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < ND; ++a) pr[2][a] = 0.0;
      add_W(HOLD ? q[1] : qb[HOLD ? 0 : 1], tb[1][0], 0, mu, pr[0]);       // W_10
      add_W(HOLD ? q[1] : qb[HOLD ? 0 : 1], tb[1][1], 2, mu, pr[2]);       // W_12
      pin(pr[0]);
      pin(pr[2]);
      __builtin_amdgcn_sched_barrier(0);
      add_W(HOLD ? q[2] : qb[0], tb[0][0], 0, mu, pr[1]);       // W_20
      add_W(HOLD ? q[2] : qb[0], tb[0][1], 1, mu, pr[2]);       // W_21
      pin(pr[1]);
      pin(pr[2]);
      __builtin_amdgcn_sched_barrier(0);
      if (!SYM) {                                   // the mirror entries have their own old values (the rare path)
#pragma unroll
        for (int p = 0; p < DIM; ++p) {
          const int i = p == 2 ? 1 : 0, j = p == 0 ? 1 : 2;
          double w2[ND];
#pragma unroll
          for (int a = 0; a < ND; ++a) w2[a] = pr[p][a];
          if (MODE == 1) {
            double po[ND];
            load_old(j * DIM + i, po);
            combine(w2, po);
          }
          store(j * DIM + i, w2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 1)
        combine3(1, 2, DIM + 2, pr[0], pr[1], pr[2]);
      else
        store3(1, 2, DIM + 2, pr[0], pr[1], pr[2]);
    }
  }
}

#ifndef SG_HEX_BLOCKS
#define SG_HEX_BLOCKS 2048
#endif
template <int P>
static int launch_hex_p(int kind, const StageArgs& a, long nitems, hipStream_t s) {
  long blocks = (nitems + 3) / 4;
  if (blocks > SG_HEX_BLOCKS) blocks = SG_HEX_BLOCKS;
  blocks = (blocks + 7) / 8 * 8;
  const dim3 grid((unsigned)blocks), block(256);
#define SG_HEX_LAUNCH(K, M)                                                        \
  do {                                                                             \
    if (a.sym)                                                                     \
      SG_LAUNCH((hex_stage<P, K, M, 1>), grid, block, s, a, a);           \
    else                                                                           \
      SG_LAUNCH((hex_stage<P, K, M, 0>), grid, block, s, a, a);           \
  } while (0)
  if (kind == 0) {
    if (a.mode == 0)
      SG_HEX_LAUNCH(0, 0);
    else
      SG_HEX_LAUNCH(0, 1);
  } else {
    if (a.mode == 0)
      SG_HEX_LAUNCH(1, 0);
    else
      SG_HEX_LAUNCH(1, 1);
  }
#undef SG_HEX_LAUNCH
  return (int)hipGetLastError();
}

template <int DIM, int P>
static int launch_lane_dp(int kind, const StageArgs& a, long nitems, hipStream_t s) {
  long blocks = (nitems + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  blocks = (blocks + 7) / 8 * 8;  // every XCD label needs a block
  const dim3 grid((unsigned)blocks), block(256);
#define SG_LANE_LAUNCH(K, M)                                                            \
  do {                                                                                  \
    if (a.sym)                                                                          \
      SG_LAUNCH((lane_stage<DIM, P, K, M, 1>), grid, block, s, a, a);          \
    else                                                                                \
      SG_LAUNCH((lane_stage<DIM, P, K, M, 0>), grid, block, s, a, a);          \
  } while (0)
  if (kind == 0) {
    if (a.mode == 0)
      SG_LANE_LAUNCH(0, 0);
    else
      SG_LANE_LAUNCH(0, 1);
  } else {
    if (a.mode == 0)
      SG_LANE_LAUNCH(1, 0);
    else
      SG_LANE_LAUNCH(1, 1);
  }
#undef SG_LANE_LAUNCH
  return (int)hipGetLastError();
}

template <int DIM>
static int launch_lane_d(int kind, int P, const StageArgs& a, long nitems, hipStream_t s) {
  switch (P) {
    case 1: return launch_lane_dp<DIM, 1>(kind, a, nitems, s);
    case 2: return launch_lane_dp<DIM, 2>(kind, a, nitems, s);
    case 3: return launch_lane_dp<DIM, 3>(kind, a, nitems, s);
    case 4: return launch_lane_dp<DIM, 4>(kind, a, nitems, s);
  }
  return -1;
}

int launch_stage_lane(int kind, int dim, int P, const StageArgs& a, long nitems, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.tensor) {
    if (dim == 3 && P == 1) return launch_hex_p<1>(kind, a, nitems, s);
    if (dim == 3 && P == 2) return launch_hex_p<2>(kind, a, nitems, s);
    return -1;
  }
  if (dim == 1) return launch_lane_d<1>(kind, P, a, nitems, s);
  if (dim == 2) return launch_lane_d<2>(kind, P, a, nitems, s);
  if (dim == 3 && P == 1) return launch_lane_dp<3, 1>(kind, a, nitems, s);
  if (dim == 3 && P == 2) return launch_lane_dp<3, 2>(kind, a, nitems, s);
  return -1;
}

}  // namespace sg
