// MFMA tile stage kernels for 2-D blocks: triangles P1..P4 and quadrilaterals DQ_1..4 (gfx950; FP64, and float as the
// second mode).
//
// Same data layout idea as the 3-D MFMA path (kernels_mfma.hip): gw = 16, i.e. the 16 values of one
// (node, component) of 16 consecutive squares' cells of one class form one 128-byte line, and one
// wavefront owns one such 16-cell group (an "item") at a time.  Lane l: cell l & 15, k / row-quad l >> 4.
// The element-local contractions of seigen/elastic.py:204-219 (+ the inverse mass of :358-367) are
//
//     [operator tile (nd rows) x 4]  x  [4 nodes x 16 cells]  ->  [nd rows x 16 cells]
//
// With nd <= 16 every operator is at most ONE 16-row tile per k-step (P3, P4, DQ_2, DQ_3: v_mfma_f64_16x16x4_f64;
// DQ_4's 25 rows: two such tiles, worked off one after the other)
// or one / two 4-row tiles (P1, P2: v_mfma_f64_4x4x4_4b_f64), so the whole operator set of a degree is
// 5..14 doubles per lane and lives in REGISTERS for the life of the wave: no LDS, no barrier.
//
// These kernels exist for blocks of 10^4..10^6 cells, where a stage is a few microseconds of data and the
// time goes into DEPENDENT memory round trips, not bandwidth.  Hence:
//  * everything the kernel needs to know about the mesh (sizes, class constants, neighbour rules, node
//    permutations packed four to a word) travels BY VALUE in the kernarg segment (T2Const): no load
//    whose address depends on another load, except the data themselves;
//  * a wave's items all have the same class (item = 2 group + class and every stride is even), so the
//    class constants are selected once per wave;
//  * all of an item's operands - own rows, the three neighbour traces, old values of the fused combine,
//    per-cell coefficients - are requested before the first MFMA: one memory latency per item.
//
//   F:  uh_i  = -sum_r E_r (Jinv_rj T_ij) + sum_f (1/2 L_f) [ w_f (c n)_f,j T(nbr)_ij ]  - sponge
//   G:  W_ik  = -Jinv_rk (E_r u_i) + sum_f (c n)_f,k (1/2 L_f) u(nbr)_i ;  sh = lam tr(W) I + mu (W + W^T)
// E_r = D_r - C_r folds the own-trace half of the central flux into the volume operator
// (mfma_tables.cpp); a boundary lane's "neighbour" is its own cell (w_f = -1 in F: T.n = 0; the missing
// half of the own trace in G).
#include <hip/hip_runtime.h>

#include <cstring>

#include "kernels.hpp"
#include "mfma_tables.hpp"   // SG_T2_LARGE_FROM

namespace sg {

// TP = 1: quadrilateral cells, tensor-product element DQ_P (one class per square, four facets; DQ_4: two 16-row
// tiles, MT below)
// R = float (sg_config::dtype = 1, the second mode of SURVEY 8d): v_mfma_f32_16x16x4_f32 for every degree (gfx950 has
// no 4-row f32 shape with K = 4), one zero-padded 16-row tile; its C/D rows are 4 (lane >> 4) + reg, so the float
// operator tiles hold node 4 (i & 3) + (i >> 2) in MFMA row i (mfma_tables.cpp tile2d_frags32_*) and accumulator
// register m of lane group q is node row 4 m + q, as in double.
template <int P, int TP = 0, typename R = double>
struct TG : ElemDims<2, P, TP> {
  using ElemDims<2, P, TP>::ND;
  using ElemDims<2, P, TP>::NF;
  using ElemDims<2, P, TP>::NFACES;
  using ElemDims<2, P, TP>::NCLS;
  static constexpr int KS = (ND + 3) / 4;    // k-steps over the element nodes
  static constexpr int KSF = (NF + 3) / 4;   // k-steps over the facet nodes
  static constexpr int S4 = (ND + 3) / 4;    // row-quads of the result
  static constexpr bool LARGE = sizeof(R) == 4 || ND > SG_T2_LARGE_FROM;     // one 16-row tile (P3, P4; float) or S4 4-row tiles (P1, P2)
  static constexpr int RT = LARGE ? 1 : S4;  // A fragments per (operator, k-step) and row tile
  static constexpr int MT = LARGE ? (ND + 15) / 16 : 1;   // 16-row tiles: 1, or 2 for DQ_4 (25 rows) - the tiles of an item
                                                          // are worked off one after the other (one set of accumulators)
  static constexpr int S4T = MT > 1 ? 4 : S4;              // row-quads per tile pass
  static constexpr int NFRAG_V = 2 * KS * RT;              // per row tile
  static constexpr int NFRAG_L = NFACES * KSF * RT;
};

typedef double t2d4 __attribute__((ext_vector_type(4)));
typedef float t2f4 __attribute__((ext_vector_type(4)));
template <typename R> struct T2V { typedef t2d4 v4; };
template <> struct T2V<float> { typedef t2f4 v4; };

// acc (rows 4 reg + q of the 16 cells) += A x B.  LARGE: a[0] is the 16-row fragment; else a[t] is the
// 4-row fragment of row-quad t (lane l of its result holds row l >> 4 of cell l & 15).
template <bool LARGE, int S4>
__device__ __forceinline__ void t2_mma(const double* a, double b, t2d4& acc) {
  if constexpr (LARGE) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int t = 0; t < S4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t], b, acc[t], 0, 0, 0);
  }
}
template <bool LARGE, int S4>
__device__ __forceinline__ void t2_mma(const float* a, float b, t2f4& acc) {
  static_assert(LARGE, "float kernels use the 16-row tile at every degree");
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b, acc, 0, 0, 0);
}

// Access through a wave-uniform base and a 32-bit lane offset in bytes: global_load/store with an SGPR base,
// no 64-bit lane arithmetic (each 64-bit lane pointer costs two VALU adds and two registers).
template <typename R>
__device__ __forceinline__ R t2_ld(const R* ubase, unsigned boff) {
  return *reinterpret_cast<const R*>(reinterpret_cast<const char*>(ubase) + boff);
}
template <typename R>
__device__ __forceinline__ void t2_st(R* ubase, unsigned boff, R v) {
  *reinterpret_cast<R*>(reinterpret_cast<char*>(ubase) + boff) = v;
}

// row q of a packed table word (four byte entries: the rows 4 ks + q, q = 0..3, of one k-step)
__device__ __forceinline__ int t2_row(uint32_t word, int q) { return (int)((word >> (8 * q)) & 0xffu); }

template <int P, int KIND, int MODE, int SYM, int GHOST, int TP = 0, typename R = double>
__global__ __launch_bounds__(256) void tile2d_stage(const StageArgs A, const T2Const C) {
  using G = TG<P, TP, R>;
  typedef typename T2V<R>::v4 v4;
  constexpr unsigned EBY = sizeof(R), LB = 16 * sizeof(R);   // bytes per value / per 16-cell line
  constexpr int ND = G::ND, NF = G::NF, KS = G::KS, KSF = G::KSF, S4 = G::S4, RT = G::RT;
  constexpr int NFC = G::NFACES, NCLS = G::NCLS, MT = G::MT, S4T = G::S4T;
  constexpr bool LARGE = G::LARGE;
  constexpr int NC = (KIND == 0) ? 4 : 2;  // input components per node
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, w = lane & 15;
  const R* __restrict__ in = reinterpret_cast<const R*>(A.in);
  const R* aux = reinterpret_cast<const R*>(A.aux);
  R* out = reinterpret_cast<R*>(A.out);

  // graph replay with a source: the first stage of a step names the step (kernels.hpp SrcStep; the stages that read
  // the counter - the G stages - run strictly before and after this launch, never beside it)
  if (KIND == 0 && A.src_bump != 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t* ctr = const_cast<int64_t*>(A.src_step.ctr);
    *ctr = *ctr + 1;
  }
  const bool listed = A.item_list != nullptr;  // a region of a split stage: its active items (stages.cpp)
  const int nitems = listed ? A.nlist : C.ngroups * NCLS;
  // one contiguous item range per XCD label (blocks with equal blockIdx % 8 share an L2); ranges start on
  // even items and every stride is even, so a wave keeps its class
  const int nblk = (int)gridDim.x, xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
  const int blocks_here = (nblk - xcd + 7) >> 3, ipx = ((nitems + 15) >> 4) * 2;
  const int hi = (xcd + 1) * ipx < nitems ? (xcd + 1) * ipx : nitems;
  const int i0 = A.spread ? (int)blockIdx.x * 4 + wave : xcd * ipx + slot * 4 + wave;
  const int i1 = A.spread ? nitems : hi;
  const int istep = A.spread ? nblk * 4 : blocks_here * 4;
  const int k = (NCLS == 2) ? (i0 & 1) : 0;  // class of every item of this wave (item lists hold (group, 0), (group, 1) pairs)
  const T2Class& K = C.cls[k];

  // ---- operator fragments: registers, once per wave -----------------------------------------
  // (MT > 1: the fragments of the row tile in work, re-read per item and tile - both tiles' would not fit the registers)
  R Av[G::NFRAG_V], Al[G::NFRAG_L];
  auto load_frags = [&](int tile) {
#pragma unroll
    for (int j = 0; j < G::NFRAG_V; ++j) Av[j] = reinterpret_cast<const R*>(A.fragV)[(tile * G::NFRAG_V + j) * 64 + lane];
#pragma unroll
    for (int j = 0; j < G::NFRAG_L; ++j) Al[j] = reinterpret_cast<const R*>(A.fragL)[(tile * G::NFRAG_L + j) * 64 + lane];
  };
  if constexpr (MT == 1) load_frags(0);

  // ---- class constants: one batch of scalar loads from the kernarg segment at an offset that depends on k only
  R Jv[2][2], cnv[NFC][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j) Jv[r][j] = (R)K.Jinv[r][j];
#pragma unroll
  for (int f = 0; f < NFC; ++f)
#pragma unroll
    for (int j = 0; j < 2; ++j) cnv[f][j] = (R)K.cn[f][j];
  int f_axis[NFC], f_dir[NFC], f_kn[NFC], f_ord[NFC];
  uint32_t f_tf[NFC][KSF], f_tg[NFC][KSF];
#pragma unroll
  for (int f = 0; f < NFC; ++f) {
    f_axis[f] = K.nb_axis[f];
    f_dir[f] = K.nb_dir[f];
    f_kn[f] = K.nb_cls[f];
    f_ord[f] = K.slot_ord[f];
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
      f_tf[f][ks] = K.tfw[f][ks];
      f_tg[f][ks] = K.tgw[f][ks];
    }
  }
  const int n0 = C.n0, n1 = C.n1;

  for (int it = i0; it < i1; it += istep) {
    const int item = __builtin_amdgcn_readfirstlane(listed ? A.item_list[it] : it);
    const int g = (NCLS == 2) ? (item >> 1) : item;
    // ---- this lane's cell (a padding lane stands in for the group's first square, masked) -------------
    const int c = g * 16 + w;
    const bool valid = c < C.ncube;
    const unsigned cl = valid ? (unsigned)c : (unsigned)(g * 16);
    // cl / n0 through the reciprocal (exact after one correction step; the integer division is 25 instructions)
    unsigned cy = (unsigned)((double)cl * C.inv_n0);
    int cx = (int)(cl - cy * (unsigned)n0);
    if (cx < 0) {
      cx += n0;
      cy -= 1;
    } else if (cx >= n0) {
      cx -= n0;
      cy += 1;
    }
    const int cc[2] = {cx, (int)cy};
    bool active = valid;
    if (listed) {  // a region of a split stage: mask by its boxes (REGION_ALL has no list and covers the block)
      bool inbox = false;
      for (int bx = 0; bx < A.nbox; ++bx)
        inbox = inbox || (cc[0] >= A.boxes_o[bx][0] && cc[0] < A.boxes_o[bx][0] + A.boxes_n[bx][0] &&
                          cc[1] >= A.boxes_o[bx][1] && cc[1] < A.boxes_o[bx][1] + A.boxes_n[bx][1]);
      active = valid && inbox;
      if (!__any(active)) continue;
    }
    const R* ownb = in + ((long)item * ND) * NC * 16;   // wave-uniform
    const int e = (int)cl * NCLS + k;  // cell index in the host numbering
    // G: does this item hold source nodes?  (a scalar load that is consumed only in the epilogue)
    const int sslot_src = (KIND == 1 && A.src_slot != nullptr) ? A.src_slot[item] : -1;

    // ---- own rows: requested first.  B row of this lane at k-step ks = node 4 ks + q; rows past ND meet
    //      all-zero operator columns, so any finite value will do: clamp to node 0
    R ub[KS][NC];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned orow = (unsigned)((((4 * ks + q < ND) ? 4 * ks + q : 0) * NC * 16 + w) * EBY);
      if constexpr (KIND == 0) {
        ub[ks][0] = t2_ld(ownb, orow + 0 * LB);
        ub[ks][1] = t2_ld(ownb, orow + 1 * LB);
        ub[ks][3] = t2_ld(ownb, orow + 3 * LB);
        ub[ks][2] = SYM ? ub[ks][1] : t2_ld(ownb, orow + 2 * LB);
      } else {
        ub[ks][0] = t2_ld(ownb, orow);
        ub[ks][1] = t2_ld(ownb, orow + LB);
      }
    }

    // ---- neighbour traces.  Inside the block's field every neighbour lies at most one row of squares
    //      away: offsets are taken from the uniform base of group g - gpr (gpr groups >= one row), so
    //      they are small and non-negative.  GHOST = 1 lanes may read a packed remote trace instead:
    //      a lane pointer then.
    const int g0 = (g > C.gpr) ? g - C.gpr : 0;
    const R* nbb = in + ((long)g0 * NCLS * ND) * NC * 16;  // wave-uniform
    R tn[NFC][KSF][NC];
    R wf[NFC];
#pragma unroll
    for (int f = 0; f < NFC; ++f) {
      const int axis = f_axis[f], kn = f_kn[f], dir = f_dir[f];
      unsigned noff;  // bytes from nbb to the neighbour cell (component 0 of node 0, this lane's column)
      const R* gp = nullptr;
      bool ghost = false, physical = false;
      if (axis < 0) {
        noff = (unsigned)(((((g - g0) * NCLS + kn) * ND) * NC * 16 + w) * EBY);
      } else {
        const int cn = cc[axis] + dir;
        const bool inside = cn >= 0 && cn < (axis == 0 ? n0 : n1);
        const int nc = inside ? (int)cl + dir * (axis == 0 ? 1 : n0) : (int)cl;
        noff = (unsigned)((((((nc >> 4) - g0) * NCLS + (inside ? kn : k)) * ND) * NC * 16 + (nc & 15)) * EBY);
        physical = !inside;
        if (GHOST) {
          const int side = 2 * axis + (dir > 0 ? 1 : 0);
          if (!inside && C.has_nbr[side]) {
            const long slot2 = (long)(axis == 0 ? cc[1] : cc[0]) * C.halo_per_cube + f_ord[f];
            gp = reinterpret_cast<const R*>(A.ghost[side]) + slot2 * NF * 2;  // packed trace: 2 comps per facet node (velocity, or T_i,axis)
            ghost = true;
            physical = false;
          }
        }
      }
      // F: +1/2 neighbour flux inside, -1/2 own flux on the boundary (cancels the folded half: T.n = 0);
      // G: 1/2 of the neighbour, or the missing 1/2 of the own trace; the 1/2 is part of the lift tiles
      wf[f] = (KIND == 0 && physical) ? (R)-1 : (R)1;
#pragma unroll
      for (int ks = 0; ks < KSF; ++ks) {
        // facet-trace row: the neighbour's node inside the block's field, the own node on the domain
        // boundary, the position in the neighbour's facet list in a packed remote trace
        const unsigned roff = noff + (unsigned)(t2_row(physical ? C.tpw[f][ks] : f_tf[f][ks], q) * NC * LB);
        const int grow = GHOST ? t2_row(f_tg[f][ks], q) * 2 : 0;
        const R* fp = reinterpret_cast<const R*>(reinterpret_cast<const char*>(nbb) + roff);
        if constexpr (KIND == 0) {
          // a packed remote trace holds g_i = T_i,axis only; the columns j != axis meet (c n)_j = 0 there,
          // so any finite value serves: pairs (i <= j): axis 0 -> g_j, axis 1 -> g_i; full tensor: g_i
          auto at = [&](int i, int j) {
            const int cidx = (SYM && i > j) ? j * 2 + i : i * 2 + j;
            if (!GHOST) return t2_ld(nbb, roff + cidx * LB);
            const int og = SYM ? (axis == 0 ? j : i) : i;
            return ghost ? gp[grow + og] : fp[cidx * 16];
          };
          tn[f][ks][0] = at(0, 0);
          tn[f][ks][1] = at(0, 1);
          tn[f][ks][3] = at(1, 1);
          tn[f][ks][2] = SYM ? tn[f][ks][1] : at(1, 0);
        } else {
          if (!GHOST) {
            tn[f][ks][0] = t2_ld(nbb, roff);
            tn[f][ks][1] = t2_ld(nbb, roff + LB);
          } else {
            tn[f][ks][0] = ghost ? gp[grow] : fp[0];
            tn[f][ks][1] = ghost ? gp[grow + 1] : fp[16];
          }
        }
      }
    }

    if (KIND == 0) {
      // ---- F ---------------------------------------------------------------------------------
      const long ubase = ((long)item * ND) * 2 * 16;
      R* outb = out + ubase;            // wave-uniform
      const R* auxb = aux + ubase;
      // sponge of this lane's cell: none (sig = 0), a sigma that is constant over the cell (sig: the term is sigma u at the
      // node itself, no matrix) or a varying one (sig = NaN: the cell's matrix B_e in slot sslot) - kernels.hpp sponge_sigma
      int sslot = -1;
      R sig = (R)0;
      if (A.sponge_sigma != nullptr && active) {
        sig = (R)A.sponge_sigma[e];
        if (sig != sig) sslot = A.sponge_slot[e];
      }
      R cs = (R)A.c_self, ca = (R)A.c_aux, cnw = (R)A.c_new;
      if (MODE == 1 && A.rho2 != nullptr) {  // per-cell density (kernels.hpp)
        const R r0 = (R)A.rho2[2 * e], r1 = (R)A.rho2[2 * e + 1];
        cs = r0;
        ca *= r1;
        cnw *= r1;
      }
      // sponge (elastic.py:207-208): -sum_b B_e[a][b] u_abs[b][i] on the lanes whose cell carries sigma, for every
      // row-quad of the item.  With the in-place combine u_abs may be `out`: every lane of a cell sits in this wave
      // and the wave runs in program order, so these reads precede the stores (MT > 1: they are done before the
      // first row tile is stored).
      const bool sponge_here = __any(sig != (R)0);     // (NaN != 0: the lanes with a matrix count)
      R ssum[S4][2];
      auto sponge_sums = [&]() {
        if (sslot < 0 && sig != (R)0) {     // constant sigma: sigma u_i at the lane's own rows
          const R* ua = reinterpret_cast<const R*>(A.uabs) + ubase + w;
#pragma unroll
          for (int m = 0; m < S4; ++m) {
            const int a = (4 * m + q < ND) ? 4 * m + q : 0;
            ssum[m][0] = sig * ua[(a * 2 + 0) * 16];
            ssum[m][1] = sig * ua[(a * 2 + 1) * 16];
          }
        }
        if (sslot >= 0) {
          const R* ua = reinterpret_cast<const R*>(A.uabs) + ubase + w;
          const double* B = A.sponge_B + ((long)sslot * ND + q) * ND;  // row a = 4 m + q: B + 4 m ND (double in both modes)
#pragma unroll
          for (int m = 0; m < S4; ++m) ssum[m][0] = ssum[m][1] = (R)0;
          // chunks of CH columns: all of a chunk's loads are in flight together (one round trip per chunk)
          constexpr int CH = ND <= 6 ? ND : (ND % 5 == 0 ? 5 : (ND % 4 == 0 ? 4 : 3));
          static_assert(ND % CH == 0, "the sponge column chunks must tile the element's nodes");
#pragma nounroll
          for (int b0 = 0; b0 < ND; b0 += CH) {
            R uu[CH][2], bb[S4][CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              uu[j][0] = ua[((b0 + j) * 2 + 0) * 16];
              uu[j][1] = ua[((b0 + j) * 2 + 1) * 16];
#pragma unroll
              for (int m = 0; m < S4; ++m) bb[m][j] = (R)B[((4 * m + q < ND) ? m * 4 * ND : 0) + b0 + j];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
              for (int m = 0; m < S4; ++m) {
                ssum[m][0] += bb[m][j] * uu[j][0];
                ssum[m][1] += bb[m][j] * uu[j][1];
              }
          }
        }
      };
      if (MT > 1 && sponge_here) sponge_sums();
#pragma unroll
      for (int tile = 0; tile < MT; ++tile) {
      if constexpr (MT > 1) load_frags(tile);
      // in-place combine operands, requested before the arithmetic
      // (MODE 2: no self term, out = c_aux aux + c_new rhs - stage UTEMP leaves w = dt u1 + dt^3/24 utemp, stages.cpp)
      R po[S4T][2], pa[S4T][2];
      if (MODE >= 1) {
#pragma unroll
        for (int m = 0; m < S4T; ++m) {
          const int a = 16 * tile + 4 * m + q;
          const unsigned ro = (unsigned)((((a < ND) ? a : 0) * 2 * 16 + w) * EBY);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            po[m][i] = MODE == 1 ? t2_ld(outb, ro + i * LB) : R(0);
            pa[m][i] = t2_ld(auxb, ro + i * LB);
          }
        }
      }
      v4 acc[2] = {v4{0, 0, 0, 0}, v4{0, 0, 0, 0}};
      // volume: B = T~_ir = Jinv_rj T_ij (the fragments hold -E_r)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const R tt = Jv[r][0] * ub[ks][i * 2 + 0] + Jv[r][1] * ub[ks][i * 2 + 1];
            t2_mma<LARGE, S4>(&Av[(r * KS + ks) * RT], tt, acc[i]);
          }
      // lifts of w_f (c n)_j T(nbr)_ij
#pragma unroll
      for (int f = 0; f < NFC; ++f)
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const R fl = wf[f] * (cnv[f][0] * tn[f][ks][i * 2 + 0] + cnv[f][1] * tn[f][ks][i * 2 + 1]);
            t2_mma<LARGE, S4>(&Al[(f * KSF + ks) * RT], fl, acc[i]);
          }
      if (sponge_here) {
        if (MT == 1) sponge_sums();
        if (sig != (R)0) {
#pragma unroll
          for (int m = 0; m < S4T; ++m)
            if (4 * tile + m < S4) {
              acc[0][m] -= ssum[4 * tile + m][0];
              acc[1][m] -= ssum[4 * tile + m][1];
            }
        }
      }
#pragma unroll
      for (int m = 0; m < S4T; ++m) {
        const int a = 16 * tile + 4 * m + q;
        const unsigned ro = (unsigned)((a * 2 * 16 + w) * EBY);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          R v = acc[i][m];
          if (MODE == 1) v = cs * po[m][i] + ca * pa[m][i] + cnw * v;
          if (MODE == 2) v = ca * pa[m][i] + cnw * v;
          if (active && a < ND) t2_st(outb, ro + i * LB, v);
        }
      }
      }  // row tiles
    } else {
      // ---- G ---------------------------------------------------------------------------------
      const R lam = (R)(A.per_cell ? A.lam[e] : A.lam0);
      const R mu = (R)(A.per_cell ? A.mu[e] : A.mu0);
      // fused form of a G stage: out = c_self out + c_new rhs - no second operand (stage S1 gets dt sh1 + dt^3/24 sh2 as one
      // G of dt u1 + dt^3/24 utemp, stages.cpp); the source enters with src_coef = dt + dt^3/24 there
      const R c_self = (R)A.c_self, c_new = (R)A.c_new;
      const R src_coef = (R)A.src_coef;
      const long sbase = ((long)item * ND) * 4 * 16;
      R* outb = out + sbase;            // wave-uniform
#pragma unroll
      for (int tile = 0; tile < MT; ++tile) {
      if constexpr (MT > 1) load_frags(tile);
      R po[S4T][3], pl[S4T];  // old values of the lines (0,0) (0,1) (1,1), and (1,0)
      if (MODE == 1) {
#pragma unroll
        for (int m = 0; m < S4T; ++m) {
          const int a = 16 * tile + 4 * m + q;
          const unsigned ro = (unsigned)((((a < ND) ? a : 0) * 4 * 16 + w) * EBY);
          po[m][0] = t2_ld(outb, ro + 0 * LB);
          po[m][1] = t2_ld(outb, ro + 1 * LB);
          po[m][2] = t2_ld(outb, ro + 3 * LB);
          if (!SYM) pl[m] = t2_ld(outb, ro + 2 * LB);
        }
      }
      // W_00, W_11 and W_01 + W_10 per row-quad
      v4 Sd[2] = {v4{0, 0, 0, 0}, v4{0, 0, 0, 0}}, So = v4{0, 0, 0, 0};
      auto fold = [&](const R c0, const R c1, const v4 (&v)[2]) {
        // W_ik += c_k v_i
#pragma unroll
        for (int m = 0; m < S4T; ++m) {
          Sd[0][m] += c0 * v[0][m];
          Sd[1][m] += c1 * v[1][m];
          So[m] += c1 * v[0][m] + c0 * v[1][m];
        }
      };
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        v4 acc[2] = {v4{0, 0, 0, 0}, v4{0, 0, 0, 0}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) t2_mma<LARGE, S4>(&Av[(r * KS + ks) * RT], ub[ks][i], acc[i]);
        fold(-Jv[r][0], -Jv[r][1], acc);
      }
#pragma unroll
      for (int f = 0; f < NFC; ++f) {
        v4 acc[2] = {v4{0, 0, 0, 0}, v4{0, 0, 0, 0}};
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) t2_mma<LARGE, S4>(&Al[(f * KSF + ks) * RT], tn[f][ks][i], acc[i]);
        fold(cnv[f][0], cnv[f][1], acc);
      }
#pragma unroll
      for (int m = 0; m < S4T; ++m) {
        const int a = 16 * tile + 4 * m + q;
        const unsigned ro = (unsigned)((a * 4 * 16 + w) * EBY);
        const R tr = lam * (Sd[0][m] + Sd[1][m]);
        R v00 = (R)2 * mu * Sd[0][m] + tr, v11 = (R)2 * mu * Sd[1][m] + tr, v01 = mu * So[m], v10 = v01;
        if (sslot_src >= 0) {  // wave-uniform, a handful of items: + S_ij at the source nodes (elastic.py:217-218)
          const int ix = (a < ND) ? A.src_idx[(sslot_src * ND + a) * 16 + w] : -1;
          const double* svb = A.src_vals;
          double ssc = A.src_scale;
          bool son = true;
          if (A.src_step.ctr != nullptr) {   // graph replay: slice and weight of the step the device-side counter names
            const int64_t st = *A.src_step.ctr;
            son = A.src_step.is_static || st < A.src_step.nsteps;
            const int64_t s2 = son ? st : 0;
            if (!A.src_step.is_static) svb += s2 * A.src_step.stride;
            if (A.src_step.weights != nullptr) ssc = A.src_step.weights[s2];
          }
          if (ix >= 0 && son) {
            const double* sv = svb + (long)ix * 4;
            v00 += src_coef * (R)sg_mul_rounded(ssc, sv[0]);   // rounded product first: bitwise = a table of the products
            v01 += src_coef * (R)sg_mul_rounded(ssc, sv[1]);
            v10 += src_coef * (R)sg_mul_rounded(ssc, sv[2]);
            v11 += src_coef * (R)sg_mul_rounded(ssc, sv[3]);
          }
        }
        if (MODE == 1) {
          v00 = c_self * po[m][0] + c_new * v00;
          v11 = c_self * po[m][2] + c_new * v11;
          if (!SYM) v10 = c_self * pl[m] + c_new * v10;
          v01 = c_self * po[m][1] + c_new * v01;
        }
        if (active && a < ND) {
          t2_st(outb, ro + 0 * LB, v00);
          t2_st(outb, ro + 1 * LB, v01);
          if (!SYM) t2_st(outb, ro + 2 * LB, v10);
          t2_st(outb, ro + 3 * LB, v11);
        }
      }
      }  // row tiles
    }
  }
}

// Blocks of four waves the device holds of one kernel instantiation (asked once per instantiation): the size of the
// persistent grid when the caller names none (StageArgs::grid_blocks = 0, stages.cpp).  Exactly resident is a sharp optimum
// on the meshes of the benchmark protocol - a few blocks more start a second, nearly empty round, a few less leave slots
// unused with the same number of items per wave (profiles/r05/tile_grid_sweep.txt) - and the instantiations differ: the
// fused F stages hold two waves per SIMD at degree 4, the other stages three.
template <typename K>
static int t2_resident_blocks(K kernel) {
  int per_cu = 0, dev = 0, ncu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
    ncu = 256;
  return per_cu * ncu;
}

template <int P, int SYM, int GHOST, int TP = 0, typename R = double>
static int launch_t2(int kind, const StageArgs& a, const T2Const& c, long nitems, hipStream_t s) {
  const dim3 block(256);
#define SG_T2_LAUNCH(K, M)                                                                          \
  do {                                                                                              \
    static const int resident = t2_resident_blocks(tile2d_stage<P, K, M, SYM, GHOST, TP, R>);       \
    long blocks = (nitems + 3) / 4;                                                                 \
    const long cap = a.grid_blocks > 0 ? a.grid_blocks : resident;                                  \
    if (blocks > cap) blocks = cap;                                                                 \
    blocks = (blocks + 7) / 8 * 8; /* every XCD label needs a block */                              \
    const dim3 grid((unsigned)blocks);                                                              \
    SG_LAUNCH((tile2d_stage<P, K, M, SYM, GHOST, TP, R>), grid, block, s, a, a, c);                 \
  } while (0)
  if (kind == 0) {
    if (a.mode == 0)
      SG_T2_LAUNCH(0, 0);
    else if (a.mode == 2)
      SG_T2_LAUNCH(0, 2);
    else
      SG_T2_LAUNCH(0, 1);
  } else {
    if (a.mode == 0)
      SG_T2_LAUNCH(1, 0);
    else
      SG_T2_LAUNCH(1, 1);
  }
#undef SG_T2_LAUNCH
  return (int)hipGetLastError();
}

template <int P, int TP, typename R>
static int launch_t2r(int kind, const StageArgs& a, const T2Const& c, long nitems, hipStream_t s) {
  bool ghosts = false;
  for (int sd = 0; sd < 4; ++sd) ghosts = ghosts || (a.ghost[sd] != nullptr);
  if (a.sym) return ghosts ? launch_t2<P, 1, 1, TP, R>(kind, a, c, nitems, s) : launch_t2<P, 1, 0, TP, R>(kind, a, c, nitems, s);
  return ghosts ? launch_t2<P, 0, 1, TP, R>(kind, a, c, nitems, s) : launch_t2<P, 0, 0, TP, R>(kind, a, c, nitems, s);
}
template <int P, int TP = 0>
static int launch_t2p(int kind, const StageArgs& a, const T2Const& c, long nitems, hipStream_t s) {
  return a.f32 ? launch_t2r<P, TP, float>(kind, a, c, nitems, s) : launch_t2r<P, TP, double>(kind, a, c, nitems, s);
}

int launch_stage_tile2d(int kind, int P, const StageArgs& a, const T2Const& c, long nitems, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.tensor) {
    switch (P) {
      case 1: return launch_t2p<1, 1>(kind, a, c, nitems, s);
      case 2: return launch_t2p<2, 1>(kind, a, c, nitems, s);
      case 3: return launch_t2p<3, 1>(kind, a, c, nitems, s);
      case 4: return launch_t2p<4, 1>(kind, a, c, nitems, s);
    }
    return -1;
  }
  switch (P) {
    case 1: return launch_t2p<1>(kind, a, c, nitems, s);
    case 2: return launch_t2p<2>(kind, a, c, nitems, s);
    case 3: return launch_t2p<3>(kind, a, c, nitems, s);
    case 4: return launch_t2p<4>(kind, a, c, nitems, s);
  }
  return -1;
}

}  // namespace sg
