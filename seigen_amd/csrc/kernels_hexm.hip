// Hexahedral cells of degree 3 and 4 (DQ_3: 64 nodes, DQ_4: 125; sg_config::diagonal = 2, dim 3) - sum-factorised stage
// kernels with the lines of a cube in REGISTERS and the 16 cubes of a cell group on the columns of the matrix
// instructions (gfx950, FP64).
//
// On the cubes of a structured block the tensor-product element factorises (kernels_lane.hip hex_stage):
//   D_r = I x D1 x I along axis r,  L_f = lift1[s] on the facet node with the same transverse indices,
//   Jinv = diag(1 / h),  (c n)_f = -+ 1 / h_m on its own axis only.
// The right-hand sides of seigen/elastic.py:204-219 are then sums of LINE operators: for the line of N1 = P + 1 nodes
// through a node along axis k,
//   out[a'] += sum_a E_k[a'][a] v[a] + lw_k[0][a'] tn_0 + lw_k[1][a'] tn_1,
//   E_k = -D1 / h_k + the own-trace half of the central flux on the line's two end nodes (folded in, as the MFMA
//   kernels of the tetrahedra fold it into their volume tiles),  lw_k[s] = 1/2 (c n)_{2k+s} lift1[s],
// with tn_s the NEIGHBOUR's value across facet 2k + s (on the domain boundary: minus the own value for f - no facet
// term there, elastic.py:206 - and plus the own value for g, :214-216).  F: uh_i = sum_j line_j(T_ij); G: W_ik = line_k(u_i),
// sh_ii = 2 mu W_ii + lam tr W, sh_ij = mu (W_ij + W_ji).  Tables: mfma_tables.cpp hexm_table.
//
// Layout gw = 16 (mesh_tables.hpp): the 16 values of one (node, component) of 16 consecutive cubes are one 128-byte line.
// A wave owns one such group at a time; lane l = (q = l >> 4, w = l & 15) holds, of cube w, the nodes whose FIRST index is
// i0 = q + 4 ks (ks < KSX = ceil(N1 / 4)): NV = KSX N1^2 values per component and lane, every wave load instruction
// fetches four whole lines.  Lines along y and z lie inside a lane: plain FMAs with the operator entries as scalar
// operands.  Lines along x run ACROSS the four lane groups: v_mfma_f64_4x4x4_4b with B = the lane's value (k = q, the 16
// cubes as the columns of the four 4x4 blocks), A = the 4x4 block of E_x (lane l: row l & 3, column l >> 4) and the result
// row q' of cube w back in lane (q', w) - the mapping the values are held in; the two x-facet traces ride along as one
// more k-step (lane group 0: facet 0, group 1: facet 1, A = the lift columns).
// One thread-per-node kernel did these elements before (kernels.hip, TP = 2: 34 - 37 G DoF-updates/s, latency-bound,
// profiles/r04/hexahedra.txt).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sg {

template <int P>
struct HX {
  static constexpr int N1 = P + 1, ND = N1 * N1 * N1, NF = N1 * N1;
  static constexpr int KSX = (N1 + 3) / 4;
  static constexpr int NV = KSX * N1 * N1;
  // table (doubles, StageArgs::Dt): E[3][N1][N1], lw[3][2][N1], then the x-pass A operands in lane order
  static constexpr int OFF_LW = 3 * N1 * N1, OFF_AX = OFF_LW + 6 * N1, OFF_AT = OFF_AX + KSX * KSX * 64, SIZE = OFF_AT + KSX * 64;
};

#define HXM_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)
#define HXM_ST(p, v) __builtin_nontemporal_store((v), (p))
#define HXM_LDS(p) __builtin_nontemporal_load(p)

// waves per SIMD: DQ_3 holds everything twice over in 256 registers; DQ_4 (50 values per component and lane) needs the
// whole register file for three sets of results and a component of operands
template <int P>
struct HXW {
  static constexpr int WPE = P <= 3 ? 2 : 1;
};

template <int P, int KIND, int MODE, int SYM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HXW<P>::WPE, HXW<P>::WPE))) void hexm_stage(StageArgs A) {
  using H = HX<P>;
  constexpr int N1 = H::N1, ND = H::ND, NF = H::NF, KSX = H::KSX;
  constexpr int NC = (KIND == 0) ? 9 : 3;
  constexpr double SGN = (KIND == 0) ? -1.0 : 1.0;   // a boundary lane's "neighbour" value: -own (f) / +own (g)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, w = lane & 15;
  typedef __attribute__((address_space(4))) const double cdouble;
  typedef __attribute__((address_space(4))) const MeshDev cMeshDev;
  const cdouble* tab = (const cdouble*)(unsigned long long)A.Dt;
  const cMeshDev* md = (const cMeshDev*)(unsigned long long)A.md;
  const double* __restrict__ in = A.in;
  const double* __restrict__ aux = A.aux;
  double* __restrict__ out = A.out;
  auto E = [&](int k, int ap, int a) -> double { return tab[(k * N1 + ap) * N1 + a]; };
  auto LW = [&](int k, int s, int ap) -> double { return tab[H::OFF_LW + (k * 2 + s) * N1 + ap]; };

  // A operands of the x pass (item-invariant, in registers)
  double ax[KSX][KSX], at[KSX];
#pragma unroll
  for (int a = 0; a < KSX; ++a) {
#pragma unroll
    for (int b = 0; b < KSX; ++b) ax[a][b] = A.Dt[H::OFF_AX + (a * KSX + b) * 64 + lane];
    at[a] = A.Dt[H::OFF_AT + a * 64 + lane];
  }
  // the lane's first node index per k-step (clamped: a padded row meets zero operator columns and is never stored)
  int i0c[KSX];
  bool i0ok[KSX];
#pragma unroll
  for (int ks = 0; ks < KSX; ++ks) {
    i0ok[ks] = q + 4 * ks < N1;
    i0c[ks] = i0ok[ks] ? q + 4 * ks : 0;
  }

  // degree 4, G stages: the component held through the whole item (u_2) lives in LDS (a wave-private 16 KB), not in 100
  // registers: with it the three phases of G fit the register file without spilling
  constexpr bool U2_LDS = KIND == 1 && P >= 4;
  __shared__ double sU2[U2_LDS ? 4 * ND * 16 : 1];
  (void)sU2;

  const long ncube = md->ncube;
  const long ngroups = md->ncube_pad >> 4;
  const bool listed = A.item_list != nullptr;
  const long nitems = listed ? (long)A.nlist : ngroups;
  const long nblk = gridDim.x, xcd = blockIdx.x % 8, slot8 = blockIdx.x / 8;
  const long blocks_here = (nblk - xcd + 7) / 8, ipx = (nitems + 7) / 8;
  const long lo = xcd * ipx, hi = (xcd + 1) * ipx < nitems ? (xcd + 1) * ipx : nitems;
  const long it0 = A.spread ? (long)blockIdx.x * 4 + wave : lo + slot8 * 4 + wave;
  const long it1 = A.spread ? nitems : hi;
  const long istep = A.spread ? (long)gridDim.x * 4 : blocks_here * 4;
  const int n0 = md->n[0], n1 = md->n[1], n2 = md->n[2];

  for (long it = it0; it < it1; it += istep) {
    const long g = listed ? (long)A.item_list[it] : it;
    // this lane's cube
    const long c = g * 16 + w;
    const bool valid = c < ncube;
    int cc[3];
    {
      const unsigned cl = valid ? (unsigned)c : 0u;
      const unsigned t = cl / (unsigned)n0, z = t / (unsigned)n1;
      cc[0] = (int)(cl - t * (unsigned)n0);
      cc[1] = (int)(t - z * (unsigned)n1);
      cc[2] = (int)z;
    }
    bool active = valid;
    if (!A.all_active) {
      bool inb = false;
      for (int bx = 0; bx < A.nbox; ++bx) {
        bool ib = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) ib = ib && (cc[a] >= A.boxes_o[bx][a]) && (cc[a] < A.boxes_o[bx][a] + A.boxes_n[bx][a]);
        inb = inb || ib;
      }
      active = valid && inb;
    }
    if (!__any(active)) continue;
    const double* const ug = in + (g * (long)ND) * NC * 16;   // wave-uniform: the group's cell data ([node][comp][16 cubes])
    const long e = valid ? c : 0;
    // index of the lane's own node (i0c[ks], 0, 0), component 0; node (., i1, i2) is a compile-time distance further on
    int lown[KSX];
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks) lown[ks] = i0c[ks] * NC * 16 + w;
    constexpr int D1N = N1 * NC * 16, D2N = N1 * N1 * NC * 16;   // one step in i1 / i2 inside a cell

    // Where each facet's neighbour values live: the neighbour cube's cell, a packed remote trace ([cube on the side][facet
    // node][3]: the velocity, or T_i,axis of a stress), or - on the domain boundary - the own cell, whose value counts
    // with the sign SGN.  Per lane and facet: a pointer to the facet node with the lane's own transverse position (and
    // second transverse index 0), the distance to the next facet node along the second (third) transverse index, and how a
    // component is reached - all three differ between a field and a remote record, so they are lane VALUES, not branches.
    const double* pf[6][KSX];   // facets 2..5 (y, z): transverse position (i0c[ks], tt = 0); facets 0, 1 see below
    int st[6];                  // distance from tt to tt + 1
    double sg[6];
    bool gh[6];
    const double* pxl = nullptr;   // x facets: lane group 0 reads facet 0, lane group 1 facet 1 (groups 2, 3: any finite value)
    int stx1 = 0, stx2 = 0;
    double sgx = 1.0;
    bool ghx = false;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const int axis = f >> 1, dir = (f & 1) ? 1 : -1;
      const int cnb = cc[axis] + dir;
      const int nax = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
      const bool inside = valid && cnb >= 0 && cnb < nax;
      const long stride = axis == 0 ? 1 : (axis == 1 ? (long)n0 : (long)n0 * n1);
      const long nc = inside ? c + dir * stride : (valid ? c : 0);
      const double* cell = in + ((nc >> 4) * (long)ND) * NC * 16 + (nc & 15);
      const bool ghost = !inside && valid && md->has_nbr[f] != 0;
      const bool phys = !inside && !ghost;
      // the facet node's index along the facet's own axis: the far end of the neighbour, the near end of the own cell
      const int fix = phys ? ((f & 1) ? P : 0) : ((f & 1) ? 0 : P);
      const long c2 = axis == 0 ? (cc[1] + (long)n1 * cc[2]) : (axis == 1 ? (cc[0] + (long)n0 * cc[2]) : (cc[0] + (long)n0 * cc[1]));
      const double* rec = ghost ? A.ghost[f] + c2 * NF * 3 : nullptr;
      gh[f] = ghost;
      sg[f] = phys ? SGN : 1.0;
      if (axis == 0) {
        if ((q & 1) == (f & 1)) {
          pxl = ghost ? rec : cell + fix * NC * 16;
          stx1 = ghost ? 3 : D1N;
          stx2 = ghost ? N1 * 3 : D2N;
          sgx = sg[f];
          ghx = ghost;
        }
      } else {
        st[f] = ghost ? N1 * 3 : (axis == 1 ? D2N : D1N);
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks)
          pf[f][ks] = ghost ? rec + i0c[ks] * 3 : cell + (i0c[ks] + (axis == 1 ? N1 * fix : N1 * N1 * fix)) * NC * 16;
      }
    }

    // ---- operand access.  A PLANE is the nodes with one value of the third index i2: per lane KSX x N1 values (ks, i1).
    //      The lines along x and y lie inside a plane, only the lines along z cross planes - so a stage holds ONE whole
    //      component (the one differentiated along z) in registers and streams everything else plane by plane: few live
    //      values, several planes of requests in flight.
    // (Every access below starts from a base made OPAQUE at that point: the offsets of an item's loads are item-invariant
    // lane values, and the optimiser otherwise computes all of them once, in front of the item loop, and keeps hundreds of
    // registers of addresses alive through the whole kernel - or spills them.)
    auto opq = [](int v) __attribute__((always_inline)) -> int {
      asm volatile("" : "+v"(v));
      return v;
    };
    auto opqp = [](const double* p) __attribute__((always_inline)) -> const double* {
      asm volatile("" : "+v"(p));
      return p;
    };
    auto load_full = [&](int cf, double (&U)[KSX][N1][N1]) __attribute__((always_inline)) {
      int lo[KSX];
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) lo[ks] = opq(lown[ks]) + cf * 16;
#pragma unroll
      for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) U[ks][i1][i2] = ug[lo[ks] + i1 * D1N + i2 * D2N];
    };
    auto load_plane = [&](int cf, int i2, double (&V)[KSX][N1]) __attribute__((always_inline)) {
      int lo[KSX];
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) lo[ks] = opq(lown[ks]) + cf * 16 + i2 * D2N;
#pragma unroll
      for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) V[ks][i1] = ug[lo[ks] + i1 * D1N];
    };
    // neighbour values across the x facets at the facet nodes (i1, i2) of a plane (lane group 0: facet 0, group 1: facet 1)
    auto xtr_plane = [&](int cf, int cg, int i2, double (&t)[N1]) __attribute__((always_inline)) {
      const double* pb = opqp(pxl) + (ghx ? cg : cf * 16) + i2 * stx2;
#pragma unroll
      for (int i1 = 0; i1 < N1; ++i1) t[i1] = pb[i1 * stx1];
    };
    // ... across the y facets at the facet nodes (i0, i2) of a plane
    auto ytr_plane = [&](int cf, int cg, int i2, double (&t0)[KSX], double (&t1)[KSX]) __attribute__((always_inline)) {
      const int c0 = gh[2] ? cg : cf * 16, c1 = gh[3] ? cg : cf * 16;
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) {
        t0[ks] = opqp(pf[2][ks])[c0 + i2 * st[2]];
        t1[ks] = opqp(pf[3][ks])[c1 + i2 * st[3]];
      }
    };
    // ... across the z facets at all their facet nodes (i0, i1)
    auto ztr_full = [&](int cf, int cg, double (&t0)[KSX][N1], double (&t1)[KSX][N1]) __attribute__((always_inline)) {
      const int c0 = gh[4] ? cg : cf * 16, c1 = gh[5] ? cg : cf * 16;
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) {
        const double* p0 = opqp(pf[4][ks]) + c0;
        const double* p1 = opqp(pf[5][ks]) + c1;
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) {
          t0[ks][i1] = p0[i1 * st[4]];
          t1[ks][i1] = p1[i1 * st[5]];
        }
      }
    };
    // ---- the line operators on a plane: o += line_k(V)
    auto X_plane = [&](double (&o)[KSX][N1], const double (&V)[KSX][N1], const double (&t)[N1]) __attribute__((always_inline)) {
#pragma unroll
      for (int i1 = 0; i1 < N1; ++i1) {
        const double tvs = sgx * t[i1];
#pragma unroll
        for (int kp = 0; kp < KSX; ++kp) {
          double r = o[kp][i1];
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) r = HXM_MFMA4(ax[kp][ks], V[ks][i1], r);
          o[kp][i1] = HXM_MFMA4(at[kp], tvs, r);
        }
      }
    };
    auto Y_plane = [&](double (&o)[KSX][N1], const double (&V)[KSX][N1], const double (&t0)[KSX], const double (&t1)[KSX]) __attribute__((always_inline)) {
#pragma unroll
      for (int ap = 0; ap < N1; ++ap) {
#pragma unroll
        for (int a = 0; a < N1; ++a) {
          const double cE = E(1, ap, a);
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) o[ks][ap] += cE * V[ks][a];
        }
        const double l0 = LW(1, 0, ap), l1 = LW(1, 1, ap);
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) o[ks][ap] += l0 * (sg[2] * t0[ks]) + l1 * (sg[3] * t1[ks]);
      }
    };
    // plane i2 of line_z(U): every line along z has one node in the plane
    auto Z_plane = [&](double (&o)[KSX][N1], const double (&U)[KSX][N1][N1], int i2, const double (&t0)[KSX][N1],
                       const double (&t1)[KSX][N1]) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < N1; ++a) {
        const double cE = E(2, i2, a);
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) o[ks][i1] += cE * U[ks][i1][a];
      }
      const double l0 = LW(2, 0, i2), l1 = LW(2, 1, i2);
#pragma unroll
      for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) o[ks][i1] += l0 * (sg[4] * t0[ks][i1]) + l1 * (sg[5] * t1[ks][i1]);
    };
    auto clear_plane = [&](double (&o)[KSX][N1]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) o[ks][i1] = 0.0;
    };
    // the arithmetic that produced v is complete HERE (the optimiser otherwise sinks it to its first use, past the
    // requests that follow, and everything requested meanwhile stays live: kernels_lane.hip hex_stage)
    auto pin_plane = [&](double (&v)[KSX][N1]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) asm volatile("" : "+v"(v[ks][i1]));
    };
#define HXM_FENCE() __builtin_amdgcn_sched_barrier(0)
    // ---- results: plane i2 of component `cmp` of the output field (nco components per node)
    // (a fused G stage has no second operand: out = c_self out + c_new rhs, stages.cpp - pa is then left alone)
    auto old_plane = [&](int nco, int cmp, int i2, double (&po)[KSX][N1], double (&pa)[KSX][N1]) __attribute__((always_inline)) {
      const long gbase = (g * (long)ND) * nco * 16;
      int lo[KSX];
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) lo[ks] = opq(i0c[ks] * 16) * nco + w + cmp * 16 + i2 * (N1 * N1) * nco * 16;
#pragma unroll
      for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
          const int o = lo[ks] + i1 * N1 * nco * 16;
          if (!(KIND == 0 && MODE == 2)) po[ks][i1] = HXM_LDS(&out[gbase + o]);     // (F, MODE 2: no self term)
          if (KIND == 0) pa[ks][i1] = HXM_LDS(&aux[gbase + o]);
        }
    };
    // one branch around all of a k-step's stores (a branch per store would end the scheduling region at every store)
    auto store_plane = [&](int nco, int cmp, int i2, const double (&v)[KSX][N1]) __attribute__((always_inline)) {
      double* const og = out + (g * (long)ND) * nco * 16;
      if (active) {
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks)
          if (ks == 0 || i0ok[ks]) {
            const int lo = opq(i0c[ks] * 16) * nco + w + cmp * 16 + i2 * (N1 * N1) * nco * 16;
#pragma unroll
            for (int i1 = 0; i1 < N1; ++i1) HXM_ST(&og[lo + i1 * N1 * nco * 16], v[ks][i1]);
          }
      }
    };
    auto cix = [](int i, int j) { return (SYM && i > j) ? j * 3 + i : i * 3 + j; };

    if (KIND == 0) {
      // ---- F: uh_i = line_x(T_i0) + line_y(T_i1) + line_z(T_i2) - sponge (elastic.py:204-209), one result component at
      //      a time (a rolled loop: the code of an item stays small).  T_i2 is held whole; T_i0 and T_i1 stream through
      //      plane by plane, the next plane requested while this one is worked on.
      // sponge of this lane's cube: a sigma that is one value on all its nodes is applied as sigma u_abs at the node; a varying
      // one through B_e u_abs, computed before the stage by a launch of its own (StageArgs::sponge_sigma / sponge_pre; a
      // matrix loop here - 64 or 125 columns per value - would hold the wave for every cube at the edge of a sponge strip).
      // One load per value either way: another base, stride and factor.
      double sig = 0.0;
      int sslot = -1;
      if (A.sponge_sigma != nullptr && active) {
        sig = A.sponge_sigma[e];
        if (sig != sig) sslot = A.sponge_slot[e];
      }
      double cs = A.c_self, ca = A.c_aux, cn = A.c_new;
      if (MODE == 1 && A.rho2 != nullptr) {   // per-cell density (kernels.hpp)
        cs = A.rho2[2 * e];
        ca *= A.rho2[2 * e + 1];
        cn *= A.rho2[2 * e + 1];
      }
      // in the fused stages u_abs IS one of the combine's operands (`out` in stage U1, `aux` in stage UTEMP): a constant
      // sigma then changes that operand's coefficient - no load, no arithmetic of its own
      if (MODE >= 1 && sslot < 0 && sig != 0.0) {
        if (MODE == 1 && A.uabs == A.out) {
          cs -= cn * sig;
          sig = 0.0;
        } else if (A.uabs == A.aux) {
          ca -= cn * sig;
          sig = 0.0;
        }
      }
      const bool any_sponge = A.sponge_sigma != nullptr && __any(sig != 0.0);     // (NaN != 0: the lanes with a matrix count)
      const double* sp_pb = sslot >= 0 ? reinterpret_cast<const double*>(A.sponge_pre) + (long)sslot * ND * 3
                                       : A.uabs + (g * (long)ND) * 3 * 16 + w;
      const int sp_es = sslot >= 0 ? 1 : 16;
      const double sp_sc = sslot >= 0 ? 1.0 : sig;
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        const int c0 = cix(i, 0), c1 = cix(i, 1), c2 = cix(i, 2);
        double Uz[KSX][N1][N1], tz0[KSX][N1], tz1[KSX][N1];
        double Va[2][KSX][N1], Vb[2][KSX][N1], xt[2][N1], y0[2][KSX], y1[2][KSX], po[KSX][N1], pa[KSX][N1];
        auto request = [&](int i2, int s2) __attribute__((always_inline)) {
          xtr_plane(c0, i, i2, xt[s2]);
          load_plane(c0, i2, Va[s2]);
          ytr_plane(c1, i, i2, y0[s2], y1[s2]);
          load_plane(c1, i2, Vb[s2]);
        };
        constexpr bool AHEAD = !(MODE >= 1 && P >= 4);   // (the fused F stages at degree 4: no room for a second set of plane operands)
        HXM_FENCE();
        ztr_full(c2, i, tz0, tz1);
        load_full(c2, Uz);
        if (AHEAD) request(0, 0);
        HXM_FENCE();
#pragma unroll
        for (int i2 = 0; i2 < N1; ++i2) {
          const int s2 = AHEAD ? (i2 & 1) : 0;
          if (AHEAD) {
            if (i2 + 1 < N1) request(i2 + 1, s2 ^ 1);
          } else {
            request(i2, 0);
          }
          if (MODE >= 1) old_plane(3, i, i2, po, pa);     // (behind the plane operands: needed last)
          HXM_FENCE();
          double o[KSX][N1];
          clear_plane(o);
          Z_plane(o, Uz, i2, tz0, tz1);
          X_plane(o, Va[s2], xt[s2]);
          Y_plane(o, Vb[s2], y0[s2], y1[s2]);
          if (any_sponge) {
            if (sig != 0.0) {
              double sv[KSX][N1];
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
                for (int ks = 0; ks < KSX; ++ks) sv[ks][i1] = sp_pb[((i0c[ks] + N1 * (i1 + N1 * i2)) * 3 + i) * sp_es];
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
                for (int ks = 0; ks < KSX; ++ks) o[ks][i1] -= sp_sc * sv[ks][i1];
            }
          }
          if (MODE >= 1) {     // (MODE 2, stage UTEMP: out = c_aux aux + c_new rhs, stages.cpp)
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1) o[ks][i1] = MODE == 1 ? cs * po[ks][i1] + ca * pa[ks][i1] + cn * o[ks][i1] : ca * pa[ks][i1] + cn * o[ks][i1];
          }
          pin_plane(o);
          HXM_FENCE();
          store_plane(3, i, i2, o);
          HXM_FENCE();
        }
      }
    } else {
      // ---- G: W_ik = line_k(u_i); sh_ii = 2 mu W_ii + lam tr W, sh_ij = mu (W_ij + W_ji) (elastic.py:211-219)
      const double lam = A.per_cell ? A.lam[e] : A.lam0;
      const double mu = A.per_cell ? A.mu[e] : A.mu0;
      const double cs = A.c_self, cn = A.c_new;
      auto emit = [&](int cmp, int i2, double (&v)[KSX][N1], const double (&po)[KSX][N1], const double (&pa)[KSX][N1]) __attribute__((always_inline)) {
        if (MODE == 1) {
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
            for (int i1 = 0; i1 < N1; ++i1) v[ks][i1] = cs * po[ks][i1] + cn * v[ks][i1];
        }
        pin_plane(v);
      };
      // -- phase A: u_2 held whole (its lines along z give W_22); u_0 and u_1 stream through plane by plane and give W_00,
      //    W_11 and the pair (0,1) on the way: sh_00, sh_11, sh_22, sh_01 of every plane leave together
      double U2[U2_LDS ? 1 : KSX][U2_LDS ? 1 : N1][U2_LDS ? 1 : N1];
      double* const su2 = sU2 + (U2_LDS ? (size_t)wave * ND * 16 + w : 0);
      // u_2 at the lane's node (ks, i1, a): a register, or the wave's LDS copy (a padded row reads a valid slot)
      auto u2 = [&](int ks, int i1, int a) __attribute__((always_inline)) -> double {
        if constexpr (U2_LDS)
          return su2[(i0c[ks] + N1 * (i1 + N1 * a)) * 16];
        else
          return U2[U2_LDS ? 0 : ks][U2_LDS ? 0 : i1][U2_LDS ? 0 : a];
      };
      auto Z_plane_u2 = [&](double (&o)[KSX][N1], int i2, const double (&t0)[KSX][N1], const double (&t1)[KSX][N1]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < N1; ++a) {
          const double cE = E(2, i2, a);
#pragma unroll
          for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) o[ks][i1] += cE * u2(ks, i1, a);
        }
        const double l0 = LW(2, 0, i2), l1 = LW(2, 1, i2);
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) o[ks][i1] += l0 * (sg[4] * t0[ks][i1]) + l1 * (sg[5] * t1[ks][i1]);
      };
      {
        double tz0[KSX][N1], tz1[KSX][N1];
        double V0[2][KSX][N1], V1[2][KSX][N1], x0[2][N1], x1[2][N1], y00[2][KSX], y01[2][KSX], y10[2][KSX], y11[2][KSX];
        double po[4][KSX][N1], pa[4][KSX][N1];
        constexpr int OC[4] = {0, 4, 8, 1};
        constexpr bool AHEAD = true;      // (the next plane's operands are requested while this plane is worked on)
        auto request = [&](int i2, int s2) __attribute__((always_inline)) {
          xtr_plane(0, 0, i2, x0[s2]);
          xtr_plane(1, 1, i2, x1[s2]);
          load_plane(0, i2, V0[s2]);
          load_plane(1, i2, V1[s2]);
          ytr_plane(0, 0, i2, y00[s2], y01[s2]);
          ytr_plane(1, 1, i2, y10[s2], y11[s2]);
        };
        HXM_FENCE();
        ztr_full(2, 2, tz0, tz1);
        if constexpr (U2_LDS) {
          double T[KSX][N1][N1];
          load_full(2, T);
          if (AHEAD) request(0, 0);
          HXM_FENCE();
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks)
            if (ks == 0 || i0ok[ks]) {
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
                for (int a = 0; a < N1; ++a) su2[(i0c[ks] + N1 * (i1 + N1 * a)) * 16] = T[ks][i1][a];
            }
        } else {
          double (&Ur)[KSX][N1][N1] = reinterpret_cast<double (&)[KSX][N1][N1]>(U2);
          load_full(2, Ur);
          if (AHEAD) request(0, 0);
        }
        HXM_FENCE();
#pragma unroll
        for (int i2 = 0; i2 < N1; ++i2) {
          const int s2 = AHEAD ? (i2 & 1) : 0;
          if (AHEAD) {
            if (i2 + 1 < N1) request(i2 + 1, s2 ^ 1);
          } else {
            request(i2, 0);
          }
          if (MODE == 1) {      // old values of two of the four results: behind the next plane's operands (needed last) ...
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) old_plane(9, OC[c4], i2, po[c4], pa[c4]);
          }
          HXM_FENCE();
          double w00[KSX][N1], w11[KSX][N1], w22[KSX][N1], w01[KSX][N1];
          clear_plane(w00);
          clear_plane(w11);
          clear_plane(w22);
          clear_plane(w01);
          Z_plane_u2(w22, i2, tz0, tz1);
          X_plane(w00, V0[s2], x0[s2]);
          Y_plane(w11, V1[s2], y10[s2], y11[s2]);
          Y_plane(w01, V0[s2], y00[s2], y01[s2]);
          X_plane(w01, V1[s2], x1[s2]);
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
            for (int i1 = 0; i1 < N1; ++i1) {
              const double tr = lam * (w00[ks][i1] + w11[ks][i1] + w22[ks][i1]);
              w00[ks][i1] = 2.0 * mu * w00[ks][i1] + tr;
              w11[ks][i1] = 2.0 * mu * w11[ks][i1] + tr;
              w22[ks][i1] = 2.0 * mu * w22[ks][i1] + tr;
              w01[ks][i1] = mu * w01[ks][i1];
            }
          double w10[SYM ? 1 : KSX][SYM ? 1 : N1];      // the mirror entry (1,0) (asymmetric user data, rare): own old values
          if (!SYM) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1) w10[SYM ? 0 : ks][SYM ? 0 : i1] = w01[ks][i1];
          }
          if (MODE == 1) {      // ... and of the other two once this plane's operands have left the registers
            pin_plane(w00);
            pin_plane(w11);
            pin_plane(w22);
            pin_plane(w01);
            HXM_FENCE();
#pragma unroll
            for (int c4 = 2; c4 < 4; ++c4) old_plane(9, OC[c4], i2, po[c4], pa[c4]);
            HXM_FENCE();
          }
          emit(0, i2, w00, po[0], pa[0]);
          emit(4, i2, w11, po[1], pa[1]);
          emit(8, i2, w22, po[2], pa[2]);
          emit(1, i2, w01, po[3], pa[3]);
          if constexpr (!SYM) {
            if (MODE == 1) {      // (not requested ahead)
              double qo[KSX][N1], qa[KSX][N1];
              old_plane(9, 3, i2, qo, qa);
#pragma unroll
              for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
                for (int i1 = 0; i1 < N1; ++i1) w10[SYM ? 0 : ks][SYM ? 0 : i1] = cs * qo[ks][i1] + cn * w10[SYM ? 0 : ks][SYM ? 0 : i1];
            }
            HXM_FENCE();
            double (&m10)[KSX][N1] = reinterpret_cast<double (&)[KSX][N1]>(w10);
            pin_plane(m10);
            store_plane(9, 3, i2, m10);
          }
          HXM_FENCE();
          store_plane(9, 0, i2, w00);
          store_plane(9, 4, i2, w11);
          store_plane(9, 8, i2, w22);
          store_plane(9, 1, i2, w01);
          HXM_FENCE();
        }
      }
      // -- phases B1, B2: the pairs (0,2) and (1,2) = line_z of u_0 / u_1 (held whole, one after the other) + line_x /
      //    line_y of u_2 (still held): plane by plane out of registers, only old values and traces come from memory
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double Ui[KSX][N1][N1], tz0[KSX][N1], tz1[KSX][N1];
        double xt[2][N1], y0[2][KSX], y1[2][KSX], po[KSX][N1], pa[KSX][N1];
        const int cmp = i * 3 + 2;
        auto request = [&](int i2, int s2) __attribute__((always_inline)) {
          if (i == 0)
            xtr_plane(2, 2, i2, xt[s2]);
          else
            ytr_plane(2, 2, i2, y0[s2], y1[s2]);
        };
        HXM_FENCE();
        ztr_full(i, i, tz0, tz1);
        load_full(i, Ui);
        request(0, 0);
        HXM_FENCE();
#pragma unroll
        for (int i2 = 0; i2 < N1; ++i2) {
          const int s2 = i2 & 1;
          if (i2 + 1 < N1) request(i2 + 1, s2 ^ 1);
          if (MODE == 1) old_plane(9, cmp, i2, po, pa);
          HXM_FENCE();
          double pr[KSX][N1], V2[KSX][N1];
          clear_plane(pr);
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
            for (int i1 = 0; i1 < N1; ++i1) V2[ks][i1] = u2(ks, i1, i2);
          Z_plane(pr, Ui, i2, tz0, tz1);                  // W_i2
          if (i == 0)
            X_plane(pr, V2, xt[s2]);                      // W_20
          else
            Y_plane(pr, V2, y0[s2], y1[s2]);              // W_21
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
            for (int i1 = 0; i1 < N1; ++i1) pr[ks][i1] *= mu;
          if (!SYM) {
            double m2[KSX][N1];
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1) m2[ks][i1] = pr[ks][i1];
            if (MODE == 1) {
              double qo[KSX][N1], qa[KSX][N1];
              old_plane(9, 6 + i, i2, qo, qa);
#pragma unroll
              for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
                for (int i1 = 0; i1 < N1; ++i1) m2[ks][i1] = cs * qo[ks][i1] + cn * m2[ks][i1];
            }
            pin_plane(m2);
            HXM_FENCE();
            store_plane(9, 6 + i, i2, m2);
          }
          emit(cmp, i2, pr, po, pa);
          HXM_FENCE();
          store_plane(9, cmp, i2, pr);
          HXM_FENCE();
        }
      }
    }
  }
}

template <int P>
static int launch_hexm_p(int kind, const StageArgs& a, long nitems, hipStream_t s) {
  // persistent grid: HXW<P>::WPE blocks of four waves per CU, a multiple of 8 (one item range per XCD label)
  long blocks = (nitems + 3) / 4;
  const long cap = a.grid_blocks > 0 ? a.grid_blocks : 256L * HXW<P>::WPE;
  if (blocks > cap) blocks = cap;
  blocks = (blocks + 7) / 8 * 8;
  const dim3 grid((unsigned)blocks), block(256);
#define SG_HEXM_LAUNCH(K, M)                                               \
  do {                                                                     \
    if (a.sym)                                                             \
      SG_LAUNCH((hexm_stage<P, K, M, 1>), grid, block, s, a, a);           \
    else                                                                   \
      SG_LAUNCH((hexm_stage<P, K, M, 0>), grid, block, s, a, a);           \
  } while (0)
  if (kind == 0) {
    if (a.mode == 0)
      SG_HEXM_LAUNCH(0, 0);
    else if (a.mode == 2)
      SG_HEXM_LAUNCH(0, 2);
    else
      SG_HEXM_LAUNCH(0, 1);
  } else {
    if (a.mode == 0)
      SG_HEXM_LAUNCH(1, 0);
    else
      SG_HEXM_LAUNCH(1, 1);
  }
#undef SG_HEXM_LAUNCH
  return (int)hipGetLastError();
}

int hexm_blocks_per_cu(int P) { return P <= 3 ? 2 : 1; }

int launch_stage_hexm(int kind, int P, const StageArgs& a, long nitems, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (P == 3) return launch_hexm_p<3>(kind, a, nitems, s);
  if (P == 4) return launch_hexm_p<4>(kind, a, nitems, s);
  return -1;
}

}  // namespace sg
