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
    const double* own = in + (g * (long)ND) * NC * 16 + w;
    const long e = valid ? c : 0;

    // where each facet's neighbour trace lives: the neighbour cube's cell, a packed remote trace ([cube on the side][facet
    // node][3]: the velocity, or T_i,axis of a stress), or - on the domain boundary - the own cell
    const double* np[6];
    bool gh[6], ph[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const int axis = f >> 1, dir = (f & 1) ? 1 : -1;
      const int cn = cc[axis] + dir;
      const int nax = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
      const bool inside = valid && cn >= 0 && cn < nax;
      const long stride = axis == 0 ? 1 : (axis == 1 ? (long)n0 : (long)n0 * n1);
      const long nc = inside ? c + dir * stride : (valid ? c : 0);
      np[f] = in + ((nc >> 4) * (long)ND) * NC * 16 + (nc & 15);
      gh[f] = false;
      ph[f] = !inside;
      if (!inside && valid && md->has_nbr[f]) {
        const long c2 = axis == 0 ? (cc[1] + (long)n1 * cc[2]) : (axis == 1 ? (cc[0] + (long)n0 * cc[2]) : (cc[0] + (long)n0 * cc[1]));
        np[f] = A.ghost[f] + c2 * NF * 3;
        gh[f] = true;
        ph[f] = false;
      }
    }
    // x facets: lane group 0 reads facet 0, lane group 1 facet 1 (the extra k-step of the x pass), groups 2 and 3 nothing
    const double* const npx = (q & 1) ? np[1] : np[0];
    const bool ghx = (q & 1) ? gh[1] : gh[0], phx = (q & 1) ? ph[1] : ph[0];
    const int x_own = (q & 1) ? P : 0, x_acr = (q & 1) ? 0 : P;

    // ---- operand access
    // component cf of the cell's own nodes -> U[ks][i1][i2]
    auto load_comp = [&](int cf, double (&U)[KSX][N1][N1]) __attribute__((always_inline)) {
#pragma unroll
      for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) U[ks][i1][i2] = own[((i0c[ks] + N1 * (i1 + N1 * i2)) * NC + cf) * 16];
    };
    // neighbour values across the two x facets at facet node (i1, i2), as the B operand of the extra k-step
    auto trace_x = [&](int i1, int i2, int cf, int cg) __attribute__((always_inline)) -> double {
      const int off = ghx ? (i1 + N1 * i2) * 3 + cg : (((phx ? x_own : x_acr) + N1 * (i1 + N1 * i2)) * NC + cf) * 16;
      double v = 0.0;
      if (q < 2) {
        v = npx[off];
        v = phx ? SGN * v : v;
      }
      return v;
    };
    // ... across a y facet (f = 2, 3) at facet node (i0, i2), across a z facet (f = 4, 5) at facet node (i0, i1): tt = the
    // second transverse index
    auto trace_yz = [&](int f, int ks, int tt, int cf, int cg) __attribute__((always_inline)) -> double {
      const int axis = f >> 1;
      const int fo = (f & 1) ? P : 0, fa = (f & 1) ? 0 : P;
      const int node_own = axis == 1 ? i0c[ks] + N1 * (fo + N1 * tt) : i0c[ks] + N1 * (tt + N1 * fo);
      const int node_acr = axis == 1 ? i0c[ks] + N1 * (fa + N1 * tt) : i0c[ks] + N1 * (tt + N1 * fa);
      const int off = gh[f] ? (i0c[ks] + N1 * tt) * 3 + cg : ((ph[f] ? node_own : node_acr) * NC + cf) * 16;
      const double v = np[f][off];
      return ph[f] ? SGN * v : v;
    };

    // ---- the three line passes: acc += line_k(U) with the traces of component (cf in a field, cg in a remote record)
    auto pass_x = [&](double (&acc)[KSX][N1][N1], const double (&U)[KSX][N1][N1], int cf, int cg) __attribute__((always_inline)) {
      double tv[N1][N1];
#pragma unroll
      for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) tv[i1][i2] = trace_x(i1, i2, cf, cg);
#pragma unroll
      for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int kp = 0; kp < KSX; ++kp) {
            double r = acc[kp][i1][i2];
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) r = HXM_MFMA4(ax[kp][ks], U[ks][i1][i2], r);
            acc[kp][i1][i2] = HXM_MFMA4(at[kp], tv[i1][i2], r);
          }
    };
    auto pass_y = [&](double (&acc)[KSX][N1][N1], const double (&U)[KSX][N1][N1], int cf, int cg) __attribute__((always_inline)) {
      double t0[KSX][N1], t1[KSX][N1];
#pragma unroll
      for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
          t0[ks][i2] = trace_yz(2, ks, i2, cf, cg);
          t1[ks][i2] = trace_yz(3, ks, i2, cf, cg);
        }
#pragma unroll
      for (int ap = 0; ap < N1; ++ap) {
#pragma unroll
        for (int a = 0; a < N1; ++a) {
          const double cE = E(1, ap, a);
#pragma unroll
          for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) acc[ks][ap][i2] += cE * U[ks][a][i2];
        }
        const double l0 = LW(1, 0, ap), l1 = LW(1, 1, ap);
#pragma unroll
        for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) acc[ks][ap][i2] += l0 * t0[ks][i2] + l1 * t1[ks][i2];
      }
    };
    auto pass_z = [&](double (&acc)[KSX][N1][N1], const double (&U)[KSX][N1][N1], int cf, int cg) __attribute__((always_inline)) {
      double t0[KSX][N1], t1[KSX][N1];
#pragma unroll
      for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
          t0[ks][i1] = trace_yz(4, ks, i1, cf, cg);
          t1[ks][i1] = trace_yz(5, ks, i1, cf, cg);
        }
#pragma unroll
      for (int ap = 0; ap < N1; ++ap) {
#pragma unroll
        for (int a = 0; a < N1; ++a) {
          const double cE = E(2, ap, a);
#pragma unroll
          for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) acc[ks][i1][ap] += cE * U[ks][i1][a];
        }
        const double l0 = LW(2, 0, ap), l1 = LW(2, 1, ap);
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) acc[ks][i1][ap] += l0 * t0[ks][i1] + l1 * t1[ks][i1];
      }
    };
    auto clear = [&](double (&acc)[KSX][N1][N1]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int i2 = 0; i2 < N1; ++i2) acc[ks][i1][i2] = 0.0;
    };
    // the arithmetic that produced v is complete HERE (the optimiser otherwise sinks it to its first use, past the
    // requests that follow: kernels_lane.hip hex_stage)
    auto pin = [&](double (&v)[KSX][N1][N1]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int i2 = 0; i2 < N1; ++i2) asm volatile("" : "+v"(v[ks][i1][i2]));
    };
    // results -> component `cmp` of the output field (NCO components per node); MODE 1: out = cs out + ca aux + cn v
    auto finish = [&](double (&v)[KSX][N1][N1], int nco, int cmp, double cs, double ca, double cn) __attribute__((always_inline)) {
      const long obase = (g * (long)ND) * nco * 16 + w;
      if (MODE == 1) {
        double po[KSX][N1][N1], pa[KSX][N1][N1];
#pragma unroll
        for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
          for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
              const long o = obase + ((long)(i0c[ks] + N1 * (i1 + N1 * i2)) * nco + cmp) * 16;
              po[ks][i1][i2] = HXM_LDS(&out[o]);
              pa[ks][i1][i2] = HXM_LDS(&aux[o]);
            }
#pragma unroll
        for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
          for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) v[ks][i1][i2] = cs * po[ks][i1][i2] + ca * pa[ks][i1][i2] + cn * v[ks][i1][i2];
        pin(v);
      }
#pragma unroll
      for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks)
            if (active && i0ok[ks]) HXM_ST(&out[obase + ((long)(i0c[ks] + N1 * (i1 + N1 * i2)) * nco + cmp) * 16], v[ks][i1][i2]);
    };
    auto cix = [](int i, int j) { return (SYM && i > j) ? j * 3 + i : i * 3 + j; };

    if (KIND == 0) {
      // ---- F: uh_i = sum_j line_j(T_ij) - sponge (elastic.py:204-209); one result component at a time
      int sslot = -1;
      if (A.sponge_slot != nullptr && active) sslot = A.sponge_slot[e];
      const bool any_sponge = A.sponge_slot != nullptr && __any(sslot >= 0);
      double cs = A.c_self, ca = A.c_aux, cn = A.c_new;
      if (MODE == 1 && A.rho2 != nullptr) {   // per-cell density (kernels.hpp)
        cs = A.rho2[2 * e];
        ca *= A.rho2[2 * e + 1];
        cn *= A.rho2[2 * e + 1];
      }
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        double acc[KSX][N1][N1];
        clear(acc);
        {
          double U[KSX][N1][N1];
          load_comp(cix(i, 0), U);
          pass_x(acc, U, cix(i, 0), i);
        }
        {
          double U[KSX][N1][N1];
          load_comp(cix(i, 1), U);
          pass_y(acc, U, cix(i, 1), i);
        }
        {
          double U[KSX][N1][N1];
          load_comp(cix(i, 2), U);
          pass_z(acc, U, cix(i, 2), i);
        }
        if (any_sponge) {
          // - sum_b B_e[a][b] u_abs[b][i] on the lanes whose cube carries sigma (rare: a layer of cells)
          if (sslot >= 0) {
            const double* B = A.sponge_B + (long)sslot * ND * ND;
            const double* ua = A.uabs + (g * (long)ND) * 3 * 16 + w;
#pragma unroll
            for (int i2 = 0; i2 < N1; ++i2)
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
                for (int ks = 0; ks < KSX; ++ks) {
                  const int a = i0c[ks] + N1 * (i1 + N1 * i2);
                  double s = 0.0;
                  for (int b = 0; b < ND; ++b) s += B[(long)a * ND + b] * ua[(b * 3 + i) * 16];
                  acc[ks][i1][i2] -= s;
                }
          }
        }
        finish(acc, 3, i, cs, ca, cn);
      }
    } else {
      // ---- G: W_ik = line_k(u_i); sh_ii = 2 mu W_ii + lam tr W, sh_ij = mu (W_ij + W_ji) (elastic.py:211-219)
      const double lam = A.per_cell ? A.lam[e] : A.lam0;
      const double mu = A.per_cell ? A.mu[e] : A.mu0;
      constexpr bool HOLD = P <= 3;     // the three velocity components stay in registers from sweep A to sweep B
      double UH[HOLD ? 3 : 1][KSX][N1][N1];
      {
        // sweep A: the diagonal
        double wd[3][KSX][N1][N1];
#pragma unroll
        for (int k = 0; k < 3; ++k) clear(wd[k]);
        if (HOLD) {
#pragma unroll
          for (int k = 0; k < 3; ++k) load_comp(k, UH[HOLD ? k : 0]);
          pass_x(wd[0], UH[0], 0, 0);
          pass_y(wd[1], UH[HOLD ? 1 : 0], 1, 1);
          pass_z(wd[2], UH[HOLD ? 2 : 0], 2, 2);
        } else {
          {
            double U[KSX][N1][N1];
            load_comp(0, U);
            pass_x(wd[0], U, 0, 0);
          }
          {
            double U[KSX][N1][N1];
            load_comp(1, U);
            pass_y(wd[1], U, 1, 1);
          }
          {
            double U[KSX][N1][N1];
            load_comp(2, U);
            pass_z(wd[2], U, 2, 2);
          }
        }
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
          for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
            for (int i2 = 0; i2 < N1; ++i2) {
              const double tr = lam * (wd[0][ks][i1][i2] + wd[1][ks][i1][i2] + wd[2][ks][i1][i2]);
#pragma unroll
              for (int k = 0; k < 3; ++k) wd[k][ks][i1][i2] = 2.0 * mu * wd[k][ks][i1][i2] + tr;
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) finish(wd[k], 9, 4 * k, A.c_self, A.c_aux, A.c_new);
      }
      {
        // sweep B: the pairs (0,1), (0,2), (1,2): component i feeds the two pairs it belongs to, along the other two axes
        double pr[3][KSX][N1][N1];
#pragma unroll
        for (int p = 0; p < 3; ++p) clear(pr[p]);
        if (HOLD) {
          pass_y(pr[0], UH[0], 0, 0);                  // W_01
          pass_z(pr[1], UH[0], 0, 0);                  // W_02
          pass_x(pr[0], UH[HOLD ? 1 : 0], 1, 1);       // W_10
          pass_z(pr[2], UH[HOLD ? 1 : 0], 1, 1);       // W_12
          pass_x(pr[1], UH[HOLD ? 2 : 0], 2, 2);       // W_20
          pass_y(pr[2], UH[HOLD ? 2 : 0], 2, 2);       // W_21
        } else {
          {
            double U[KSX][N1][N1];
            load_comp(0, U);
            pass_y(pr[0], U, 0, 0);
            pass_z(pr[1], U, 0, 0);
          }
          {
            double U[KSX][N1][N1];
            load_comp(1, U);
            pass_x(pr[0], U, 1, 1);
            pass_z(pr[2], U, 1, 1);
          }
          {
            double U[KSX][N1][N1];
            load_comp(2, U);
            pass_x(pr[1], U, 2, 2);
            pass_y(pr[2], U, 2, 2);
          }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const int i = p == 2 ? 1 : 0, j = p == 0 ? 1 : 2;
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
            for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
              for (int i2 = 0; i2 < N1; ++i2) pr[p][ks][i1][i2] *= mu;
          if (!SYM) {      // the mirror entry has its own old values (asymmetric user data, rare)
            double m2[KSX][N1][N1];
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
              for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
                for (int i2 = 0; i2 < N1; ++i2) m2[ks][i1][i2] = pr[p][ks][i1][i2];
            finish(m2, 9, j * 3 + i, A.c_self, A.c_aux, A.c_new);
          }
          finish(pr[p], 9, i * 3 + j, A.c_self, A.c_aux, A.c_new);
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
