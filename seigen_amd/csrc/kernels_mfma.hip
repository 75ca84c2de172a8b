// MFMA stage kernels for 3-D, degree >= 3 (gfx950, FP64).
//
// Layout (mesh_tables.hpp): 16 cubes' cells of one simplex class are interleaved
// so that the 16 values of one (node, component) form one 128-byte line.  One
// wavefront owns one such 16-cell group at a time.  The dense element-local
// contractions of seigen/elastic.py:204-219 (+ the element mass inverse,
// :358-367) become v_mfma_f64_16x16x4_f64 products
//
//     [reference operator tile 16x4]  x  [4 nodes x 16 cells]  ->  [16 rows x 16 cells]
//
// with the cells on the MFMA N (column) axis: lane l holds column (l & 15) =
// cell, and k / row-quad (l >> 4).  The B operand rows and the accumulator
// rows are exactly the 128-byte lines of the layout, so operands are loaded
// from and results stored to HBM/L2 directly in fragment form - no LDS staging
// of cell data.  The operator tiles live in LDS in lane order (mfma_tables.cpp).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sg {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int P>
struct MG {
  static constexpr int ND = (P + 1) * (P + 2) * (P + 3) / 6;
  static constexpr int NF = (P + 1) * (P + 2) / 2;
  static constexpr int KS = (ND + 3) / 4;
  static constexpr int KSF = (NF + 3) / 4;
  static constexpr int MTL = (ND + 15) / 16;
  static constexpr int S4 = (ND + 3) / 4;
  static constexpr int MTG = (12 * S4 + 15) / 16;
  static constexpr int NFRAG_F = MTL * 3 * KS;
  static constexpr int NFRAG_G = MTG * KS;
  static constexpr int NFRAG_L = 4 * MTL * KSF;
  static constexpr int NCLS = 6;
  static constexpr int GW = 16;
};

struct LaneGeo {
  long c;      // linear cube index of this lane's cell
  int cc[3];   // cube coordinates
  bool valid;  // a real cube (not layout padding)
  bool active; // valid and inside the launch's region
};

__device__ __forceinline__ LaneGeo lane_geo(const MeshDev& md, const StageArgs& A, long g, int w) {
  LaneGeo L;
  L.c = g * 16 + w;
  L.valid = L.c < md.ncube;
  long cl = L.valid ? L.c : 0;
  L.cc[0] = (int)(cl % md.n[0]);
  long t = cl / md.n[0];
  L.cc[1] = (int)(t % md.n[1]);
  L.cc[2] = (int)(t / md.n[1]);
  bool in = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) in = in && (L.cc[a] >= A.box_o[a]) && (L.cc[a] < A.box_o[a] + A.box_n[a]);
  L.active = L.valid && in;
  return L;
}

// Where this lane finds its neighbour's trace across facet f.
struct NbrRef {
  const double* p;  // base pointer of the neighbour cell (or ghost slot)
  int cstride;      // stride between (node, comp) entries: 16 in a field, 1 in a packed ghost buffer
  bool ghost;
  bool physical;    // domain boundary: no neighbour
};

template <int ND, int NF, int NC>
__device__ __forceinline__ NbrRef nbr_ref(const MeshDev* md, const StageArgs& A, const LaneGeo& L, long g, int k, int f,
                                          int w, const double* own_base) {
  NbrRef R;
  R.p = own_base;
  R.cstride = 16;
  R.ghost = false;
  R.physical = false;
  const int axis = md->nb_axis[k][f];
  const int kn = md->nb_cls[k][f];
  if (axis < 0) {
    R.p = A.in + ((g * 6 + kn) * (long)ND) * NC * 16 + w;
    return R;
  }
  const int dir = md->nb_dir[k][f];
  const int cn = L.cc[axis] + dir;
  if (!L.valid) {
    R.physical = true;
    return R;
  }
  if (cn >= 0 && cn < md->n[axis]) {
    long stride = (axis == 0) ? 1 : (axis == 1) ? md->n[0] : (long)md->n[0] * md->n[1];
    long nc = L.c + dir * stride;
    R.p = A.in + (((nc >> 4) * 6 + kn) * (long)ND) * NC * 16 + (nc & 15);
    return R;
  }
  const int side = 2 * axis + (dir > 0 ? 1 : 0);
  if (md->has_nbr[side]) {
    long c2 = (axis == 0) ? (L.cc[1] + (long)md->n[1] * L.cc[2])
                          : (axis == 1) ? (L.cc[0] + (long)md->n[0] * L.cc[2]) : (L.cc[0] + (long)md->n[0] * L.cc[1]);
    long slot = c2 * md->halo_per_cube + md->face_ord[kn][md->nb_face[k][f]];
    R.p = A.ghost[side] + slot * NF * NC;
    R.cstride = 1;
    R.ghost = true;
    return R;
  }
  R.physical = true;
  return R;
}

// XCD-aware work split: blocks with equal blockIdx % 8 share an XCD (and its L2), so each
// label gets one contiguous range of items; a different placement only changes speed.
struct ItemRange {
  long lo, hi, step;
};
__device__ __forceinline__ ItemRange item_range(long nitems, int wave) {
  const long nblk = gridDim.x;
  const long xcd = blockIdx.x % 8;
  const long slot = blockIdx.x / 8;
  const long blocks_here = (nblk - xcd + 7) / 8;
  const long ipx = (nitems + 7) / 8;
  ItemRange r;
  r.lo = xcd * ipx + slot * 4 + wave;
  r.hi = (xcd + 1) * ipx < nitems ? (xcd + 1) * ipx : nitems;
  r.step = blocks_here * 4;
  return r;
}

// --------------------------------------------------------------------------------------------
//  G: sh_ij = lam d_ij W_kk + mu (W_ij + W_ji),  W_ik = -Jinv_rk (D_r u_i) + sum_f (c n)_k L_f u^_i
// --------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256, 2) void mfma_stage_G(StageArgs A) {
  using M = MG<P>;
  constexpr int ND = M::ND, NF = M::NF, KS = M::KS, KSF = M::KSF, MTL = M::MTL, S4 = M::S4, MTG = M::MTG;
  __shared__ double sAV[M::NFRAG_G * 64];
  __shared__ double sAL[M::NFRAG_L * 64];
  __shared__ MeshDev sMd;
  for (int i = threadIdx.x; i < M::NFRAG_G * 64; i += 256) sAV[i] = A.fragV[i];
  for (int i = threadIdx.x; i < M::NFRAG_L * 64; i += 256) sAL[i] = A.fragL[i];
  {
    const int* src = reinterpret_cast<const int*>(A.md);
    int* dst = reinterpret_cast<int*>(&sMd);
    for (int i = threadIdx.x; i < (int)(sizeof(MeshDev) / sizeof(int)); i += 256) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, w = lane & 15;
  const MeshDev* md = A.md;  // uniform reads of class constants go through the scalar cache
  const long ngroups = sMd.ncube_pad >> 4;
  const ItemRange ir = item_range(ngroups * 6, wave);

  for (long item = ir.lo; item < ir.hi; item += ir.step) {
    const long g = item / 6;
    const int k = (int)(item - g * 6);
    const LaneGeo L = lane_geo(sMd, A, g, w);
    if (!__any(L.active)) continue;
    const double* own = A.in + ((g * 6 + k) * (long)ND) * 3 * 16 + w;

    double Sd[3][S4], So[3][S4];  // diagonal W_ii and symmetric sums W_ij + W_ji ((0,1),(0,2),(1,2))
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int m = 0; m < S4; ++m) Sd[i][m] = So[i][m] = 0.0;

    // ---- volume: B fragments = the cells' own nodal values, straight from memory
    {
      double bf[KS][3];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int b = 4 * ks + q;
#pragma unroll
        for (int i = 0; i < 3; ++i) bf[ks][i] = (b < ND) ? own[(b * 3 + i) * 16] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < MTG; ++t) {
        d4 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = d4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const double a = sAV[(t * KS + ks) * 64 + lane];
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bf[ks][i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int rho4 = 4 * t + reg;
          const int r = rho4 / S4, m = rho4 % S4;
          if (r < 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const double v = acc[i][reg];
#pragma unroll
              for (int kk = 0; kk < 3; ++kk) {
                const double wv = -md->Jinv[k][r][kk] * v;
                if (i == kk)
                  Sd[i][m] += wv;
                else
                  So[i + kk - 1][m] += wv;
              }
            }
          }
        }
      }
    }

    // ---- facet lifts
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const NbrRef R = nbr_ref<ND, NF, 3>(md, A, L, g, k, f, w, own);
      double fl[KSF][3];
#pragma unroll
      for (int ks = 0; ks < KSF; ++ks) {
        const int b = 4 * ks + q;
        const int bb = b < NF ? b : 0;
        const int on = sMd.fnode[f][bb];
        const int nn = R.ghost ? sMd.nb_fnode[k][f][bb] : (R.physical ? on : sMd.nb_node[k][f][bb]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          // u^ = avg(u) on interior facets, own trace on the boundary (elastic.py:213-216)
          const double ov = own[(on * 3 + i) * 16];
          const double nv = R.p[(nn * 3 + i) * R.cstride];
          fl[ks][i] = (b < NF) ? 0.5 * (ov + nv) : 0.0;
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        d4 tmp[MTL];
#pragma unroll
        for (int t = 0; t < MTL; ++t) tmp[t] = d4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
          for (int t = 0; t < MTL; ++t)
            tmp[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(sAL[((f * MTL + t) * KSF + ks) * 64 + lane], fl[ks][i], tmp[t], 0,
                                                          0, 0);
#pragma unroll
        for (int t = 0; t < MTL; ++t)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int m = 4 * t + reg;
            if (m < S4) {
              const double v = tmp[t][reg];
#pragma unroll
              for (int kk = 0; kk < 3; ++kk) {
                const double wv = md->cn[k][f][kk] * v;
                if (i == kk)
                  Sd[i][m] += wv;
                else
                  So[i + kk - 1][m] += wv;
              }
            }
          }
      }
    }

    // ---- stress and epilogue
    if (L.active) {
      const long e = L.c * 6 + k;
      const double lam = A.per_cell ? A.lam[e] : A.lam0;
      const double mu = A.per_cell ? A.mu[e] : A.mu0;
      const long obase = ((g * 6 + k) * (long)ND) * 9 * 16 + w;
#pragma unroll
      for (int m = 0; m < S4; ++m) {
        const int a = 4 * m + q;
        if (a < ND) {
          const double tr = lam * (Sd[0][m] + Sd[1][m] + Sd[2][m]);
          double s[9];
          s[0] = 2.0 * mu * Sd[0][m] + tr;
          s[4] = 2.0 * mu * Sd[1][m] + tr;
          s[8] = 2.0 * mu * Sd[2][m] + tr;
          s[1] = s[3] = mu * So[0][m];
          s[2] = s[6] = mu * So[1][m];
          s[5] = s[7] = mu * So[2][m];
#pragma unroll
          for (int ij = 0; ij < 9; ++ij) {
            const long o = obase + (a * 9 + ij) * 16;
            if (A.mode == 0)
              A.out[o] = s[ij];
            else
              A.out[o] = A.c_self * A.out[o] + A.c_aux * A.aux[o] + A.c_new * s[ij];
          }
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
//  F: uh_i = -sum_r D_r (Jinv_rj T_ij) + sum_f L_f [ (c n)_j {T_ij} ] - sponge
// --------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256, 2) void mfma_stage_F(StageArgs A) {
  using M = MG<P>;
  constexpr int ND = M::ND, NF = M::NF, KS = M::KS, KSF = M::KSF, MTL = M::MTL;
  __shared__ double sAV[M::NFRAG_F * 64];
  __shared__ double sAL[M::NFRAG_L * 64];
  __shared__ MeshDev sMd;
  for (int i = threadIdx.x; i < M::NFRAG_F * 64; i += 256) sAV[i] = A.fragV[i];
  for (int i = threadIdx.x; i < M::NFRAG_L * 64; i += 256) sAL[i] = A.fragL[i];
  {
    const int* src = reinterpret_cast<const int*>(A.md);
    int* dst = reinterpret_cast<int*>(&sMd);
    for (int i = threadIdx.x; i < (int)(sizeof(MeshDev) / sizeof(int)); i += 256) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, w = lane & 15;
  const MeshDev* md = A.md;
  const long ngroups = sMd.ncube_pad >> 4;
  const ItemRange ir = item_range(ngroups * 6, wave);

  for (long item = ir.lo; item < ir.hi; item += ir.step) {
    const long g = item / 6;
    const int k = (int)(item - g * 6);
    const LaneGeo L = lane_geo(sMd, A, g, w);
    if (!__any(L.active)) continue;
    const double* own = A.in + ((g * 6 + k) * (long)ND) * 9 * 16 + w;

    d4 acc[3][MTL];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int t = 0; t < MTL; ++t) acc[i][t] = d4{0, 0, 0, 0};

    // ---- volume: K runs over (r, node); B = T~_ir = Jinv_rj T_ij formed in registers
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int b = 4 * ks + q;
      double T[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) T[c] = (b < ND) ? own[(b * 9 + c) * 16] : 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double Tt[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
          Tt[i] = md->Jinv[k][r][0] * T[i * 3 + 0] + md->Jinv[k][r][1] * T[i * 3 + 1] + md->Jinv[k][r][2] * T[i * 3 + 2];
#pragma unroll
        for (int t = 0; t < MTL; ++t) {
          const double a = sAV[(t * 3 * KS + KS * r + ks) * 64 + lane];
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Tt[i], acc[i][t], 0, 0, 0);
        }
      }
    }

    // ---- facet lifts of (c n)_j {T_ij}; no ds term in f => zero flux on the boundary
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const NbrRef R = nbr_ref<ND, NF, 9>(md, A, L, g, k, f, w, own);
      const double pf = R.physical ? 0.0 : 0.5;
#pragma unroll
      for (int ks = 0; ks < KSF; ++ks) {
        const int b = 4 * ks + q;
        const int bb = b < NF ? b : 0;
        const int on = sMd.fnode[f][bb];
        const int nn = R.ghost ? sMd.nb_fnode[k][f][bb] : (R.physical ? on : sMd.nb_node[k][f][bb]);
        double fl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            s += md->cn[k][f][j] * (own[(on * 9 + i * 3 + j) * 16] + R.p[(nn * 9 + i * 3 + j) * R.cstride]);
          fl[i] = (b < NF) ? pf * s : 0.0;
        }
#pragma unroll
        for (int t = 0; t < MTL; ++t) {
          const double a = sAL[((f * MTL + t) * KSF + ks) * 64 + lane];
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fl[i], acc[i][t], 0, 0, 0);
        }
      }
    }

    // ---- sponge (rare): -sum_b B_e[a][b] u_abs[b][i] on the lanes whose cell carries sigma
    const long e = L.c * 6 + k;
    int slot = -1;
    if (A.sponge_slot != nullptr && L.active) slot = A.sponge_slot[e];
    const long ubase = ((g * 6 + k) * (long)ND) * 3 * 16 + w;
    if (__any(slot >= 0)) {
      if (slot >= 0) {
#pragma unroll
        for (int t = 0; t < MTL; ++t)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int a = 16 * t + 4 * reg + q;
            if (a < ND) {
              const double* B = A.sponge_B + ((long)slot * ND + a) * ND;
              double s0 = 0, s1 = 0, s2 = 0;
              for (int b = 0; b < ND; ++b) {
                const double bb = B[b];
                s0 += bb * A.uabs[ubase + (b * 3 + 0) * 16];
                s1 += bb * A.uabs[ubase + (b * 3 + 1) * 16];
                s2 += bb * A.uabs[ubase + (b * 3 + 2) * 16];
              }
              acc[0][t][reg] -= s0;
              acc[1][t][reg] -= s1;
              acc[2][t][reg] -= s2;
            }
          }
      }
      // in-place combine: all sponge reads of u (= out) must precede the writes below.  Lanes of
      // one cell sit in this wave only, and a wave executes in program order, so no barrier is needed.
    }

    if (L.active) {
#pragma unroll
      for (int t = 0; t < MTL; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int a = 16 * t + 4 * reg + q;
          if (a < ND) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const long o = ubase + (a * 3 + i) * 16;
              const double v = acc[i][t][reg];
              if (A.mode == 0)
                A.out[o] = v;
              else
                A.out[o] = A.c_self * A.out[o] + A.c_aux * A.aux[o] + A.c_new * v;
            }
          }
        }
    }
  }
}

template <int P>
static int launch_p(int kind, const StageArgs& a, hipStream_t s) {
  const long grid = 512;  // 2 blocks per CU, a multiple of 8 (one contiguous item range per XCD label)
  if (kind == 0)
    hipLaunchKernelGGL((mfma_stage_F<P>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((mfma_stage_G<P>), dim3((unsigned)grid), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

bool mfma_supported(int dim, int P) { return dim == 3 && (P == 3 || P == 4); }

int launch_stage_mfma(int kind, int P, const StageArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (P) {
    case 3: return launch_p<3>(kind, a, s);
    case 4: return launch_p<4>(kind, a, s);
  }
  return -1;
}

}  // namespace sg
