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
// The nd % 16 rows left over by the 16-row tiles use v_mfma_f64_4x4x4_4b_f64 (MFMA4 below).
//
// Loads are software-pipelined by hand (PF k-steps ahead): hipcc otherwise issues
// each B operand right before the MFMA that consumes it and waits for it.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sg {

// The kernels are written once for the field type R: double (the parity and headline mode) or float
// (the second mode of SURVEY 8d: FP32 storage and arithmetic, 32 B per DoF-update).  What differs:
//   double: v_mfma_f64_16x16x4_f64 (C/D row = 4 reg + (lane >> 4)) for full 16-row tiles and
//           v_mfma_f64_4x4x4_4b_f64 for the nd % 16 rows left over;
//   float:  v_mfma_f32_16x16x4_f32 (C/D row = 4 (lane >> 4) + reg) for every row tile, the last one
//           zero-padded (gfx950 has no 4-row f32 shape with K = 4).  The operator tiles of the float
//           tables have their rows permuted (mfma_tables.cpp) so that register `reg` of lane group q
//           holds node 16 t + 4 reg + q in both cases and everything below is type-independent.
template <typename R>
struct RT;
template <>
struct RT<double> {
  typedef double v4 __attribute__((ext_vector_type(4)));
  static constexpr bool SMALL = true;
  static __device__ __forceinline__ v4 big(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  // 4 rows x 16 cells (four 4x4 blocks), same B register as the large shape; lane l of the result holds
  // row (l >> 4) of cell (l & 15).  About 1/6 of the issue time of the large shape (mfma_tables.hpp).
  static __device__ __forceinline__ double small(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
};
template <>
struct RT<float> {
  typedef float v4 __attribute__((ext_vector_type(4)));
  static constexpr bool SMALL = false;
  static __device__ __forceinline__ v4 big(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float small(float, float, float c) { return c; }  // never reached (NSM = 0)
};

// Resident blocks per CU every instantiation of a (type, degree) must fit (launch bound) and the persistent grid is
// sized for (mfma_blocks_per_cu): registers and LDS allow exactly two at degrees 3 and 4 (a third block at degree 3 measured
// slower); the low orders are memory-bound and light (12 / 28 KB of tiles) and want more waves in flight.  One number for
// both: a grid sized for more blocks than a fused instantiation can hold runs a second, partly empty round.
template <typename R, int P>
constexpr int mfma_resident_blocks() {
  return sizeof(R) == 4 ? 3 : (P == 1 ? 4 : (P == 2 ? 3 : 2));   // float kernels: at most 168 VGPRs and half the table bytes
}

template <int P, typename R>
struct MG {
  static constexpr int ND = ElemDims<3, P>::ND;
  static constexpr int NF = ElemDims<3, P>::NF;
  static constexpr int KS = (ND + 3) / 4;
  static constexpr int KSF = (NF + 3) / 4;
  static constexpr int MTF = RT<R>::SMALL ? ND / 16 : (ND + 15) / 16;        // 16-row tiles over the nodes
  static constexpr int NSM = RT<R>::SMALL ? (ND % 16 + 3) / 4 : 0;            // 4-row tiles (4x4x4 MFMA) over the remaining rows
  static constexpr int MTT = MTF + NSM;
  static constexpr int MTFA = MTF > 0 ? MTF : 1;     // array extent (degrees 1 and 2 have no full tile)
  static constexpr int NSMA = NSM > 0 ? NSM : 1;
  static constexpr int S4 = (ND + 3) / 4;
  static constexpr int NFRAG_F = MTT * 3 * KS;
  static constexpr int NFRAG_G = 3 * MTT * KS;
  static constexpr int NFRAG_L = 4 * MTT * KSF;
  // factorised G volume (mfma_tables.hpp MfmaFactGeom): rank of the D_r = dim P_{p-1}, row tiles of Q, k-steps over the rank
  static constexpr int RK = (P >= 2) ? ElemDims<3, (P >= 2 ? P - 1 : 1)>::ND : 1;
  static constexpr int QLT = RK / 16;
  static constexpr int QST = (RK % 16 + 3) / 4;
  static constexpr int KR = (RK + 3) / 4;
  static constexpr int NFRAG_Q = (QLT + QST) * KS;
  static constexpr int NFRAG_P = 3 * MTT * KR;
};

// Streaming accesses: old values of the fused combine and all results are touched once per launch; the
// non-temporal hint keeps them from displacing the cell data that the neighbours' lift phases are about to ask
// the L2 for: fabric reads -4 % (F) to -23 % (G<4,0>), step time -2 %.  (The same hint on the trace loads
// themselves changes no traffic and costs time; variants are tools/experiments/kernels_mfma_switches.patch.)
#define LD_STREAM(p) __builtin_nontemporal_load(p)
#define ST_STREAM(p, v) __builtin_nontemporal_store((v), (p))
// A wave raises its issue priority for its MFMA phases (volume, lifts) and drops it for the epilogue,
// so that the other wave of the SIMD cannot hold up matrix instructions with its loads and stores
// (+1 % on the step).
#define SG_PRIO(p) __builtin_amdgcn_s_setprio(p)
// Issue priorities per phase, one hex digit each: 0xVLE = volume, lifts, epilogue; per kernel (F / G, plain / fused).
// The lift phase is the one that waits for memory (neighbour traces) and has little matrix work per load; the volume
// phase has matrix work in abundance.  With the lifts ABOVE the volume a wave in its lifts gets its few matrix
// instructions issued the moment their operands arrive and the SIMD's other wave fills the gaps from its volume
// phase, instead of both waves queueing at equal priority: round 2's 0x330 -> 0x120 is 2-4 % on the step
// (profiles/r03/priority_sweep.txt: plain stages 1.22-1.28 -> 1.17-1.20 ms; the fused G stage does not care; the
// fused F stage, bandwidth-bound, wants the opposite order once the neighbours come from the table: 0x320).
// Round 5: stage UTEMP (F, MODE 2: reads u1 beside its result) sides with the fused F stage, 1.284 -> 1.270 ms; the fused G
// stage (S1 = s0 + g(w)) still does not care (0x320 / 0x210: +2 % on that stage).
constexpr int SG_PRIO_F0 = 0x120, SG_PRIO_F1 = 0x320, SG_PRIO_F2 = 0x320, SG_PRIO_G0 = 0x120, SG_PRIO_G1 = 0x120;
#define SG_PRIO_VOL ((PRIO3 >> 8) & 3)
#define SG_PRIO_LIFT ((PRIO3 >> 4) & 3)
#define SG_PRIO_EPI (PRIO3 & 3)
#define MFMA64(a, b, c) RT<R>::big((a), (b), (c))
#define MFMA4(a, b, c) RT<R>::small((a), (b), (c))

// Diagnostic build only (-DSG_STAMPS, SEIGEN_HIP_STAMPS=1): cycle stamps at the phase boundaries
// of an item, summed per wave in scalars and added to A.dbg at the end.  Never in the shipped build.
#ifdef SG_STAMPS
#define STAMP(t)                                                                   \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#define STAMP_DECL unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, sta = 0, stb = 0, sacc[4] = {0, 0, 0, 0}, ssub[3] = {0, 0, 0}, sitems = 0
#define STAMP_ACC                                                                  \
  do {                                                                             \
    sacc[0] += st1 - st0;                                                          \
    sacc[1] += st2 - st1;                                                          \
    sacc[2] += st3 - st2;                                                          \
    sacc[3] += st4 - st3;                                                          \
    if (sta) { ssub[0] += sta - st2; ssub[1] += stb - sta; ssub[2] += st3 - stb; } \
    sitems += 1;                                                                   \
  } while (0)
#define STAMP_FLUSH                                                                \
  do {                                                                             \
    if (A.dbg && lane == 0) {                                                      \
      for (int z = 0; z < 4; ++z) atomicAdd(&A.dbg[z], sacc[z]);                   \
      atomicAdd(&A.dbg[4], sitems);                                                \
      for (int z = 0; z < 3; ++z) atomicAdd(&A.dbg[5 + z], ssub[z]);               \
    }                                                                              \
  } while (0)
#else
#define STAMP(t) do { } while (0)
#define STAMP_DECL
#define STAMP_ACC do { } while (0)
#define STAMP_FLUSH do { } while (0)
#endif

// Symmetric-stress mode (SYM = 1): every stress field of the run is symmetric (g always produces
// a symmetric tensor, elastic.py:211-219; transfer.cpp checks what the user uploads), so only the six
// lines (i <= j) of each node are read and written; the other three keep their slots but are
// never touched.  Same results, 1/3 less stress traffic.
template <int SYM, typename R>
__device__ __forceinline__ void load_tensor(const R* p, int stride, R (&T)[9]) {
  if (SYM) {
    T[0] = p[0];
    T[1] = p[1 * stride];
    T[2] = p[2 * stride];
    T[4] = p[4 * stride];
    T[5] = p[5 * stride];
    T[8] = p[8 * stride];
    T[3] = T[1];
    T[6] = T[2];
    T[7] = T[5];
  } else {
#pragma unroll
    for (int c = 0; c < 9; ++c) T[c] = p[c * stride];
  }
}
__device__ __forceinline__ constexpr bool upper(int ij) { return (ij / 3) <= (ij % 3); }

// Neighbour tensor at one facet node.  GHOST = 0 (a block without neighbour blocks): out of the
// field, exactly load_tensor.  GHOST = 1: a lane whose neighbour lives in another block reads the
// packed remote trace instead, which holds only the column g_i = T_i,axis of the block side's axis
// (kernels.hip pack_one).  The other columns meet (c n)_j = 0 on such a facet, so ANY finite value
// in their place gives the same flux bit for bit: the loads stay unconditional and only their
// offsets differ per lane - c * 16 into a field, or into the 3-word record such that every needed
// (i, axis) entry (or, in symmetric mode, the pair that stands for it) finds g_i:
//   full tensor: (i, j) -> i;   pairs i <= j: axis 0 -> j, axis 1 -> i + j - 1 (clamped), axis 2 -> i.
// `axis` is uniform over the wave (a property of class and facet).  No branch: a divergent branch
// around the loads costs the F stages a factor two (the hand-pipelined load/MFMA interleaving is lost).
template <int SYM, int GHOST, typename R>
__device__ __forceinline__ void load_trace(const R* p, bool ghost, int axis, R (&T)[9]) {
  if (!GHOST) {
    load_tensor<SYM>(p, 16, T);
    return;
  }
  auto at = [&](int i, int j) {
    int og = i;
    if (SYM) {
      const int mid = (i + j - 1 < 0) ? 0 : (i + j - 1 > 2 ? 2 : i + j - 1);
      og = (axis == 0) ? j : (axis == 1 ? mid : i);
    }
    return p[ghost ? og : (i * 3 + j) * 16];
  };
  if (SYM) {
    T[0] = at(0, 0);
    T[1] = at(0, 1);
    T[2] = at(0, 2);
    T[4] = at(1, 1);
    T[5] = at(1, 2);
    T[8] = at(2, 2);
    T[3] = T[1];
    T[6] = T[2];
    T[7] = T[5];
  } else {
#pragma unroll
    for (int c = 0; c < 9; ++c) T[c] = at(c / 3, c % 3);
  }
}

// wave-uniform test on the bits (scalar compare and branch; there is no scalar f64 compare)
__device__ __forceinline__ bool uniform_nonzero(double c) {
  return (__builtin_bit_cast(unsigned long long, c) << 1) != 0ull;
}
__device__ __forceinline__ bool uniform_nonzero(float c) { return (__builtin_bit_cast(unsigned, c) << 1) != 0u; }

struct LaneGeo {
  long c;      // linear cube index of this lane's cell
  int cc[3];   // cube coordinates
  bool valid;  // a real cube (not layout padding)
  bool active; // valid and inside the launch's region
};

typedef __attribute__((address_space(4))) const MfmaConst cMfmaConst;
typedef __attribute__((address_space(4))) const MfmaClassConst cMfmaClassConst;

// (MD: the scalar-load copy of the mesh constants, MfmaConst, or the MeshDev copy in LDS)
template <typename MD>
__device__ __forceinline__ LaneGeo lane_geo(const MD& md, const StageArgs& A, long g, int w) {
  LaneGeo L;
  L.c = g * 16 + w;
  L.valid = L.c < md.ncube;
  // 32-bit arithmetic: a block has far fewer than 2^31 cubes (288 GB hold about 2^24 of them at
  // degree 1), and the 64-bit divisions this replaces are some hundred instructions each
  const unsigned cl = L.valid ? (unsigned)L.c : 0u;
  const unsigned n0 = (unsigned)md.n[0], n1 = (unsigned)md.n[1];
  const unsigned t = cl / n0, z = t / n1;
  L.cc[0] = (int)(cl - t * n0);
  L.cc[1] = (int)(t - z * n1);
  L.cc[2] = (int)z;
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

// Where this lane finds its neighbour's trace across facet f.
template <typename R>
struct NbrRef {
  const R* p;  // base pointer of the neighbour cell (or ghost slot)
  int cstride;      // stride between (node, comp) entries: 16 in a field, 1 in a packed ghost buffer
  bool ghost;
  bool physical;    // domain boundary: no neighbour (p then points at the own cell)
};

// Whole-block launches test no region boxes: the cube coordinates (two integer divisions per lane) are not needed
__device__ __forceinline__ LaneGeo lane_geo_all(long ncube, long g, int w) {
  LaneGeo L;
  L.c = g * 16 + w;
  L.valid = L.c < ncube;
  L.cc[0] = L.cc[1] = L.cc[2] = 0;
  L.active = L.valid;
  return L;
}

// The four neighbours of this lane's cell from the block's table (StageArgs::nbr_tab): one 16-byte load per lane
typedef int nbr4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ nbr4 load_nbr4(const StageArgs& A, long item, int w) {
  return *reinterpret_cast<const nbr4*>(A.nbr_tab + (item * 16 + w) * 4);
}
// e: table entry of facet f; side: block side the facet crosses (wave-uniform; only used for a remote trace)
template <int ND, int NF, int NC, typename R_>
__device__ __forceinline__ NbrRef<R_> nbr_from_entry(const StageArgs& A, int e, int side, const R_* own_base) {
  NbrRef<R_> R;
  const R_* const fin = reinterpret_cast<const R_*>(A.in);
  R.cstride = 16;
  R.ghost = e < -1;
  R.physical = e == -1;
  const long slot = e < 0 ? 0 : e;
  const R_* pin = fin + ((slot >> 4) * (long)ND) * NC * 16 + (slot & 15);
  R.p = e < 0 ? own_base : pin;
  if (__any(R.ghost)) {        // blocks with neighbour blocks only, and there only the shell items
    if (R.ghost) {
      R.p = reinterpret_cast<const R_*>(A.ghost[side]) + (long)(-2 - e) * NF * 3;
      R.cstride = 1;
    }
  }
  return R;
}

// XCD-aware work split: blocks with equal blockIdx % 8 share an XCD (and its L2), so each
// label gets one contiguous range of items; a different placement only changes speed.
struct ItemRange {
  long lo, hi, step;
  long chunk, xcd, nitems;  // chunk > 0: `it` counts the XCD's own items; item_of() maps to the global item
};
__device__ __forceinline__ ItemRange item_range(long nitems, int wave, int spread, long chunk) {
  ItemRange r;
  const long wpb = blockDim.x >> 6;   // waves per block (4)
  r.chunk = 0;
  r.xcd = 0;
  r.nitems = nitems;
  if (spread) {  // boundary shells: round-robin over every wave of the grid (over the item list if given)
    r.lo = (long)blockIdx.x * wpb + wave;
    r.hi = nitems;
    r.step = (long)gridDim.x * wpb;
    return r;
  }
  const long nblk = gridDim.x;
  const long xcd = blockIdx.x % 8;
  const long slot = blockIdx.x / 8;
  const long blocks_here = (nblk - xcd + 7) / 8;
  r.step = blocks_here * wpb;
  if (chunk > 0) {
    // chunks xcd, xcd + 8, ... of `chunk` consecutive items each
    const long nchunks = (nitems + chunk - 1) / chunk;
    const long mine = (nchunks - xcd + 7) / 8;
    r.chunk = chunk;
    r.xcd = xcd;
    r.lo = slot * wpb + wave;
    r.hi = mine > 0 ? mine * chunk : 0;
    return r;
  }
  const long ipx = (nitems + 7) / 8;
  r.lo = xcd * ipx + slot * wpb + wave;
  r.hi = (xcd + 1) * ipx < nitems ? (xcd + 1) * ipx : nitems;
  return r;
}
// global item of loop index `it`, or -1 (past the end of the last chunk)
__device__ __forceinline__ long item_of(const ItemRange& r, long it) {
  if (r.chunk == 0) return it;
  const long c = it / r.chunk;
  const long item = (c * 8 + r.xcd) * r.chunk + (it - c * r.chunk);
  return item < r.nitems ? item : -1;
}

// Operator tiles and mesh tables into LDS, once per block.  All of a thread's loads are issued
// before the first one is consumed (a plain copy loop compiles to load / wait / write per element:
// 33 dependent L2 round trips, about 20 us of every launch).
template <int N, typename R, int NT = 256>
__device__ __forceinline__ void copy_to_lds(R* dst, const R* __restrict__ src) {
  constexpr int PER = (N + NT - 1) / NT;
  R v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = threadIdx.x + j * NT;
    v[j] = (i < N) ? src[i] : R(0);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = threadIdx.x + j * NT;
    if (i < N) dst[i] = v[j];
  }
}

template <int NV, int NL, typename R, int NT = 256>
__device__ __forceinline__ void load_tables(R* sAV, R* sAL, const StageArgs& A) {
  copy_to_lds<NV * 64, R, NT>(sAV, reinterpret_cast<const R*>(A.fragV));
  copy_to_lds<NL * 64, R, NT>(sAL, reinterpret_cast<const R*>(A.fragL));
  __syncthreads();
}

template <int NV, int NL, typename R>
__device__ __forceinline__ void load_tables(R* sAV, R* sAL, MeshDev* sMd, const StageArgs& A) {
  const int* src = reinterpret_cast<const int*>(A.md);
  int* dst = reinterpret_cast<int*>(sMd);
  constexpr int NI = (int)(sizeof(MeshDev) / sizeof(int)), PERI = (NI + 255) / 256;
  int w[PERI];
#pragma unroll
  for (int j = 0; j < PERI; ++j) {
    const int i = threadIdx.x + j * 256;
    w[j] = (i < NI) ? src[i] : 0;
  }
  copy_to_lds<NV * 64>(sAV, reinterpret_cast<const R*>(A.fragV));
  copy_to_lds<NL * 64>(sAL, reinterpret_cast<const R*>(A.fragL));
#pragma unroll
  for (int j = 0; j < PERI; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < NI) dst[i] = w[j];
  }
  __syncthreads();
}


// (c n)_j T_ij for one row i of a tensor, in ONE fixed order of operations (not left to the contraction heuristics: a
// facet's flux must not depend on which instantiation - whole block, shell, ghost-reading - a launch happens to use).
template <typename R>
__device__ __forceinline__ R ndot(R c0, R c1, R c2, R t0, R t1, R t2) {
  R r = c0 * t0;
  r = __builtin_fma(c1, t1, r);
  r = __builtin_fma(c2, t2, r);
  return r;
}
template <>
__device__ __forceinline__ float ndot<float>(float c0, float c1, float c2, float t0, float t1, float t2) {
  float r = c0 * t0;
  r = __builtin_fmaf(c1, t1, r);
  r = __builtin_fmaf(c2, t2, r);
  return r;
}


// --------------------------------------------------------------------------------------------
//  G: sh_ij = lam d_ij W_kk + mu (W_ij + W_ji),  W_ik = -Jinv_rk (D_r u_i) + sum_f (c n)_k L_f u^_i
//  Row tile t of every D_r leaves node a = 4 m + q in lane-group q (m = 4 t + reg for a large
//  tile, m = 4*MTF + s for small tile s), so the three D_r u_i of one node meet in the same lane;
//  only W_ii and W_ij + W_ji are accumulated (6 * S4 values).
// --------------------------------------------------------------------------------------------
template <typename R, int P, int MODE, int SYM, int FACT>
__global__ __launch_bounds__(256, (mfma_resident_blocks<R, P>())) void mfma_stage_G(StageArgs A) {
  using M = MG<P, R>;
  typedef typename RT<R>::v4 d4;
  constexpr int PRIO3 = MODE ? SG_PRIO_G1 : SG_PRIO_G0;
  constexpr int ND = M::ND, NF = M::NF, KS = M::KS, KSF = M::KSF, MTF = M::MTF, NSM = M::NSM, MTT = M::MTT, S4 = M::S4;
  // B-operand prefetch distance, in k-steps.  A float k-step is 96 matrix cycles per component against 144 in
  // double, so the same latency needs more k-steps in flight (measured: G<float,4,0> 1.10 -> 0.97 ms with 6, 0.94 with 8,
  // which costs the fused kernel more than it gains)
  constexpr int PF = sizeof(R) == 4 ? 6 : 4;
  constexpr int QLT = M::QLT, QST = M::QST, KR = M::KR;
  // operator tiles in LDS: the row tiles of the three D_r (E_r: with the own-trace half of the flux folded in), or
  // (FACT) those of the P_r in their place (A.fragV) and the Q tiles beside them
  __shared__ R sAV[(FACT ? M::NFRAG_P : M::NFRAG_G) * 64];
  __shared__ R sQ[FACT ? M::NFRAG_Q * 64 : 1];
  __shared__ R sAL[M::NFRAG_L * 64];
  __shared__ MeshDev sMd;
  if constexpr (FACT) copy_to_lds<M::NFRAG_Q * 64>(sQ, reinterpret_cast<const R*>(A.fragQ));
  load_tables<(FACT ? M::NFRAG_P : M::NFRAG_G), M::NFRAG_L>(sAV, sAL, &sMd, A);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, w = lane & 15;
  // uniform reads of class constants: constant address space => s_load (scalar cache)
  typedef __attribute__((address_space(4))) const MeshDev cMeshDev;
  const cMeshDev* md = (const cMeshDev*)(unsigned long long)A.md;
  const R* __restrict__ in = reinterpret_cast<const R*>(A.in);
  R* __restrict__ out = reinterpret_cast<R*>(A.out);
  const R c_self = (R)A.c_self, c_new = (R)A.c_new;
  const long ngroups = sMd.ncube_pad >> 4;
  const ItemRange ir = item_range(A.item_list ? (long)A.nlist : ngroups * 6, wave, A.spread, A.order_chunk);

  STAMP_DECL;
  for (long it = ir.lo; it < ir.hi; it += ir.step) {
    STAMP(st0);
    const long iti = item_of(ir, it);
    if (iti < 0) continue;
    const long item = A.item_list ? (long)A.item_list[iti] : iti;
    const long g = item / 6;
    const int k = (int)(item - g * 6);
    const LaneGeo L = A.all_active ? lane_geo_all(sMd.ncube, g, w) : lane_geo(sMd, A, g, w);
    const nbr4 nbe = load_nbr4(A, item, w);
    if (!__any(L.active)) continue;
    const R* own = in + ((g * 6 + k) * (long)ND) * 3 * 16 + w;
    // B row of this lane at k-step ks = node 4 ks + q: one pointer per item plus compile-time
    // offsets (a table of per-k-step offsets is item-invariant, gets hoisted out of the item loop
    // as 64-bit values and spilled; every reload then waits for ALL loads in flight).  Only the last
    // k-step can run past ND; those rows meet all-zero operator columns, so any finite value will
    // do there: clamp instead of branching.
    int qo = q * 3 * 16;
    asm volatile("" : "+v"(qo));
    const R* ownq = own + qo;
    const R* ownl = (4 * (KS - 1) + q < ND) ? ownq + (KS - 1) * 4 * 3 * 16 : own;
    auto brow = [&](int ks) { return (ks == KS - 1) ? ownl : ownq + ks * 4 * 3 * 16; };
    // operator tiles are item-invariant: an opaque lane offset keeps the compiler from hoisting
    // all of them into registers (and, unlike a laundered pointer, keeps the reads ds_read_b64)
    int lo = lane;
    asm volatile("" : "+v"(lo));

    R Jm[3][3], cnf[4][3];  // class constants (wave-uniform)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) Jm[r][j] = (R)(-md->Jinv[k][r][j]);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 3; ++j) cnf[f][j] = (R)md->cn[k][f][j];

    R Sd[3][S4], So[3][S4];  // W_ii and W_ij + W_ji for (0,1), (0,2), (1,2)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int m = 0; m < S4; ++m) Sd[i][m] = So[i][m] = R(0);

    constexpr int NB = 2;   // trace buffers: facet f in nx[f % NB], the next NB - 1 facets on their way
    constexpr int NBO = FACT ? NB : 1;
    R nx[NB][KSF][3], nxo[NBO][KSF][3];   // nxo (FACT): the own traces of the same facet nodes
    auto request = [&](int f, R (&dst)[KSF][3], R (&dso)[KSF][3]) {
      const NbrRef<R> NR = nbr_from_entry<ND, NF, 3>(A, nbe[f], 2 * sMd.nb_axis[k][f] + (sMd.nb_dir[k][f] > 0 ? 1 : 0), own);
#pragma unroll
      for (int ks = 0; ks < KSF; ++ks) {
        const int bb = (4 * ks + q < NF) ? 4 * ks + q : 0;  // padded rows meet zero lift columns
        const int on = sMd.fnode[f][bb];
        const int nn = NR.ghost ? sMd.nb_fnode[k][f][bb] : (NR.physical ? on : sMd.nb_node[k][f][bb]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          dst[ks][i] = NR.p[(nn * 3 + i) * NR.cstride];
          // FACT: u^ = 1/2 (own + neighbour) (own + own on a boundary facet), the 1/2 being part of the lift tiles.
          // Two loads per value: the own rows were read a phase ago and come out of the L2.
          if constexpr (FACT) dso[ks][i] = own[(on * 3 + i) * 16];
        }
      }
    };
    SG_PRIO(SG_PRIO_VOL);
    STAMP(st1);
    // W_ik += Jm[r][k] (D_r u_i) for the row tiles of one reference direction.  On the Kuhn classes J^-1 has five
    // non-zeros out of nine: the zero columns are skipped with a scalar branch (the class is uniform over the wave)
    auto fold_dir = [&](int r, const d4 (&acc)[M::MTFA][3], const R (&accs)[M::NSMA][3]) {
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const R c = Jm[r][kk];
        if (uniform_nonzero(c)) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int t = 0; t < MTF; ++t)
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                if (4 * t + reg >= S4) continue;  // padded rows of the last float tile
                if (i == kk)
                  Sd[i][4 * t + reg] += c * acc[t][i][reg];
                else
                  So[i + kk - 1][4 * t + reg] += c * acc[t][i][reg];
              }
#pragma unroll
            for (int t = 0; t < NSM; ++t) {
              if (i == kk)
                Sd[i][4 * MTF + t] += c * accs[t][i];
              else
                So[i + kk - 1][4 * MTF + t] += c * accs[t][i];
            }
          }
        }
      }
      // pin the folds here: LLVM otherwise sinks these FMA chains down to the epilogue (their
      // only use), which keeps every accumulator tile live and spills
#pragma unroll
      for (int m = 0; m < S4; ++m)
#pragma unroll
        for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(Sd[i][m]), "+v"(So[i][m]));
    };
    // one k-step of a direction's row tiles: acc += tile(r, t, ks) x b
    auto tiles_step = [&](int nks, int r, int ks, const R (&b)[3], d4 (&acc)[M::MTFA][3], R (&accs)[M::NSMA][3]) {
#pragma unroll
      for (int t = 0; t < MTT; ++t) {
        const R a = sAV[((r * MTT + t) * nks + ks) * 64 + lo];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (t < MTF)
            acc[t < MTF ? t : 0][i] = MFMA64(a, b[i], acc[t < MTF ? t : 0][i]);
          else
            accs[t < MTF ? 0 : t - MTF][i] = MFMA4(a, b[i], accs[t < MTF ? 0 : t - MTF][i]);
        }
      }
    };
    auto clear_acc = [&](d4 (&acc)[M::MTFA][3], R (&accs)[M::NSMA][3]) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int t = 0; t < MTF; ++t) acc[t][i] = d4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NSM; ++t) accs[t][i] = R(0);
      }
    };
    if constexpr (!FACT) {
      // ---- volume: B = the cells' own nodal values, straight from memory, PF k-steps ahead.
      //      One pass over the k-steps per reference direction r with all row tiles of D_r live:
      //      one load of a B row feeds 3*MTT MFMAs, and the cell data are read three times (twice
      //      through L1/L2); 3*MTF accumulator tiles + 3*NSM values are live next to Sd/So.
      constexpr int NS = 3 * KS;
      R bq[PF][3];
#pragma unroll
      for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int i = 0; i < 3; ++i) bq[s][i] = brow(s % KS)[i * 16];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        d4 acc[M::MTFA][3];
        R accs[M::NSMA][3];
        clear_acc(acc, accs);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int s = r * KS + ks;
          R b[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) b[i] = bq[s % PF][i];
          if (s + PF < NS) {
#pragma unroll
            for (int i = 0; i < 3; ++i) bq[s % PF][i] = brow((s + PF) % KS)[i * 16];
          }
          tiles_step(KS, r, ks, b, acc, accs);
        }
        fold_dir(r, acc, accs);
      }
    } else {
      // ---- volume, FACTORISED (double, degrees 3 and 4; mfma_tables.hpp): the three D_r of a degree-p element
      //      differentiate, so they have rank dim P_{p-1} and share their row space: D_r = P_r Q.  y_i = Q u_i once
      //      (one pass over the own rows, PF k-steps ahead), then per reference direction z = P_r y_i with the results
      //      of the first product as B operands of the second as they are (accumulator register `reg` of a large tile
      //      = k-step `reg`; a small tile's value = its k-step): 8640 matrix cycles per 16 cells at degree 4 instead
      //      of 11664.  The price: the own-trace half of the central flux cannot stay folded into rank-deficient
      //      tiles, so the lifts take 1/2 (own + neighbour) and load the own traces as well (request).
      d4 yl[QLT > 0 ? QLT : 1][3];
      R ys[QST > 0 ? QST : 1][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int t = 0; t < QLT; ++t) yl[t][i] = d4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < QST; ++t) ys[t][i] = R(0);
      }
      R bq[PF][3];
#pragma unroll
      for (int s = 0; s < PF && s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 3; ++i) bq[s][i] = brow(s)[i * 16];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        R b[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = bq[ks % PF][i];
        if (ks + PF < KS) {
#pragma unroll
          for (int i = 0; i < 3; ++i) bq[ks % PF][i] = brow(ks + PF)[i * 16];
        }
#pragma unroll
        for (int t = 0; t < QLT + QST; ++t) {
          const R a = sQ[(t * KS + ks) * 64 + lo];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            if (t < QLT)
              yl[t < QLT ? t : 0][i] = MFMA64(a, b[i], yl[t < QLT ? t : 0][i]);
            else
              ys[t < QLT ? 0 : t - QLT][i] = MFMA4(a, b[i], ys[t < QLT ? 0 : t - QLT][i]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        d4 acc[M::MTFA][3];
        R accs[M::NSMA][3];
        clear_acc(acc, accs);
#pragma unroll
        for (int ks = 0; ks < KR; ++ks) {
          R b[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) b[i] = (ks < 4 * QLT) ? yl[(ks / 4) < QLT ? ks / 4 : 0][i][ks % 4] : ys[(ks - 4 * QLT) >= 0 && (ks - 4 * QLT) < QST ? ks - 4 * QLT : 0][i];
          tiles_step(KR, r, ks, b, acc, accs);
        }
        fold_dir(r, acc, accs);
      }
    }

    STAMP(st2);
    // ---- facet lifts.  u^ = avg(u) on interior facets, own trace on the boundary
    //      (elastic.py:213-216).  Plain tiles: the own half of avg(u) is part of the volume tiles (E_r,
    //      mfma_tables.cpp), so the lift carries 1/2 u- on interior facets and the missing
    //      1/2 u+ on boundary facets: in both cases half of whatever np[f] points at
    //      (a boundary lane's neighbour pointer is its own cell); the 1/2 is in the lift tiles.
    //      FACT: 1/2 (own + that).
    //      Register budget: the facet's KSF*3 B values stay resident while the row tiles are
    //      accumulated one at a time (3 accumulators instead of 3*MTT); the next facet's values are
    //      requested before the last tile pass.  Row tile t covers the row-quads m = 4t .. 4t+3
    //      (large) or the single row-quad m = 4*MTF + (t - MTF) (small).
    {
#pragma unroll
      for (int f0 = 0; f0 < NB - 1; ++f0) request(f0, nx[f0], nxo[f0 % NBO]);
      SG_PRIO(SG_PRIO_LIFT);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        R(&flf)[KSF][3] = nx[f % NB];
        if constexpr (FACT) {
#pragma unroll
          for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
            for (int i = 0; i < 3; ++i) flf[ks][i] += nxo[f % NBO][ks][i];
        }
#pragma unroll
        for (int t = 0; t < MTT; ++t) {
          // next facet's traces: asked for a whole facet ahead
          if (t == 0 && f + NB - 1 < 4) request(f + NB - 1, nx[(f + NB - 1) % NB], nxo[(f + NB - 1) % NBO]);
          // W_ik += (c n)_f,k (L_f u^_i): half of the normal components of a Kuhn class are zero
          auto fold = [&](int m0, int nm, const R (&v)[3][4]) {
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
              const R c = cnf[f][kk];
              if (uniform_nonzero(c)) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    if (j < nm && m0 + j < S4) {
                      if (i == kk)
                        Sd[i][m0 + j] += c * v[i][j];
                      else
                        So[i + kk - 1][m0 + j] += c * v[i][j];
                    }
              }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nm && m0 + j < S4) {
#pragma unroll
                for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(Sd[i][m0 + j]), "+v"(So[i][m0 + j]));
              }
          };
          if (t < MTF) {
            d4 tmp[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) tmp[i] = d4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks) {
              const R a = sAL[((f * MTT + t) * KSF + ks) * 64 + lo];
#pragma unroll
              for (int i = 0; i < 3; ++i) tmp[i] = MFMA64(a, flf[ks][i], tmp[i]);
            }
            {
              R v[3][4];
#pragma unroll
              for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) v[i][reg] = tmp[i][reg];
              fold(4 * t, 4, v);
            }
          } else {
            R tmp[3] = {R(0), R(0), R(0)};
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks) {
              const R a = sAL[((f * MTT + t) * KSF + ks) * 64 + lo];
#pragma unroll
              for (int i = 0; i < 3; ++i) tmp[i] = MFMA4(a, flf[ks][i], tmp[i]);
            }
            {
              const R v[3][4] = {{tmp[0], 0, 0, 0}, {tmp[1], 0, 0, 0}, {tmp[2], 0, 0, 0}};
              fold(4 * MTF + (t - MTF), 1, v);
            }
          }
        }
      }
    }

    SG_PRIO(SG_PRIO_EPI);
    STAMP(st3);
    // ---- stress and epilogue (MODE 1: s = c_self*s + c_new*rhs in place - the rhs being G(dt u1 + dt^3/24 utemp),
    //      which is dt sh1 + dt^3/24 sh2 of elastic.py:348-352 in one application of the linear g, stages.cpp)
    //      vmcnt counts loads and stores together and the two kinds complete out of order with
    //      each other, so any wait for a load (or a scratch reload) with stores in flight becomes a
    //      wait for every store's acknowledgement.  Hence: finish ALL loads first, building the
    //      results in place in Sd/So, and issue the item's stores back to back at the very end.
    {
      const long e = (L.valid ? L.c : 0) * 6 + k;
      const R lam = (R)(A.per_cell ? A.lam[e] : A.lam0);
      const R mu = (R)(A.per_cell ? A.mu[e] : A.mu0);
      const long obase = ((g * 6 + k) * (long)ND) * 9 * 16 + w;
      // row-quad m of this lane = node 4 m + q: per-item base + compile-time offsets (see brow)
      int qo9 = q * 9 * 16;
      asm volatile("" : "+v"(qo9));
      const long ob_q = obase + qo9;
      const long ob_l = (4 * (S4 - 1) + q < ND) ? ob_q + (long)(S4 - 1) * 4 * 9 * 16 : obase;
      auto orow = [&](int m) { return (m == S4 - 1) ? ob_l : ob_q + (long)m * 4 * 9 * 16; };
      if constexpr (SYM || MODE == 0) {
        constexpr int NL = SYM ? 6 : 9;  // SYM: the lines (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
        constexpr int PDE = 4;           // row-quads of old values in flight (MODE 1)
        R po[PDE][6];
        auto line = [](int c) { return SYM ? (c < 3 ? c : (c < 5 ? c + 1 : 8)) : c; };
        auto fetch_old = [&](int m) {
          const long o1 = orow(m);
#pragma unroll
          for (int c = 0; c < 6; ++c) po[m % PDE][c] = LD_STREAM(&out[o1 + line(c) * 16]);
        };
        if (MODE == 1) {
#pragma unroll
          for (int m = 0; m < PDE && m < S4; ++m) fetch_old(m);
        }
#pragma unroll
        for (int m = 0; m < S4; ++m) {
          const R tr = lam * (Sd[0][m] + Sd[1][m] + Sd[2][m]);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            Sd[i][m] = R(2) * mu * Sd[i][m] + tr;
            So[i][m] = mu * So[i][m];
          }
          if (MODE == 1) {
            // line order of the six slots: Sd0 So0 So1 Sd1 So2 Sd2
            Sd[0][m] = c_self * po[m % PDE][0] + c_new * Sd[0][m];
            So[0][m] = c_self * po[m % PDE][1] + c_new * So[0][m];
            So[1][m] = c_self * po[m % PDE][2] + c_new * So[1][m];
            Sd[1][m] = c_self * po[m % PDE][3] + c_new * Sd[1][m];
            So[2][m] = c_self * po[m % PDE][4] + c_new * So[2][m];
            Sd[2][m] = c_self * po[m % PDE][5] + c_new * Sd[2][m];
            if (m + PDE < S4) fetch_old(m + PDE);
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(Sd[i][m]), "+v"(So[i][m]));
        }
#pragma unroll
        for (int m = 0; m < S4; ++m) {
          const int a = 4 * m + q;
          const long o = orow(m);
          if (L.active && a < ND) {
            ST_STREAM(&out[o + 0 * 16], Sd[0][m]);
            ST_STREAM(&out[o + 1 * 16], So[0][m]);
            ST_STREAM(&out[o + 2 * 16], So[1][m]);
            ST_STREAM(&out[o + 4 * 16], Sd[1][m]);
            ST_STREAM(&out[o + 5 * 16], So[2][m]);
            ST_STREAM(&out[o + 8 * 16], Sd[2][m]);
            if (!SYM) {
              ST_STREAM(&out[o + 3 * 16], So[0][m]);
              ST_STREAM(&out[o + 6 * 16], So[1][m]);
              ST_STREAM(&out[o + 7 * 16], So[2][m]);
            }
          }
        }
        (void)NL;
      } else {
        // full-tensor in-place combine (asymmetric user data, rare): nine results per node
#pragma unroll
        for (int m = 0; m < S4; ++m) {
          const int a = 4 * m + q;
          const long o = orow(m);
          const R tr = lam * (Sd[0][m] + Sd[1][m] + Sd[2][m]);
          R s[9];
          s[0] = R(2) * mu * Sd[0][m] + tr;
          s[4] = R(2) * mu * Sd[1][m] + tr;
          s[8] = R(2) * mu * Sd[2][m] + tr;
          s[1] = s[3] = mu * So[0][m];
          s[2] = s[6] = mu * So[1][m];
          s[5] = s[7] = mu * So[2][m];
#pragma unroll
          for (int ij = 0; ij < 9; ++ij) s[ij] = c_self * out[o + ij * 16] + c_new * s[ij];
          if (L.active && a < ND) {
#pragma unroll
            for (int ij = 0; ij < 9; ++ij) out[o + ij * 16] = s[ij];
          }
        }
      }
    }
    STAMP(st4);
    STAMP_ACC;
  }
  STAMP_FLUSH;
}

// --------------------------------------------------------------------------------------------
//  F: uh_i = -sum_r D_r (Jinv_rj T_ij) + sum_f L_f [ (c n)_j {T_ij} ] - sponge
// --------------------------------------------------------------------------------------------
template <typename R, int P, int MODE, int SYM, int GHOST>
__global__ __launch_bounds__(256, (mfma_resident_blocks<R, P>())) void mfma_stage_F(StageArgs A) {
  using M = MG<P, R>;
  typedef typename RT<R>::v4 d4;
  constexpr int PRIO3 = MODE == 1 ? SG_PRIO_F1 : (MODE == 2 ? SG_PRIO_F2 : SG_PRIO_F0);
  constexpr int ND = M::ND, NF = M::NF, KS = M::KS, KSF = M::KSF, MTF = M::MTF, NSM = M::NSM, MTT = M::MTT;
  __shared__ R sAV[M::NFRAG_F * 64];
  __shared__ R sAL[M::NFRAG_L * 64];
  // where a lane finds the nodes of a neighbour's trace: offsets of the four facet k-steps per (variant: interior /
  // remote record / domain boundary, class, facet, lane group), tabulated on the host (mfma_tables.cpp
  // mfma_trace_offsets) - one ds_read_b128 per facet instead of three batches of scalar loads, a byte extraction and a
  // multiplication per k-step, and a dozen SGPRs fewer (the neighbour set-up took 4.6 k of an item's 53 k cycles)
  __shared__ nbr4 sFt[3 * 6 * 4 * 4];
  {
    const nbr4* src = reinterpret_cast<const nbr4*>(A.ftab);
    for (int i = threadIdx.x; i < 3 * 6 * 4 * 4; i += 256) sFt[i] = src[i];
  }
  load_tables<M::NFRAG_F, M::NFRAG_L>(sAV, sAL, A);
  const cMfmaConst& mk = *(const cMfmaConst*)(unsigned long long)A.mk;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, w = lane & 15;
  typedef __attribute__((address_space(4))) const MeshDev cMeshDev;
  const cMeshDev* md = (const cMeshDev*)(unsigned long long)A.md;
  const R* __restrict__ in = reinterpret_cast<const R*>(A.in);
  const R* __restrict__ aux = reinterpret_cast<const R*>(A.aux);
  R* __restrict__ out = reinterpret_cast<R*>(A.out);
  const R c_self = (R)A.c_self, c_aux = (R)A.c_aux, c_new = (R)A.c_new;
  const long ngroups = mk.ncube_pad >> 4;
  const ItemRange ir = item_range(A.item_list ? (long)A.nlist : ngroups * 6, wave, A.spread, A.order_chunk);

  STAMP_DECL;
  for (long it = ir.lo; it < ir.hi; it += ir.step) {
    STAMP(st0);
    const long iti = item_of(ir, it);
    if (iti < 0) continue;
    const long item = A.item_list ? (long)A.item_list[iti] : iti;
    const long g = item / 6;
    const int k = (int)(item - g * 6);
    const LaneGeo L = A.all_active ? lane_geo_all(mk.ncube, g, w) : lane_geo(mk, A, g, w);
    const cMfmaClassConst& kc = mk.cls[k];
    const nbr4 nbe = load_nbr4(A, item, w);
    if (!__any(L.active)) continue;
    const R* own = in + ((g * 6 + k) * (long)ND) * 9 * 16 + w;
    int qo = q * 9 * 16;  // B rows: see mfma_stage_G
    asm volatile("" : "+v"(qo));
    const R* ownq = own + qo;
    const R* ownl = (4 * (KS - 1) + q < ND) ? ownq + (KS - 1) * 4 * 9 * 16 : own;
    auto brow = [&](int ks) { return (ks == KS - 1) ? ownl : ownq + ks * 4 * 9 * 16; };
    int lo = lane;
    asm volatile("" : "+v"(lo));

    R Jm[3][3], cnf[4][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) Jm[r][j] = (R)md->Jinv[k][r][j];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 3; ++j) cnf[f][j] = (R)md->cn[k][f][j];

    // rows 16t + 4reg + q in acc[i][t][reg] (large tiles), row 16*MTF + 4s + q in accs[i][s] (small)
    d4 acc[3][M::MTFA];
    R accs[3][M::NSMA];
    const long e = (L.valid ? L.c : 0) * 6 + k;
    const long ubase = ((g * 6 + k) * (long)ND) * 3 * 16 + w;
    int qo3 = q * 3 * 16;  // output rows: per-item base + compile-time offsets (see brow)
    asm volatile("" : "+v"(qo3));
    const long ub_q = ubase + qo3;
    // Fused combine u = c_self u + c_aux uh1 + c_new rhs (MODE 1), EARLY form: the old values are requested at the
    // START of the item, together with the first own rows, and enter the accumulators as (c_self u + c_aux uh1) / c_new;
    // the matrix products add rhs on top and the epilogue only scales by c_new and stores - no loads, no second
    // memory latency at the end of the item (the epilogue of F<4,1> took 10 k of an item's 67 k cycles against 2.8 k in
    // F<4,0>).  Rounding: the sum is formed at the scale of u / c_new, i.e. with the absolute error the final u carries
    // anyway.  c_new = 0 (dt = 0) keeps the late form.
    // (double only: in float u / c_new could leave the exponent range for small dt)
    // MODE 2: the same without the self term, out = c_aux aux + c_new rhs (stage UTEMP leaves w = dt u1 + dt^3/24 utemp,
    // stages.cpp): no old value of `out` is read.
    constexpr bool FUSED = MODE >= 1, SELF = MODE == 1;
    const bool early = FUSED && sizeof(R) == 8 && uniform_nonzero(c_new);
    R cs = c_self, ca = c_aux, cn = c_new;
    if (MODE == 1 && A.rho2 != nullptr) {  // per-cell density (kernels.hpp)
      cs = (R)A.rho2[2 * e];
      ca *= (R)A.rho2[2 * e + 1];
      cn *= (R)A.rho2[2 * e + 1];
    }
    // Sponge of this lane's cell (the block further down): a sigma that is constant over the cell enters as -sigma u_abs at
    // the node.  In the fused stages u_abs IS one of the combine's operands - `out` in stage U1, `aux` in stage UTEMP - so
    // the term is a change of that operand's coefficient, no load and no arithmetic of its own.
    // (the plain stage asks for its sigma where it uses it, further down: asked here it changes the register allocation of
    // the whole item and costs the stage 3 % - without any sponge set)
    R sig = (R)0;
    bool sig_folded = false;
    if (FUSED && A.sponge_sigma != nullptr && L.active) {
      sig = (R)A.sponge_sigma[e];
      if (sig == sig && sig != (R)0) {
        if (SELF && A.uabs == A.out) {
          cs -= cn * sig;
          sig_folded = true;
        } else if (A.uabs == A.aux) {
          ca -= cn * sig;
          sig_folded = true;
        }
      }
    }
    // old values of the in-place combine: every one of the item requested before the first is used
    R po[M::MTFA][4][3], pa[M::MTFA][4][3], pos[M::NSMA][3], pas[M::NSMA][3];
    auto fetch_old = [&]() {
#pragma unroll
      for (int t = 0; t < MTF; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          // the last float tile is zero-padded: its rows beyond ND read a valid address and are not stored
          const bool row_ok = RT<R>::SMALL || (16 * t + 4 * reg + q < ND);
          const long o = row_ok ? ub_q + (long)(16 * t + 4 * reg) * 3 * 16 : ubase;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            po[t][reg][i] = SELF ? LD_STREAM(&out[o + i * 16]) : R(0);
            pa[t][reg][i] = LD_STREAM(&aux[o + i * 16]);
          }
        }
#pragma unroll
      for (int t = 0; t < NSM; ++t) {
        const int a = 16 * MTF + 4 * t + q;
        const long o = (a < ND) ? ub_q + (long)(16 * MTF + 4 * t) * 3 * 16 : ubase;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          pos[t][i] = SELF ? LD_STREAM(&out[o + i * 16]) : R(0);
          pas[t][i] = LD_STREAM(&aux[o + i * 16]);
        }
      }
    };
    if (early) {
      fetch_old();
      const R icn = R(1) / cn;
#pragma unroll
      for (int t = 0; t < MTF; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i][t][reg] = SELF ? (cs * po[t][reg][i] + ca * pa[t][reg][i]) * icn : (ca * pa[t][reg][i]) * icn;
#pragma unroll
      for (int t = 0; t < NSM; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i) accs[i][t] = SELF ? (cs * pos[t][i] + ca * pas[t][i]) * icn : (ca * pas[t][i]) * icn;
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int t = 0; t < MTF; ++t) acc[i][t] = d4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NSM; ++t) accs[i][t] = R(0);
      }
    }

    SG_PRIO(SG_PRIO_VOL);
    STAMP(st1);
    auto do_volume = [&]() {
      // ---- volume: K runs over (r, node); B = T~_ir = Jinv_rj T_ij formed in registers
      {
        constexpr int PFV = sizeof(R) == 4 ? 2 : 1;  // k-steps of own tensors in flight ahead of the MFMAs
        R Tq[PFV][9];
#pragma unroll
        for (int s0 = 0; s0 < PFV && s0 < KS; ++s0) load_tensor<SYM>(brow(s0), 16, Tq[s0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          R T[9];
#pragma unroll
          for (int c = 0; c < 9; ++c) T[c] = Tq[ks % PFV][c];
          if (ks + PFV < KS) {
            load_tensor<SYM>(brow(ks + PFV), 16, Tq[ks % PFV]);
          }
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            R Tt[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) Tt[i] = Jm[r][0] * T[i * 3 + 0] + Jm[r][1] * T[i * 3 + 1] + Jm[r][2] * T[i * 3 + 2];
#pragma unroll
            for (int t = 0; t < MTT; ++t) {
              const R a = sAV[(t * 3 * KS + KS * r + ks) * 64 + lo];
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                if (t < MTF)
                  acc[i][t < MTF ? t : 0] = MFMA64(a, Tt[i], acc[i][t < MTF ? t : 0]);
                else
                  accs[i][t < MTF ? 0 : t - MTF] = MFMA4(a, Tt[i], accs[i][t < MTF ? 0 : t - MTF]);
              }
            }
          }
        }
      }

    };
    auto do_lifts = [&]() {
      // ---- facet lifts of (c n)_j {T_ij}; no ds term in f => zero flux on the boundary
      //      (elastic.py:206).  The own half of {T} is part of the volume tiles (E_r), so the lift
      //      carries +1/2 (c n).T- on interior facets and -1/2 (c n).T+ on boundary facets (which
      //      cancels the folded half): wf * (c n).T of whatever np[f] points at, the 1/2 being part
      //      of the lift tiles.
      constexpr int PFL = sizeof(R) == 4 ? 3 : 2;  // facet k-steps of neighbour traces in flight
      const R* np[4];
      R wf[4];
      int noff[4][KSF];
      bool gh[4];
      int fax[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const NbrRef<R> NR = nbr_from_entry<ND, NF, 9>(A, nbe[f], 2 * kc.nb_axis[f] + (kc.nb_dir[f] > 0 ? 1 : 0), own);
        np[f] = NR.p;
        gh[f] = GHOST && NR.ghost;
        fax[f] = GHOST ? kc.nb_axis[f] : 0;
        wf[f] = NR.physical ? R(-1) : R(1);
        const int var = NR.ghost ? 1 : (NR.physical ? 2 : 0);
        const nbr4 o4 = sFt[((var * 6 + k) * 4 + f) * 4 + q];
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks) noff[f][ks] = o4[ks];
      }
      R nq[PFL][9];
      STAMP(sta);   // diagnostic builds: end of the neighbour set-up
      {
        constexpr int NS = 4 * KSF;
#pragma unroll
        for (int s = 0; s < PFL; ++s)
          load_trace<SYM, GHOST>(np[s / KSF] + noff[s / KSF][s % KSF], gh[s / KSF], fax[s / KSF], nq[s]);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          if (f == 1) STAMP(stb);   // diagnostic builds: end of facet 0
#pragma unroll
          for (int ks = 0; ks < KSF; ++ks) {
            const int s = f * KSF + ks;
            R fl[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
              fl[i] = wf[f] * ndot<R>(cnf[f][0], cnf[f][1], cnf[f][2], nq[s % PFL][i * 3 + 0], nq[s % PFL][i * 3 + 1],
                                      nq[s % PFL][i * 3 + 2]);
            if (s + PFL < NS) {
              const int f1 = (s + PFL) / KSF, k1 = (s + PFL) % KSF;
              load_trace<SYM, GHOST>(np[f1] + noff[f1][k1], gh[f1], fax[f1], nq[s % PFL]);
            }
#pragma unroll
            for (int t = 0; t < MTT; ++t) {
              const R a = sAL[((f * MTT + t) * KSF + ks) * 64 + lo];
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                if (t < MTF)
                  acc[i][t < MTF ? t : 0] = MFMA64(a, fl[i], acc[i][t < MTF ? t : 0]);
                else
                  accs[i][t < MTF ? 0 : t - MTF] = MFMA4(a, fl[i], accs[i][t < MTF ? 0 : t - MTF]);
              }
            }
          }
        }
      }

    };
    do_volume();
    STAMP(st2);
    SG_PRIO(SG_PRIO_LIFT);
    do_lifts();

    SG_PRIO(SG_PRIO_EPI);
    STAMP(st3);
    // ---- sponge: -M^-1 int sigma phi_a phi_b u_abs[b][i] (elastic.py:207-208) on the lanes whose cell carries sigma.
    // A sigma that is one value on all nodes of the cell - the piecewise-constant sponges of the reference's problem
    // scripts - makes that -sigma u_abs at the node itself (StageArgs::sponge_sigma holds the value).  A varying one has a
    // matrix B_e (sponge_sigma = NaN, sponge_slot): B_e u_abs of those cells is computed by a launch of its own before the
    // stage (StageArgs::sponge_pre, kernels.hip sponge_pre_kernel) - a matrix loop here would hold the whole wave for the
    // sake of one lane, and an item that straddles the edge of a sponge strip has such a lane.  Either way a lane reads
    // one value per row and component: same loads, different base, stride and factor.  With the in-place combine u_abs
    // may be `out`: every lane of a cell sits in this wave and the wave runs in program order, so the reads below precede
    // the writes of the epilogue (and the pre-pass ran before the stage).
    if (A.sponge_sigma != nullptr) {
      if (!FUSED && L.active) sig = (R)A.sponge_sigma[e];
      int slot = -1;
      if (sig != sig) slot = A.sponge_slot[e];
      const bool sp_here = sig != (R)0 && !sig_folded;      // (NaN != 0: the lanes with a matrix count)
      if (__any(sp_here)) {
        if (sp_here) {
          constexpr int NR = 4 * MTF + NSM;     // row-quads of this lane: node 4 r + q
          const bool dense = slot >= 0;
          // (the pre-pass results are lines like the fields': slot = their item * 16 + the cell's column)
          const R* pb = dense ? reinterpret_cast<const R*>(A.sponge_pre) + ((long)(slot >> 4) * ND * 3 + q * 3) * 16 + (slot & 15)
                              : reinterpret_cast<const R*>(A.uabs) + ub_q;
          constexpr int es = 16;
          const R sc = dense ? (R)1 : sig;
          R uo[NR][3];
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int o = (4 * r + q < ND) ? 4 * r * 3 * es : 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) uo[r][i] = pb[o + i * es];
          }
#pragma unroll
          for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int i = 0; i < 3; ++i)
              if (4 * r + q < ND) {
                const R v = sc * uo[r][i];
                if (r < 4 * MTF)
                  acc[i][r >> 2][r & 3] -= v;
                else
                  accs[i][r - 4 * MTF] -= v;
              }
        }
      }
    }

    // ---- epilogue (MODE 1: u = c_self*u + c_aux*uh1 + c_new*rhs in place, elastic.py:341-345).
    //      All loads first (results built in place in the accumulators), all stores last: see G.
    if (FUSED && early) {
#pragma unroll
      for (int t = 0; t < MTF; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i][t][reg] = cn * acc[i][t][reg];
#pragma unroll
      for (int t = 0; t < NSM; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i) accs[i][t] = cn * accs[i][t];
    } else if (FUSED) {
      // one memory latency per item instead of one per row tile (the lifts' registers are free by now)
      fetch_old();
#pragma unroll
      for (int t = 0; t < MTF; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            R v = SELF ? cs * po[t][reg][i] + ca * pa[t][reg][i] + cn * acc[i][t][reg] : ca * pa[t][reg][i] + cn * acc[i][t][reg];
            asm volatile("" : "+v"(v));
            acc[i][t][reg] = v;
          }
#pragma unroll
      for (int t = 0; t < NSM; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          accs[i][t] = SELF ? cs * pos[t][i] + ca * pas[t][i] + cn * accs[i][t] : ca * pas[t][i] + cn * accs[i][t];
          asm volatile("" : "+v"(accs[i][t]));
        }
    }
    if (L.active) {
#pragma unroll
      for (int t = 0; t < MTF; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const long o = ub_q + (long)(16 * t + 4 * reg) * 3 * 16;
          if (RT<R>::SMALL || (16 * t + 4 * reg + q < ND)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) ST_STREAM(&out[o + i * 16], acc[i][t][reg]);
          }
        }
#pragma unroll
      for (int t = 0; t < NSM; ++t) {
        const int a = 16 * MTF + 4 * t + q;
        if (a < ND) {
          const long o = ub_q + (long)(16 * MTF + 4 * t) * 3 * 16;
#pragma unroll
          for (int i = 0; i < 3; ++i) ST_STREAM(&out[o + i * 16], accs[i][t]);
        }
      }
    }
    STAMP(st4);
    STAMP_ACC;
  }
  STAMP_FLUSH;
}

template <typename R, int P, int SYM>
static int launch_ps(int kind, const StageArgs& a, hipStream_t s) {
  // persistent grid, at most 2 blocks per CU; a multiple of 8 (one item range per XCD label)
  // ... and no more blocks than there are items for their four waves (small blocks - the reference's own 3-D sweeps run
  // N <= 8 - are launch-bound: every block copies the operator tiles into LDS before its first item)
  unsigned nblk = (unsigned)(a.grid_blocks > 0 ? a.grid_blocks : 512);
  if (a.nitems > 0 && !a.spread) {
    const unsigned need = (((unsigned)a.nitems + 3u) / 4u + 7u) / 8u * 8u;
    nblk = need < nblk ? need : nblk;
  }
  const dim3 grid(nblk), block(256);
  if (kind == 0) {
    // blocks without neighbour blocks never meet a packed remote trace: GHOST = 0 instantiation
    bool ghosts = false;
    for (int sd = 0; sd < 6; ++sd) ghosts = ghosts || (a.ghost[sd] != nullptr);
    if (a.mode == 0) {
      if (ghosts)
        SG_LAUNCH((mfma_stage_F<R, P, 0, SYM, 1>), grid, block, s, a, a);
      else
        SG_LAUNCH((mfma_stage_F<R, P, 0, SYM, 0>), grid, block, s, a, a);
    } else if (a.mode == 2) {
      if (ghosts)
        SG_LAUNCH((mfma_stage_F<R, P, 2, SYM, 1>), grid, block, s, a, a);
      else
        SG_LAUNCH((mfma_stage_F<R, P, 2, SYM, 0>), grid, block, s, a, a);
    } else {
      if (ghosts)
        SG_LAUNCH((mfma_stage_F<R, P, 1, SYM, 1>), grid, block, s, a, a);
      else
        SG_LAUNCH((mfma_stage_F<R, P, 1, SYM, 0>), grid, block, s, a, a);
    }
  } else {
    if constexpr (sizeof(R) == 8 && P >= 3) {
      if (a.fragQ != nullptr) {      // factorised volume term (StageArgs::fragQ; a.fragV then holds the P_r tiles)
        if (a.mode == 0)
          SG_LAUNCH((mfma_stage_G<R, P, 0, SYM, 1>), grid, block, s, a, a);
        else
          SG_LAUNCH((mfma_stage_G<R, P, 1, SYM, 1>), grid, block, s, a, a);
        return (int)hipGetLastError();
      }
    }
    if (a.mode == 0)
      SG_LAUNCH((mfma_stage_G<R, P, 0, SYM, 0>), grid, block, s, a, a);
    else
      SG_LAUNCH((mfma_stage_G<R, P, 1, SYM, 0>), grid, block, s, a, a);
  }
  return (int)hipGetLastError();
}

template <typename R, int P>
static int launch_p(int kind, const StageArgs& a, hipStream_t s) {
  return a.sym ? launch_ps<R, P, 1>(kind, a, s) : launch_ps<R, P, 0>(kind, a, s);
}

// --------------------------------------------------------------------------------------------
//  Sponge of the cells whose sigma is affine in the reference coordinates (kernels.hpp launch_sponge_pre_affine has the
//  family-independent form): sp[slot][a][i] = s_0 u_i[a] + sum_k s_k (X_k u_i)[a] with the element-constant X_k as row
//  tiles in LDS and the cells' own values as B operands straight from their lines - the first product of a G stage with
//  other matrices; the per-cell coefficients s_k scale the accumulator COLUMNS (a lane's cell) as the results come out,
//  and s_0 u needs no load: accumulator row 4 m + q of a lane is the node of its B operand at k-step m.
// --------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(512) void sponge_affine_mfma(const double* __restrict__ uabs, const double* __restrict__ fragX,
                                                          const int32_t* __restrict__ items, const int32_t* __restrict__ item_slots,
                                                          const double* __restrict__ coef, double* __restrict__ sp, int nitems) {
  using M = MG<P, double>;
  typedef RT<double>::v4 d4;
  constexpr int ND = M::ND, KS = M::KS, MTF = M::MTF, NSM = M::NSM, MTT = M::MTT, S4 = M::S4;
  static_assert(KS == S4, "k-step m and row-quad m name the same nodes");
  __shared__ double sX[3 * MTT * KS * 64];
  copy_to_lds<3 * MTT * KS * 64, double, 512>(sX, fragX);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, w = lane & 15;
  // A unit of work is ONE velocity component of one item: 9 B rows, 9 + 9 accumulator values - 108 registers at degree 4, so
  // four waves per SIMD hide the unit's two dependent load latencies (slots -> coefficients, rows) behind each other's matrix
  // work.  (A wave per whole item - 27 rows, 81 accumulators, two waves per SIMD - took 0.50 ms per launch for 48 000 items
  // whose matrix work is 0.07 ms; the same wave with the next item's operands requested ahead spilled and took 0.9 ms.)
  const long nunits = (long)nitems * 3;
  for (long un = (long)blockIdx.x * 8 + wave; un < nunits; un += (long)gridDim.x * 8) {
    const long n = un / 3;
    const int i = (int)(un - n * 3);
    const int sl = item_slots[n * 16 + w];
    double sc[4] = {0.0, 0.0, 0.0, 0.0};
    if (sl >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] = coef[(long)sl * 4 + k];
    }
    const double* own = uabs + ((long)items[n] * ND * 3 + i) * 16 + w;
    double ub[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ub[ks] = own[((4 * ks + q < ND) ? 4 * ks + q : 0) * 3 * 16];      // padded rows meet zero operator columns
    int lo = lane;      // opaque: keeps the item-invariant tile reads inside the loop (they would not fit the registers)
    asm volatile("" : "+v"(lo));
    double r[S4];
#pragma unroll
    for (int m = 0; m < S4; ++m) r[m] = sc[0] * ub[m];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      d4 acc[M::MTFA];
      double accs[M::NSMA];
#pragma unroll
      for (int t = 0; t < MTF; ++t) acc[t] = d4{0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < NSM; ++t) accs[t] = 0.0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < MTT; ++t) {
          const double a = sX[((k * MTT + t) * KS + ks) * 64 + lo];
          if (t < MTF)
            acc[t < MTF ? t : 0] = RT<double>::big(a, ub[ks], acc[t < MTF ? t : 0]);
          else
            accs[t < MTF ? 0 : t - MTF] = RT<double>::small(a, ub[ks], accs[t < MTF ? 0 : t - MTF]);
        }
      const double sk = sc[1 + k];
#pragma unroll
      for (int t = 0; t < MTF; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) r[4 * t + reg] += sk * acc[t][reg];
#pragma unroll
      for (int t = 0; t < NSM; ++t) r[4 * MTF + t] += sk * accs[t];
    }
    // line layout (slot = item' * 16 + w): whole 128-byte lines; the columns of cells without an affine sigma belong to
    // nobody or to the matrix pre-pass (another kernel, same stream): masked
    if (sl >= 0) {
      double* rec = sp + ((long)(sl >> 4) * ND * 3 + i) * 16 + w;
#pragma unroll
      for (int m = 0; m < S4; ++m) {
        const int a = 4 * m + q;
        if (a < ND) rec[a * 3 * 16] = r[m];
      }
    }
  }
}

// blocks of eight waves the device holds of the degree's instantiation (asked from the runtime once: registers and the tiles'
// LDS decide - two per CU at degree 4).  sg_set_absorption asks at set-up (prepare_...), i.e. outside any stream capture.
static int sponge_affine_resident(int P) {
  static int resident[5] = {0, 0, 0, 0, 0};
  if (P < 1 || P > 4) return 0;
  if (resident[P] == 0) {
    int per_cu = 0, dev = 0, ncu = 0;
    const void* k = P == 1 ? (const void*)sponge_affine_mfma<1> : P == 2 ? (const void*)sponge_affine_mfma<2>
                  : P == 3 ? (const void*)sponge_affine_mfma<3> : (const void*)sponge_affine_mfma<4>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 512, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
      ncu = 256;
    resident[P] = per_cu * ncu;
  }
  return resident[P];
}
int prepare_sponge_affine_mfma(int P) { return sponge_affine_resident(P) > 0 ? 0 : -1; }

int launch_sponge_affine_mfma(int P, const void* uabs, const double* fragX, const int32_t* items, const int32_t* item_slots,
                              const double* coef, void* sp, int32_t nitems, void* stream) {
  if (nitems <= 0) return 0;
  // a persistent grid of exactly the resident blocks: more would run a second, partly empty round
  const int resident = sponge_affine_resident(P);
  if (resident <= 0) return -1;
  long blocks = ((long)nitems * 3 + 7) / 8;
  if (blocks > resident) blocks = resident;
  const dim3 grid((unsigned)blocks), block(512);
  hipStream_t s = (hipStream_t)stream;
  switch (P) {
    case 1: hipLaunchKernelGGL(sponge_affine_mfma<1>, grid, block, 0, s, (const double*)uabs, fragX, items, item_slots, coef, (double*)sp, nitems); break;
    case 2: hipLaunchKernelGGL(sponge_affine_mfma<2>, grid, block, 0, s, (const double*)uabs, fragX, items, item_slots, coef, (double*)sp, nitems); break;
    case 3: hipLaunchKernelGGL(sponge_affine_mfma<3>, grid, block, 0, s, (const double*)uabs, fragX, items, item_slots, coef, (double*)sp, nitems); break;
    case 4: hipLaunchKernelGGL(sponge_affine_mfma<4>, grid, block, 0, s, (const double*)uabs, fragX, items, item_slots, coef, (double*)sp, nitems); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

int mfma_blocks_per_cu(int P, int f32) {
  if (f32) return mfma_resident_blocks<float, 4>();
  return P == 1 ? mfma_resident_blocks<double, 1>() : (P == 2 ? mfma_resident_blocks<double, 2>() : mfma_resident_blocks<double, 4>());
}

int launch_stage_mfma(int kind, int P, const StageArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.f32) {
    switch (P) {
      case 1: return launch_p<float, 1>(kind, a, s);
      case 2: return launch_p<float, 2>(kind, a, s);
      case 3: return launch_p<float, 3>(kind, a, s);
      case 4: return launch_p<float, 4>(kind, a, s);
    }
    return -1;
  }
  switch (P) {
    case 1: return launch_p<double, 1>(kind, a, s);
    case 2: return launch_p<double, 2>(kind, a, s);
    case 3: return launch_p<double, 3>(kind, a, s);
    case 4: return launch_p<double, 4>(kind, a, s);
  }
  return -1;
}

}  // namespace sg
