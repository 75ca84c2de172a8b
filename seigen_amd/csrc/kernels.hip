// HIP kernels (gfx950) for the velocity / stress right-hand sides.
//
// One launch evaluates, for every cell of a box of cubes, the whole of what
// Firedrake runs as a cell loop + interior-facet loop (+ exterior-facet loop)
// + inverse-mass mat-vec (seigen/elastic.py:204-219, :358-367), optionally
// fused with the LF4 combine (elastic.py:340-352):
//
//   F:  uh_i  = -sum_r D_r (Jinv_rj T_ij) + sum_f L_f [ (c n)_j {T_ij} ]  - sponge
//   G:  W_ik  = -Jinv_rk (D_r u_i)        + sum_f (c n)_k L_f [ u^_i ]
//       sh_ij = lam d_ij W_kk + mu (W_ij + W_ji)
//
// Generic path: a 256-thread workgroup owns EB consecutive cells; thread
// (cell, node) accumulates all components of its node.  Cell data, numerical
// fluxes, the transposed reference operators and the mesh tables live in LDS.
// TP = 2 (hexahedra, DQ_1..4): the operators are applied SUM-FACTORISED - D_r = I x D1 x I acts along the node's
// line in direction r, L_f is one 1-D lift factor on the facet node with the node's transverse indices - so a
// node costs 3 (P + 1) + 6 terms instead of 3 nd + 6 nf, and the tables in LDS are D1 and lift1 (the dense D_r of
// DQ_4, 3 x 125 x 125 doubles, would not fit).  A.Dt = { D1 [P+1][P+1] row-major, lift1 [2][P+1] } (api.cpp).
#include <hip/hip_runtime.h>

#include <cxxabi.h>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace sg {

// The host-side handle of a __global__ function carries the kernel's own mangled name: dladdr finds it in the library's
// dynamic symbol table, __cxa_demangle gives the spelling rocprofv3 prints.
std::string kernel_name_of(const void* host_fn) {
  Dl_info info;
  if (!dladdr(host_fn, &info) || !info.dli_sname) return "(unnamed kernel)";
  int status = 0;
  char* d = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &status);
  std::string out = (status == 0 && d) ? d : info.dli_sname;
  std::free(d);
  return out;
}


template <int DIM, int P, int TP = 0>
struct Geo : ElemDims<DIM, P, TP> {
  using ElemDims<DIM, P, TP>::ND;
  static constexpr int BLOCK = 256;
  static constexpr int EB_RAW = BLOCK / ND;
  static constexpr int EB = EB_RAW > 32 ? 32 : EB_RAW;  // cells per workgroup
};

__device__ __forceinline__ void decode_cube(long cube, const int n[3], int c[3]) {
  c[0] = (int)(cube % n[0]);
  long t = cube / n[0];
  c[1] = (int)(t % n[1]);
  c[2] = (int)(t / n[1]);
}

// index of a boundary cube within its side's 2-D array
__device__ __forceinline__ long cube2d(int axis, const int c[3], const int n[3]) {
  if (axis == 0) return c[1] + (long)n[1] * c[2];
  if (axis == 1) return c[0] + (long)n[0] * c[2];
  return c[0] + (long)n[0] * c[1];
}

template <int DIM, int P, int KIND, int TP = 0>
__global__ __launch_bounds__(256) void stage_kernel(StageArgs A) {
  using G = Geo<DIM, P, TP>;
  constexpr int ND = G::ND, NF = G::NF, NFACES = G::NFACES, NCLS = G::NCLS, EB = G::EB;
  constexpr int NC = (KIND == 0) ? DIM * DIM : DIM;  // input components per node
  constexpr int NT = 256;

  constexpr bool SF = TP == 2;                     // sum-factorised hexahedra
  constexpr int N1 = P + 1;
  __shared__ double sDt[SF ? N1 * N1 + 2 * N1 : DIM * ND * ND];
  __shared__ double sLt[SF ? 1 : NFACES * NF * ND];
  // sum-factorised hexahedra: results in the field's [cell][node][comp] order, written back by consecutive threads (a
  // thread's own 3 or 9 components sit 24 / 72 bytes from its neighbour's: a third / a ninth of every line per store)
  constexpr int NCO = (KIND == 0) ? DIM : DIM * DIM;
  __shared__ double sOut[SF ? EB * ND * NCO : 1];
  __shared__ double sQ[EB * ND * NC];
  __shared__ double sFlux[EB * NFACES * NF * DIM];
  __shared__ long sElem[EB];
  __shared__ MeshDev sMd;

  const int tid = threadIdx.x;
  for (int i = tid; i < (SF ? N1 * N1 + 2 * N1 : DIM * ND * ND); i += NT) sDt[i] = A.Dt[i];
  if (!SF)
    for (int i = tid; i < NFACES * NF * ND; i += NT) sLt[i] = A.Lt[i];
  {
    const int* src = reinterpret_cast<const int*>(A.md);
    int* dst = reinterpret_cast<int*>(&sMd);
    for (int i = tid; i < (int)(sizeof(MeshDev) / sizeof(int)); i += NT) dst[i] = src[i];
  }
  __syncthreads();

  auto write_back = [&]() {
    __syncthreads();
    for (int idx = tid; idx < EB * ND * NCO; idx += NT) {
      const int el2 = idx / (ND * NCO);
      const long g2 = sElem[el2];
      if (g2 < 0) continue;
      const long o = g2 * (ND * NCO) + (idx - el2 * (ND * NCO));
      if (A.mode == 0) {
        A.out[o] = sOut[idx];
      } else {
        double cs = A.c_self, ca = A.c_aux, cn = A.c_new;
        if (KIND == 0 && A.rho2 != nullptr) {  // per-cell density (kernels.hpp)
          cs = A.rho2[2 * g2];
          ca *= A.rho2[2 * g2 + 1];
          cn *= A.rho2[2 * g2 + 1];
        }
        // (a fused G stage has no second operand: out = c_self out + c_new rhs, stages.cpp)
        // (a.mode 2 - UTEMP, no self term - does not read `out`)
        A.out[o] = KIND == 0 ? (A.mode == 2 ? 0.0 : cs * A.out[o]) + ca * A.aux[o] + cn * sOut[idx] : cs * A.out[o] + cn * sOut[idx];
      }
    }
  };
  const long ncube_box = (long)A.box_n[0] * A.box_n[1] * A.box_n[2];
  const long nelem_box = ncube_box * NCLS;
  const long nbatch = (nelem_box + EB - 1) / EB;

  for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    __syncthreads();
    if (tid < EB) {
      long eb = batch * EB + tid;
      long g = -1;
      if (eb < nelem_box) {
        long cb = eb / NCLS;
        int cls = (int)(eb % NCLS);
        int c[3];
        decode_cube(cb, A.box_n, c);
        long cube = (c[0] + A.box_o[0]) + (long)sMd.n[0] * ((c[1] + A.box_o[1]) + (long)sMd.n[1] * (c[2] + A.box_o[2]));
        g = cube * NCLS + cls;
      }
      sElem[tid] = g;
    }
    __syncthreads();

    // ---- stage the cells' own data -------------------------------------------------
    for (int idx = tid; idx < EB * ND * NC; idx += NT) {
      int el = idx / (ND * NC);
      int rem = idx - el * (ND * NC);
      long g = sElem[el];
      sQ[idx] = (g >= 0) ? A.in[g * (ND * NC) + rem] : 0.0;
    }
    __syncthreads();

    // ---- numerical fluxes at facet nodes --------------------------------------------
    for (int it = tid; it < EB * NFACES * NF; it += NT) {
      int el = it / (NFACES * NF);
      int r2 = it - el * (NFACES * NF);
      int f = r2 / NF;
      int b = r2 - f * NF;
      long g = sElem[el];
      if (g < 0) continue;
      int cls = (int)(g % NCLS);
      long cube = g / NCLS;
      int c[3];
      decode_cube(cube, sMd.n, c);
      const double* own = &sQ[(el * ND + sMd.fnode[f][b]) * NC];
      const double* nbr = nullptr;
      int axis = sMd.nb_axis[cls][f];
      bool physical = false, ghost = false;
      if (axis < 0) {
        long ng = cube * NCLS + sMd.nb_cls[cls][f];
        nbr = A.in + (ng * ND + sMd.nb_node[cls][f][b]) * NC;
      } else {
        int dir = sMd.nb_dir[cls][f];
        int cn = c[axis] + dir;
        if (cn >= 0 && cn < sMd.n[axis]) {
          long stride = (axis == 0) ? 1 : (axis == 1) ? sMd.n[0] : (long)sMd.n[0] * sMd.n[1];
          long ng = (cube + dir * stride) * NCLS + sMd.nb_cls[cls][f];
          nbr = A.in + (ng * ND + sMd.nb_node[cls][f][b]) * NC;
        } else {
          int side = 2 * axis + (dir > 0 ? 1 : 0);
          if (sMd.has_nbr[side]) {
            long slot = cube2d(axis, c, sMd.n) * sMd.halo_per_cube + sMd.face_ord[sMd.nb_cls[cls][f]][sMd.nb_face[cls][f]];
            nbr = A.ghost[side] + (slot * NF + sMd.nb_fnode[cls][f][b]) * DIM;  // packed trace: DIM comps
            ghost = true;
          } else {
            physical = true;
          }
        }
      }
      double* fl = &sFlux[it * DIM];
      if (KIND == 0) {
        // (c n)_j {T_ij}; no ds term in f => traction-free boundary (elastic.py:206)
#pragma unroll
        for (int i = 0; i < DIM; ++i) {
          double s = 0.0;
          if (!physical) {
            // a packed remote trace holds g_i = T_i,axis only; the columns j != axis meet cn_j = 0 on a
            // block side, so any finite value (here g_i again) gives the same flux bit for bit
#pragma unroll
            for (int j = 0; j < DIM; ++j) s += 0.5 * (own[i * DIM + j] + nbr[ghost ? i : i * DIM + j]) * sMd.cn[cls][f][j];
          }
          fl[i] = s;
        }
      } else {
        // u^ = avg(u) on dS, own trace on ds (elastic.py:213-216)
#pragma unroll
        for (int i = 0; i < DIM; ++i) fl[i] = physical ? own[i] : 0.5 * (own[i] + nbr[i]);
      }
    }
    __syncthreads();

    const int el = tid / ND;
    const int a = tid - el * ND;
    const bool active = (el < EB) && (sElem[el < EB ? el : 0] >= 0);
    const long g = active ? sElem[el] : 0;
    const int cls = (int)(g % NCLS);
    // sum-factorised hexahedra: the node's lattice indices; its line along axis r starts at a - ai[r] * st[r]; on facet
    // 2 r + s it meets the facet node with its two transverse indices (lower axis first: the order of MeshDev::fnode,
    // checked at create)
    const int ai[3] = {a % N1, (a / N1) % N1, a / (N1 * N1)};
    constexpr int st[3] = {1, N1, N1 * N1};
    (void)ai;
    (void)st;

    if (KIND == 0) {
      // T~_ir = sum_j Jinv[r][j] T_ij, in place (each thread owns its node)
      if (active) {
        double* t = &sQ[(el * ND + a) * NC];
        double tt[DIM * DIM];
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int r = 0; r < DIM; ++r) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < DIM; ++j) s += sMd.Jinv[cls][r][j] * t[i * DIM + j];
            tt[i * DIM + r] = s;
          }
#pragma unroll
        for (int k = 0; k < DIM * DIM; ++k) t[k] = tt[k];
      }
      __syncthreads();
      double acc[DIM];
#pragma unroll
      for (int i = 0; i < DIM; ++i) acc[i] = 0.0;
      if (active && SF) {
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          const int b0 = a - ai[r] * st[r];
          for (int m = 0; m < N1; ++m) {
            const double d = sDt[ai[r] * N1 + m];
            const double* t = &sQ[(el * ND + b0 + m * st[r]) * NC];
#pragma unroll
            for (int i = 0; i < DIM; ++i) acc[i] -= d * t[i * DIM + r];
          }
          const int bp = r == 0 ? ai[1] + N1 * ai[2] : r == 1 ? ai[0] + N1 * ai[2] : ai[0] + N1 * ai[1];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const double l = sDt[N1 * N1 + s2 * N1 + ai[r]];
            const double* fl = &sFlux[((el * NFACES + 2 * r + s2) * NF + bp) * DIM];
#pragma unroll
            for (int i = 0; i < DIM; ++i) acc[i] += l * fl[i];
          }
        }
      }
      if (active) {
        if (!SF) {
        for (int b = 0; b < ND; ++b) {
          double d[DIM];
#pragma unroll
          for (int r = 0; r < DIM; ++r) d[r] = sDt[(r * ND + b) * ND + a];
          const double* t = &sQ[(el * ND + b) * NC];
#pragma unroll
          for (int i = 0; i < DIM; ++i)
#pragma unroll
            for (int r = 0; r < DIM; ++r) acc[i] -= d[r] * t[i * DIM + r];
        }
#pragma unroll
        for (int f = 0; f < NFACES; ++f)
          for (int b = 0; b < NF; ++b) {
            double l = sLt[(f * NF + b) * ND + a];
            const double* fl = &sFlux[((el * NFACES + f) * NF + b) * DIM];
#pragma unroll
            for (int i = 0; i < DIM; ++i) acc[i] += l * fl[i];
          }
        }
        if (A.sponge_slot != nullptr) {
          int slot = A.sponge_slot[g];
          if (slot >= 0) {
            const double* B = A.sponge_B + ((long)slot * ND + a) * ND;
            const double* ua = A.uabs + g * (ND * DIM);
            for (int b = 0; b < ND; ++b) {
              double bb = B[b];
#pragma unroll
              for (int i = 0; i < DIM; ++i) acc[i] -= bb * ua[b * DIM + i];
            }
          }
        }
      }
      // the fused combine writes `out` in place, and `uabs` may be that same buffer:
      // every sponge read of this cell must land before any of its nodes is overwritten
      if (SF) {
        if (active) {
#pragma unroll
          for (int i = 0; i < DIM; ++i) sOut[(el * ND + a) * DIM + i] = acc[i];
        }
        write_back();       // (its barrier also orders every sponge read of the batch before the in-place writes)
        continue;
      }
      if (A.sponge_slot != nullptr && A.mode != 0) __syncthreads();
      if (active) {
        long o = (g * ND + a) * DIM;
        if (A.mode == 0) {
#pragma unroll
          for (int i = 0; i < DIM; ++i) A.out[o + i] = acc[i];
        } else {
          double cs = A.c_self, ca = A.c_aux, cn = A.c_new;
          if (A.rho2 != nullptr) {  // per-cell density (kernels.hpp)
            cs = A.rho2[2 * g];
            ca *= A.rho2[2 * g + 1];
            cn *= A.rho2[2 * g + 1];
          }
#pragma unroll
          for (int i = 0; i < DIM; ++i) A.out[o + i] = (A.mode == 2 ? 0.0 : cs * A.out[o + i]) + ca * A.aux[o + i] + cn * acc[i];
        }
      }
    } else {
      if (active) {
        double R[DIM][DIM];  // R[i][r] = (D_r u_i)_a
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int r = 0; r < DIM; ++r) R[i][r] = 0.0;
        if (SF) {
#pragma unroll
          for (int r = 0; r < DIM; ++r) {
            const int b0 = a - ai[r] * st[r];
            for (int m = 0; m < N1; ++m) {
              const double d = sDt[ai[r] * N1 + m];
              const double* u = &sQ[(el * ND + b0 + m * st[r]) * NC];
#pragma unroll
              for (int i = 0; i < DIM; ++i) R[i][r] += d * u[i];
            }
          }
        } else {
        for (int b = 0; b < ND; ++b) {
          double d[DIM];
#pragma unroll
          for (int r = 0; r < DIM; ++r) d[r] = sDt[(r * ND + b) * ND + a];
          const double* u = &sQ[(el * ND + b) * NC];
#pragma unroll
          for (int i = 0; i < DIM; ++i)
#pragma unroll
            for (int r = 0; r < DIM; ++r) R[i][r] += d[r] * u[i];
        }
        }
        double W[DIM][DIM];  // W[i][k] = weak d u_i / d x_k
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) s -= sMd.Jinv[cls][r][k] * R[i][r];
            W[i][k] = s;
          }
#pragma unroll
        for (int f = 0; f < NFACES; ++f) {
          double lu[DIM];
#pragma unroll
          for (int i = 0; i < DIM; ++i) lu[i] = 0.0;
          if (SF) {
            const int r = f / 2;
            const int bp = r == 0 ? ai[1] + N1 * ai[2] : r == 1 ? ai[0] + N1 * ai[2] : ai[0] + N1 * ai[1];
            const double l = sDt[N1 * N1 + (f & 1) * N1 + ai[r]];
            const double* fl = &sFlux[((el * NFACES + f) * NF + bp) * DIM];
#pragma unroll
            for (int i = 0; i < DIM; ++i) lu[i] = l * fl[i];
          } else {
          for (int b = 0; b < NF; ++b) {
            double l = sLt[(f * NF + b) * ND + a];
            const double* fl = &sFlux[((el * NFACES + f) * NF + b) * DIM];
#pragma unroll
            for (int i = 0; i < DIM; ++i) lu[i] += l * fl[i];
          }
          }
#pragma unroll
          for (int i = 0; i < DIM; ++i)
#pragma unroll
            for (int k = 0; k < DIM; ++k) W[i][k] += sMd.cn[cls][f][k] * lu[i];
        }
        double lam = A.per_cell ? A.lam[g] : A.lam0;
        double mu = A.per_cell ? A.mu[g] : A.mu0;
        double tr = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) tr += W[k][k];
        long o = (g * ND + a) * (DIM * DIM);
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int j = 0; j < DIM; ++j) {
            double v = mu * (W[i][j] + W[j][i]) + ((i == j) ? lam * tr : 0.0);
            if (SF)
              sOut[(el * ND + a) * (DIM * DIM) + i * DIM + j] = v;
            else if (A.mode == 0)
              A.out[o + i * DIM + j] = v;
            else
              A.out[o + i * DIM + j] = A.c_self * A.out[o + i * DIM + j] + A.c_new * v;
          }
      }
      if (SF) write_back();
    }
  }
}

template <int DIM, int P, int TP = 0>
static int launch_dp(int kind, const StageArgs& a, hipStream_t s) {
  using G = Geo<DIM, P, TP>;
  long nelem = (long)a.box_n[0] * a.box_n[1] * a.box_n[2] * G::NCLS;
  if (nelem <= 0) return 0;
  long nbatch = (nelem + G::EB - 1) / G::EB;
  long grid = nbatch < 256L * 8 ? nbatch : 256L * 8;
  if (kind == 0)
    SG_LAUNCH((stage_kernel<DIM, P, 0, TP>), dim3((unsigned)grid), dim3(256), s, a, a);
  else
    SG_LAUNCH((stage_kernel<DIM, P, 1, TP>), dim3((unsigned)grid), dim3(256), s, a, a);
  return (int)hipGetLastError();
}

// quadrilateral cells (tensor-product element): the same kernel over the tables of build_quad_tables
static int launch_quad(int kind, int P, const StageArgs& a, hipStream_t s) {
  switch (P) {
    case 1: return launch_dp<2, 1, 1>(kind, a, s);
    case 2: return launch_dp<2, 2, 1>(kind, a, s);
    case 3: return launch_dp<2, 3, 1>(kind, a, s);
    case 4: return launch_dp<2, 4, 1>(kind, a, s);
  }
  return -1;
}

template <int DIM>
static int launch_d(int kind, int P, const StageArgs& a, hipStream_t s) {
  switch (P) {
    case 1: return launch_dp<DIM, 1>(kind, a, s);
    case 2: return launch_dp<DIM, 2>(kind, a, s);
    case 3: return launch_dp<DIM, 3>(kind, a, s);
    case 4: return launch_dp<DIM, 4>(kind, a, s);
  }
  return -1;
}

int launch_stage(int kind, int dim, int P, const StageArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.tensor) {
    if (dim == 2) return launch_quad(kind, P, a, s);
    if (dim != 3) return -1;
    switch (P) {                                                        // hexahedra, sum-factorised (TP = 2)
      case 1: return launch_dp<3, 1, 2>(kind, a, s);                    // 8 nodes, facets of 4
      case 2: return launch_dp<3, 2, 2>(kind, a, s);                    // 27 / 9
      case 3: return launch_dp<3, 3, 2>(kind, a, s);                    // 64 / 16
      case 4: return launch_dp<3, 4, 2>(kind, a, s);                    // 125 / 25
    }
    return -1;
  }
  switch (dim) {
    case 1: return launch_d<1>(kind, P, a, s);
    case 2: return launch_d<2>(kind, P, a, s);
    case 3: return launch_d<3>(kind, P, a, s);
  }
  return -1;
}

// ---- halo pack ------------------------------------------------------------------------
// one element (slot, facet node, component) of the packed traces of block side `side`.
// A packed trace has dim components per facet node: the velocity, or - for a stress field - the
// column T_i,axis of the side's axis: the only part of the neighbour's tensor that f's interior-facet
// term `avg(s)*n` (elastic.py:206) uses on an axis-aligned block side, where n = +-e_axis.
// Lanes run over consecutive boundary cubes (x fastest on the y and z sides) for one (facet, facet node, component):
// on the interleaved layouts 16 consecutive lanes then read one 128-byte line.  (The first version ran the lanes over
// the components and nodes of one facet: 64 different lines per wavefront, and every index through 64-bit divisions
// and loads from the device copy of MeshDev - 0.12 ms per launch for 6-12 MB, on the critical path of a split stage:
// the FIRST launch, its pack and the exchange follow one another.)  Everything the kernel needs travels by value.
struct PackArgs {
  int nside;
  int side[6];
  void* out[6];
  int start[7];          // element ranges of the sides within the launch
  int n2[6];             // boundary cubes of each side
  int n[3], nd, nf, ncls, hpc, dim, gw;
  signed char side_cls[6][2], side_face[6][2];
  unsigned char fnode[MAX_FACES][MAX_NF + 1];
};

template <typename T>
__global__ void pack_kernel(const T* field, int ncomp, PackArgs P, int sym) {
  const int total = P.start[P.nside];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int s = 0;
    while (s + 1 < P.nside && idx >= P.start[s + 1]) ++s;
    const int side = P.side[s], axis = side >> 1, hi = side & 1, d = P.dim;
    int r = idx - P.start[s];
    const int c2 = r % P.n2[s];          // boundary cube, fastest
    r /= P.n2[s];
    const int cpt = r % d;
    r /= d;
    const int b = r % P.nf, ord = r / P.nf;
    int c[3] = {0, 0, 0};
    const int n0 = P.n[0], n1 = P.n[1];
    if (axis == 0) {
      c[1] = c2 % n1;
      c[2] = c2 / n1;
    } else if (axis == 1) {
      c[0] = c2 % n0;
      c[2] = c2 / n0;
    } else {
      c[0] = c2 % n0;
      c[1] = c2 / n0;
    }
    c[axis] = hi ? P.n[axis] - 1 : 0;
    const long cube = c[0] + (long)n0 * (c[1] + (long)n1 * c[2]);
    // the (class, facet) with this ordinal on this side
    const int cls = P.side_cls[side][ord], f = P.side_face[side][ord];
    int cs = cpt;
    if (ncomp != d) {  // stress: component (i, axis); symmetric-mode storage keeps only the i <= j lines valid
      const int i = cpt, j = axis;
      cs = (sym && i > j) ? j * d + i : i * d + j;
    }
    const long off = ((((cube / P.gw) * P.ncls + cls) * (long)P.nd + P.fnode[f][b]) * ncomp + cs) * P.gw + cube % P.gw;
    reinterpret_cast<T*>(P.out[s])[(((long)c2 * P.hpc + ord) * P.nf + b) * d + cpt] = field[off];
  }
}

int launch_pack(const MeshDev* md_dev, const MeshDev& mh, const void* field, int ncomp, int nside, const int* sides,
                void* const* outs, int sym, int f32, void* stream) {
  (void)md_dev;
  PackArgs P;
  std::memset(&P, 0, sizeof(P));
  P.nside = 0;
  P.start[0] = 0;
  for (int i = 0; i < nside && P.nside < 6; ++i) {
    const int axis = sides[i] >> 1;
    long n2 = 1;
    for (int a = 0; a < 3; ++a)
      if (a != axis) n2 *= mh.n[a];
    const long total = n2 * mh.halo_per_cube * mh.nf * mh.dim;
    if (total <= 0 || !outs[i]) continue;
    if (P.start[P.nside] + total > 0x7fffffffL) return (int)hipErrorInvalidValue;
    P.side[P.nside] = sides[i];
    P.out[P.nside] = outs[i];
    P.n2[P.nside] = (int)n2;
    P.start[P.nside + 1] = P.start[P.nside] + (int)total;
    P.nside += 1;
  }
  const long total = P.start[P.nside];
  if (total <= 0) return 0;
  for (int a = 0; a < 3; ++a) P.n[a] = mh.n[a];
  P.nd = mh.nd;
  P.nf = mh.nf;
  P.ncls = mh.ncls;
  P.hpc = mh.halo_per_cube;
  P.dim = mh.dim;
  P.gw = mh.gw;
  std::memcpy(P.side_cls, mh.side_cls, sizeof(P.side_cls));
  std::memcpy(P.side_face, mh.side_face, sizeof(P.side_face));
  std::memcpy(P.fnode, mh.fnode, sizeof(P.fnode));
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (f32)
    hipLaunchKernelGGL(pack_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float*)field, ncomp, P, sym);
  else
    hipLaunchKernelGGL(pack_kernel<double>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const double*)field, ncomp, P, sym);
  return (int)hipGetLastError();
}

// ---- sparse source ----------------------------------------------------------------------
template <typename T>
__global__ void source_kernel(T* field, int ncomp, int gw, long nnz, const int64_t* offs, const double* values,
                              double coef, double scale, SrcStep ss) {
  long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= nnz * ncomp) return;
  if (ss.ctr != nullptr) {   // graph replay: this step's slice and weight from the device-side counter
    const int64_t st = *ss.ctr;
    if (!ss.is_static && st >= ss.nsteps) return;
    if (!ss.is_static) values += st * ss.stride;
    if (ss.weights != nullptr) scale = ss.weights[st];
  }
  long k = idx / ncomp;
  int c = (int)(idx - k * ncomp);
  // every node appears once (sg_set_source merges the entries of a node listed twice): a plain read-modify-write
  T* p = &field[offs[k] + (long)c * gw];
  *p = *p + (T)(coef * sg_mul_rounded(scale, values[idx]));
}

// ---- sponge matrices of the cells whose sigma varies: sp[slot][a][i] = sum_b B_slot[a][b] u_abs[cell(slot)][b][i] ----------
// One block per cell; u_abs in the layout of the kernel family (mesh_tables.hpp: gw cells interleaved).  Runs BEFORE the F
// stage's launches (the stage may update u_abs in place); the stage kernels then read nd x dim values per such cell instead
// of its nd x nd matrix - and no wave runs a matrix loop for the sake of one lane.
template <typename T>
__global__ void sponge_pre_kernel(const T* uabs, const double* B, const int32_t* cells, const int32_t* mats, const int32_t* slots, T* sp,
                                  int nd, int dim, int ncls, int gw, int lines) {
  extern __shared__ double s_u[];   // [nd][dim]
  const int slot = slots ? slots[blockIdx.x] : (int)blockIdx.x;   // (slots: the cells with a matrix among cells with an affine sigma)
  const long e = cells[slot], c = e / ncls, k = e - c * ncls;
  const long base = (((c / gw) * ncls + k) * (long)nd) * dim * gw + c % gw;
  for (int j = threadIdx.x; j < nd * dim; j += blockDim.x) s_u[j] = (double)uabs[base + (long)j * gw];
  __syncthreads();
  const double* Bs = B + (long)mats[slot] * nd * nd;     // cells with the same nodal sigma share a matrix
  for (int j = threadIdx.x; j < nd * dim; j += blockDim.x) {
    const int a = j / dim, i = j - a * dim;
    double acc = 0.0;
    for (int b = 0; b < nd; ++b) acc += Bs[a * nd + b] * s_u[b * dim + i];
    // a record per slot, or (lines: slot = item * gw + w) the layout of the fields: [item][node][comp][gw cells]
    if (lines)
      sp[((long)(slot / gw) * nd * dim + j) * gw + slot % gw] = (T)acc;
    else
      sp[(long)slot * nd * dim + j] = (T)acc;
  }
}

int launch_sponge_pre(const void* uabs, const double* B, const int32_t* cells, const int32_t* mats, const int32_t* slots, void* sp,
                      int32_t nslots, int nd, int dim, int ncls, int gw, int lines, int f32, void* stream) {
  if (nslots <= 0) return 0;
  const int threads = nd * dim <= 64 ? 64 : 128;
  const size_t lds = (size_t)nd * dim * sizeof(double);
  if (f32)
    hipLaunchKernelGGL(sponge_pre_kernel<float>, dim3((unsigned)nslots), dim3(threads), lds, (hipStream_t)stream, (const float*)uabs, B,
                       cells, mats, slots, (float*)sp, nd, dim, ncls, gw, lines);
  else
    hipLaunchKernelGGL(sponge_pre_kernel<double>, dim3((unsigned)nslots), dim3(threads), lds, (hipStream_t)stream, (const double*)uabs,
                       B, cells, mats, slots, (double*)sp, nd, dim, ncls, gw, lines);
  return (int)hipGetLastError();
}

// ---- cells whose sigma is AFFINE in the reference coordinates (smooth ramps: every cell of a linear profile) ----------------
// sigma = s_0 + sum_k s_k xi_k makes the cell's sponge matrix B_e = M^-1 int sigma phi_a phi_b = s_0 I + sum_k s_k X_k with the
// ELEMENT-CONSTANT matrices X_k = Mhat^-1 int xi_k phi_a phi_b: dim + 1 numbers per cell instead of an nd x nd matrix (9.8 KB
// at 3-D P4 against 2.5 KB of field data - what made a ramp cost 1.4 ms per F stage on config 3's mesh).  One block works
// on one ITEM at a time - the gw cells of one class whose values share their lines (mesh_tables.hpp) - so u_abs is read in
// whole lines; the X_k sit in LDS for the life of the block (rows in ELL form: simplices dense, tensor-product cells the
// 3 P + 1 entries of a row that are not zero).  Same output as sponge_pre_kernel: sp[slot][a][i], read by the F stage.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void sponge_pre_affine_kernel(const T* __restrict__ uabs, const double* __restrict__ X,
                                                                const int32_t* __restrict__ col, int W, const int32_t* __restrict__ items,
                                                                const int32_t* __restrict__ item_slots, const double* __restrict__ coef,
                                                                T* __restrict__ sp, int nitems, int nd, int gw, int lines) {
  extern __shared__ double s_all[];
  double* sX = s_all;                                   // [DIM][nd][W]
  double* sU = sX + DIM * nd * W;                       // [nd * DIM][gw]   (then the results as [gw][nd * DIM])
  double* sC = sU + nd * DIM * gw;                      // [gw][DIM + 1]
  int32_t* sCol = reinterpret_cast<int32_t*>(sC + gw * (DIM + 1));   // [nd][W] (null col: dense, b = j)
  int32_t* sSlot = sCol + (col ? nd * W : 0);           // [gw]
  for (int j = threadIdx.x; j < DIM * nd * W; j += 256) sX[j] = X[j];
  if (col)
    for (int j = threadIdx.x; j < nd * W; j += 256) sCol[j] = col[j];
  const int rows = nd * DIM, ndW = nd * W;
  for (int n = blockIdx.x; n < nitems; n += gridDim.x) {
    __syncthreads();      // the previous item's results have left (first round: orders the table writes before their reads, with the next)
    const T* src = uabs + (long)items[n] * rows * gw;   // item = (cube group) * ncls + class: its lines are contiguous
    for (int j = threadIdx.x; j < rows * gw; j += 256) sU[j] = (double)src[j];
    for (int j = threadIdx.x; j < gw; j += 256) {
      const int sl = item_slots[(long)n * gw + j];
      sSlot[j] = sl;
#pragma unroll
      for (int k = 0; k <= DIM; ++k) sC[j * (DIM + 1) + k] = sl >= 0 ? coef[(long)sl * (DIM + 1) + k] : 0.0;
    }
    __syncthreads();
    // thread -> (cell w, node a); the results stay in registers until every thread has read its operands
    constexpr int MAXR = 8;     // rounds of (w, a) pairs per thread: gw * nd <= MAXR * 256 (checked by the launcher)
    double res[MAXR][DIM];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int idx = (int)threadIdx.x + r * 256;
      double acc[DIM];
#pragma unroll
      for (int i = 0; i < DIM; ++i) acc[i] = 0.0;
      if (idx < gw * nd) {
        const int w = idx % gw, a = idx / gw;
        if (sSlot[w] >= 0) {
          double c[DIM + 1];
#pragma unroll
          for (int k = 0; k <= DIM; ++k) c[k] = sC[w * (DIM + 1) + k];
          const double* xr = sX + a * W;
          const int32_t* cr = col ? sCol + a * W : nullptr;
          const double* uw = sU + w;
          for (int j = 0; j < W; ++j) {
            const int b = cr ? cr[j] : j;
            double m = c[1] * xr[j];
            if (DIM > 1) m += c[2] * xr[ndW + j];
            if (DIM > 2) m += c[3] * xr[2 * ndW + j];
            const double* ub = uw + b * DIM * gw;
#pragma unroll
            for (int i = 0; i < DIM; ++i) acc[i] += m * ub[i * gw];
          }
#pragma unroll
          for (int i = 0; i < DIM; ++i) acc[i] += c[0] * uw[(a * DIM + i) * gw];
        }
      }
#pragma unroll
      for (int i = 0; i < DIM; ++i) res[r][i] = acc[i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int idx = (int)threadIdx.x + r * 256;
      if (idx < gw * nd) {
        const int w = idx % gw, a = idx / gw;
#pragma unroll
        for (int i = 0; i < DIM; ++i) sU[w * rows + a * DIM + i] = res[r][i];      // [w][a][i]: a cell's record is contiguous
      }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < rows * gw; j += 256) {
      const int w = j / rows, rr = j - w * rows;
      const int sl = sSlot[w];
      // a record per slot, or (lines: slot = item' * gw + column, the 3-D MFMA family) the layout of the fields
      if (sl >= 0) sp[lines ? ((long)(sl / gw) * rows + rr) * gw + sl % gw : (long)sl * rows + rr] = (T)sU[j];
    }
  }
}

// the instantiations that may need more than 64 KB of dynamic LDS (hexahedra DQ_4) are told so ONCE, at set-up time
// (sg_set_absorption) - not at the first launch, which may sit inside a stream capture
int prepare_sponge_pre_affine(int dim, int f32, size_t lds) {
  if (lds <= ((size_t)64 << 10)) return 0;
  const void* k = nullptr;
  if (f32)
    k = dim == 1 ? (const void*)sponge_pre_affine_kernel<float, 1> : dim == 2 ? (const void*)sponge_pre_affine_kernel<float, 2>
                                                                              : (const void*)sponge_pre_affine_kernel<float, 3>;
  else
    k = dim == 1 ? (const void*)sponge_pre_affine_kernel<double, 1> : dim == 2 ? (const void*)sponge_pre_affine_kernel<double, 2>
                                                                               : (const void*)sponge_pre_affine_kernel<double, 3>;
  return (int)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

size_t sponge_pre_affine_lds(int W, int has_col, int nd, int dim, int gw) {
  return ((size_t)dim * nd * W + (size_t)nd * dim * gw + (size_t)gw * (dim + 1)) * sizeof(double) +
         ((has_col ? (size_t)nd * W : 0) + (size_t)gw) * sizeof(int32_t);
}

int launch_sponge_pre_affine(const void* uabs, const double* X, const int32_t* col, int W, const int32_t* items, const int32_t* item_slots,
                             const double* coef, void* sp, int32_t nitems, int nd, int dim, int gw, int lines, int f32, void* stream) {
  if (nitems <= 0) return 0;
  const int threads = 256;
  if (gw * nd > 8 * threads) return (int)hipErrorInvalidValue;      // MAXR of the kernel
  const size_t lds = sponge_pre_affine_lds(W, col != nullptr, nd, dim, gw);
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
      ncu = 256;
  }
  const int per_cu = (int)(((size_t)160 << 10) / (lds + 512)) < 1 ? 1 : (int)(((size_t)160 << 10) / (lds + 512));
  long grid = (long)ncu * (per_cu > 4 ? 4 : per_cu);
  if (grid > nitems) grid = nitems;
#define SG_AFF_LAUNCH(TT, DD)                                                                                                     \
  do {                                                                                                                          \
    auto kern = sponge_pre_affine_kernel<TT, DD>;   /* (more than 64 KB of LDS: prepare_sponge_pre_affine, at set-up) */      \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, (hipStream_t)stream, (const TT*)uabs, X, col, W, items, \
                       item_slots, coef, (TT*)sp, nitems, nd, gw, lines);                                                       \
  } while (0)
  if (f32) {
    if (dim == 1) SG_AFF_LAUNCH(float, 1);
    else if (dim == 2) SG_AFF_LAUNCH(float, 2);
    else SG_AFF_LAUNCH(float, 3);
  } else {
    if (dim == 1) SG_AFF_LAUNCH(double, 1);
    else if (dim == 2) SG_AFF_LAUNCH(double, 2);
    else SG_AFF_LAUNCH(double, 3);
  }
#undef SG_AFF_LAUNCH
  return (int)hipGetLastError();
}

__global__ void step_counter_kernel(int64_t* ctr, int64_t value, int add) { *ctr = add ? *ctr + value : value; }

int launch_step_counter(int64_t* ctr, int64_t value, int add, void* stream) {
  hipLaunchKernelGGL(step_counter_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, value, add);
  return (int)hipGetLastError();
}

int launch_source(void* field, int ncomp, int gw, int64_t nnz, const int64_t* offs, const double* values, double coef,
                  double scale, const SrcStep& ss, int f32, void* stream) {
  if (nnz <= 0) return 0;
  long total = nnz * ncomp;
  if (f32)
    hipLaunchKernelGGL(source_kernel<float>, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       (float*)field, ncomp, gw, (long)nnz, offs, values, coef, scale, ss);
  else
    hipLaunchKernelGGL(source_kernel<double>, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       (double*)field, ncomp, gw, (long)nnz, offs, values, coef, scale, ss);
  return (int)hipGetLastError();
}

// ---- host <-> device layout ------------------------------------------------------------------
// the staging side is always double (the C-ABI's host type); a float field converts on the way
template <typename T>
__global__ void layout_kernel(int dim, int nd, int ncls, int gw, int ncomp, int dir, T* field, double* staging,
                              long cell0, long ncells, int sym, int* flag) {
  const long per_cell = (long)nd * ncomp;
  const long total = ncells * per_cell;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long e = cell0 + idx / per_cell;
    long rem = idx % per_cell;
    long cube = e / ncls;
    int cls = (int)(e % ncls);
    if (dir == 0) {
      long off = (((cube / gw) * ncls + cls) * per_cell + rem) * gw + cube % gw;
      double v = staging[idx];
      field[off] = (T)v;
      if (flag != nullptr && ncomp == dim * dim) {  // tensor upload: report any asymmetry
        int cpt = (int)(rem % ncomp), i = cpt / dim, j = cpt % dim;
        if (i < j && staging[idx - cpt + j * dim + i] != v) *flag = 1;
      }
    } else {
      long r2 = rem;
      if (sym) {  // lower-triangle lines are stale in symmetric mode: deliver the mirror
        int cpt = (int)(rem % ncomp), i = cpt / dim, j = cpt % dim;
        if (i > j) r2 = rem - cpt + j * dim + i;
      }
      long off = (((cube / gw) * ncls + cls) * per_cell + r2) * gw + cube % gw;
      staging[idx] = field[off];
    }
  }
}

int launch_layout(const MeshDev& mh, int ncomp, int dir, void* field, double* staging, int64_t cell0, int64_t ncells,
                  int sym, int* flag, int f32, void* stream) {
  if (ncells <= 0) return 0;
  long total = ncells * (long)mh.nd * ncomp;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (f32)
    hipLaunchKernelGGL(layout_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, mh.dim, mh.nd, mh.ncls,
                       mh.gw, ncomp, dir, (float*)field, staging, (long)cell0, (long)ncells, sym, flag);
  else
    hipLaunchKernelGGL(layout_kernel<double>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, mh.dim, mh.nd, mh.ncls,
                       mh.gw, ncomp, dir, (double*)field, staging, (long)cell0, (long)ncells, sym, flag);
  return (int)hipGetLastError();
}

template <typename T>
__global__ void mirror_kernel(int dim, int nd, int ncls, int gw, T* field, long ncube_pad) {
  const int nc = dim * dim;
  const long total = ncube_pad * ncls * nd * nc;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long w = idx % gw;
    long t = idx / gw;
    int cpt = (int)(t % nc);
    int i = cpt / dim, j = cpt % dim;
    if (i > j) field[idx] = field[(t - cpt + j * dim + i) * gw + w];
  }
}

int launch_mirror(const MeshDev& mh, void* field, int f32, void* stream) {
  long total = mh.ncube_pad * mh.ncls * (long)mh.nd * mh.dim * mh.dim;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (f32)
    hipLaunchKernelGGL(mirror_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, mh.dim, mh.nd, mh.ncls,
                       mh.gw, (float*)field, (long)mh.ncube_pad);
  else
    hipLaunchKernelGGL(mirror_kernel<double>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, mh.dim, mh.nd, mh.ncls,
                       mh.gw, (double*)field, (long)mh.ncube_pad);
  return (int)hipGetLastError();
}

}  // namespace sg
