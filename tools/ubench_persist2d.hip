// What would a whole-step persistent kernel buy the launch-bound 2-D configurations (config 5: 5794 items per stage, six
// dependent stages per step, 11-16 us per launch of which ~3 us are launch floor)?  VERDICT r05 item 6 asks for the
// per-stage fixed cost measured INSIDE a persistent kernel before six real stages are written.  This is that
// measurement on a stand-in with the stage's shape, not on the stage kernels themselves:
//   * an "item" is one wave's unit of work: it reads its own chunk of the input field (OWN lines of 128 B = 16 doubles)
//     and TR lines from each of four neighbour items (i-1, i+1, i-ROW, i+ROW: the facet traces; ROW items = one row of
//     squares), does a little arithmetic and writes OUT lines of the output field;
//   * a "stage" is all items; stage s+1 reads what stage s wrote (ping-pong buffers), so item i of stage s+1 depends on
//     items i-ROW .. i+ROW of stage s - exactly the dependence of the LF4 stages on a structured mesh.
// Mode A (what ships): one launch per stage, replayed as a hipGraph.  Mode B: ONE launch for all stages; waves walk their
// items stage by stage; per (stage, block of BLK items) completion counters - the producer wave stores with agent scope,
// waits for its stores, adds 1 (agent-scope atomic); the consumer polls the counters of the blocks it reads from and loads
// its operands with agent-scope loads (they bypass the non-coherent L2 of another XCD) - the cheapest correct protocol of
// profiles/r04/xcd_barrier_ubench.txt and the guide's flag hand-off.  Mode C: the same with the stages SKEWED over bands of
// rows (diagonal d = SKEW * stage + band), so that what a wave waits for was finished at least one diagonal ago.
// Results of B / C are compared with A's bit for bit.  Every poll loop is bounded: a wave that gives up sets an abort word
// and everybody leaves (no hang).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int OWN = 60;    // lines read from the own chunk (P3 triangle: 10 nodes x 4 + 2 stress comps ... ~7.5 KB per item)
constexpr int TR = 8;      // lines read from each of the four neighbours
constexpr int OUT = 30;    // lines written
constexpr int CH = 64;     // lines per item chunk (8 KB)
constexpr int BLK = 64;    // items per completion counter

struct Args {
  const double* in;
  double* out;
  double* buf[2];
  int nitems, row, nstages, skew, nbands, band_items;
  int own_scope;   // 1: own-chunk loads at workgroup scope (stage-major order only)
  int fence;       // 1: release fence at agent scope (L2 write-back) before the count; 0: agent-scope stores + s_waitcnt only
  int* counters;   // [nstages][nblk]
  int* abort_flag;
  unsigned long long* wait_cycles;   // summed poll cycles, summed item cycles, items
};

template <bool COHERENT>
__device__ __forceinline__ double ld(const double* p) {
  if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
// the own chunk was written by THIS wave a stage ago (stage-major order, static item assignment): its lines are in this
// XCD's L2; only the L1 may still hold the buffer's lines of two stages ago -> workgroup scope (L1 bypass) is enough
template <bool COHERENT>
__device__ __forceinline__ double ld_own(const double* p, int own_scope) {
  if (COHERENT && own_scope) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return ld<COHERENT>(p);
}
template <bool COHERENT>
__device__ __forceinline__ void st(double* p, double v) {
  if (COHERENT)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __builtin_nontemporal_store(v, p);
}

// one item: lane l = (q = l >> 4, w = l & 15) reads line 4 j + q, column w - the access shape of the tile kernels
template <bool COHERENT>
__device__ __forceinline__ void item_body(const double* in, double* out, int i, int nitems, int row, int lane, int own_scope = 0) {
  const int q = lane >> 4, w = lane & 15;
  const double* own = in + (long)i * CH * 16;
  double acc = 0.0;
  double v[OWN / 4];
#pragma unroll
  for (int j = 0; j < OWN / 4; ++j) v[j] = ld_own<COHERENT>(own + (4 * j + q) * 16 + w, own_scope);
  const int nb[4] = {i - 1, i + 1, i - row, i + row};
  double t[4][TR / 4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int n = (nb[f] >= 0 && nb[f] < nitems) ? nb[f] : i;
    const double* p = in + (long)n * CH * 16;
#pragma unroll
    for (int j = 0; j < TR / 4; ++j) t[f][j] = ld<COHERENT>(p + (4 * (j + 2 * f) + q) * 16 + w);
  }
#pragma unroll
  for (int j = 0; j < OWN / 4; ++j) acc = __builtin_fma(v[j], 1.0 / (j + 2), acc);
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int j = 0; j < TR / 4; ++j) acc = __builtin_fma(t[f][j], 0.125 / (f + j + 1), acc);
  acc *= 0.2;      // keeps the values bounded over many stages
  double* o = out + (long)i * CH * 16;
#pragma unroll
  for (int j = 0; j < OUT / 4 + 1; ++j) {
    const int line = 4 * j + q;
    if (line < OUT) st<COHERENT>(o + line * 16 + w, acc + 1e-3 * line);
  }
  // the lines a stage does not write keep their old values in both buffers: copy them through so that the next stage
  // reads defined data whichever buffer it gets
#pragma unroll
  for (int j = OUT / 4; j < CH / 4; ++j) {
    const int line = 4 * j + q;
    if (line >= OUT) st<COHERENT>(o + line * 16 + w, v[(j < OWN / 4) ? j : 0] * 0.5);
  }
}

__global__ __launch_bounds__(256) void stage_kernel(const double* in, double* out, int nitems, int row) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < nitems; i += gridDim.x * 4) item_body<false>(in, out, i, nitems, row, lane);
}

// the order in which a wave visits (stage, item) pairs.  skew = 0: stage-major (all of stage s, then s + 1).
// skew > 0: diagonal-major over bands of rows: diagonal d holds (stage s, band b) with skew * s + b = d.
__global__ __launch_bounds__(256) void persist_kernel(Args A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = gridDim.x * 4, me = blockIdx.x * 4 + wave;
  const int nblk = (A.nitems + BLK - 1) / BLK;
  unsigned long long waited = 0, worked = 0, nit = 0;
  auto run_item = [&](int s, int i) -> bool {
    const long long t0 = __builtin_readcyclecounter();
    if (s > 0) {
      // everything this item reads: items i - row .. i + row of stage s - 1
      const int lo = (i - A.row < 0 ? 0 : i - A.row) / BLK, hi = (i + A.row >= A.nitems ? A.nitems - 1 : i + A.row) / BLK;
      const int* cnt = A.counters + (long)(s - 1) * nblk;
      for (int b = lo; b <= hi; ++b) {
        const int want = (b == nblk - 1) ? A.nitems - b * BLK : BLK;
        int spins = 0;
        while (__hip_atomic_load(cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (++spins > (1 << 20) || __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(A.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    item_body<true>(A.buf[s & 1], A.buf[(s + 1) & 1], i, A.nitems, A.row, lane, A.own_scope);
    // the wave's stores have reached the coherent level before the count says so
    __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) expcnt(0) lgkmcnt(0)
    if (A.fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(A.counters + (long)s * nblk + i / BLK, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t2 = __builtin_readcyclecounter();
    waited += (unsigned long long)(t1 - t0);
    worked += (unsigned long long)(t2 - t1);
    nit += 1;
    return true;
  };
  bool ok = true;
  if (A.skew == 0) {
    for (int s = 0; s < A.nstages && ok; ++s)
      for (int i = me; i < A.nitems && ok; i += W) ok = run_item(s, i);
  } else {
    // every wave walks the same global sequence (diagonal, band, item in band) and takes every W-th entry: the waves move
    // through it together, and a diagonal's bands are independent of each other and of the diagonal before
    const int ndiag = A.skew * (A.nstages - 1) + A.nbands;
    long seq = 0;
    for (int d = 0; d < ndiag && ok; ++d)
      for (int b = d % A.skew; b < A.nbands && ok; b += A.skew) {
        const int s = (d - b) / A.skew;
        if (s < 0 || s >= A.nstages) continue;
        const int i0 = b * A.band_items, n = (i0 + A.band_items > A.nitems ? A.nitems - i0 : A.band_items);
        // my entries of this band: global positions seq .. seq + n - 1
        long first = (me - seq % W + W) % W;
        for (long k = first; k < n && ok; k += W) ok = run_item(s, i0 + (int)k);
        seq += n;
      }
  }
  if (lane == 0) {
    atomicAdd(&A.wait_cycles[0], waited);
    atomicAdd(&A.wait_cycles[1], worked);
    atomicAdd(&A.wait_cycles[2], nit);
  }
}

int main(int argc, char** argv) {
  const int nitems = argc > 1 ? atoi(argv[1]) : 5794;      // config 5: 2897 groups x 2 classes
  const int row = argc > 2 ? atoi(argv[2]) : 50;           // 24 groups of 16 squares + 1, two classes
  const int steps = argc > 3 ? atoi(argv[3]) : 8;
  const int nstages = 6 * steps;
  const size_t n = (size_t)nitems * CH * 16;
  double *a, *b, *ra, *rb;
  CHECK(hipMalloc(&a, n * 8));
  CHECK(hipMalloc(&b, n * 8));
  CHECK(hipMalloc(&ra, n * 8));
  CHECK(hipMalloc(&rb, n * 8));
  std::vector<double> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = 1.0 + 1e-3 * (double)((i * 2654435761u) % 1000);
  int dev = 0, ncu = 0, per_cu = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel, 256, 0));
  const int resident = per_cu * ncu;
  printf("items %d (%.1f MB read + %.1f MB written per stage), row %d, %d stages; persistent kernel: %d blocks per CU -> %d resident blocks\n",
         nitems, nitems * (OWN + 4 * TR) * 128 / 1e6, nitems * CH * 128 / 1e6, row, nstages, per_cu, resident);
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  // ---- A: one launch per stage, as a graph of `nstages` dependent launches -------------------------------------------
  for (int grid : {768, 1024, 1456}) {
    CHECK(hipMemcpy(ra, h.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(rb, h.data(), n * 8, hipMemcpyHostToDevice));
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int st = 0; st < nstages; ++st)
      hipLaunchKernelGGL(stage_kernel, dim3(grid), dim3(256), 0, s, (st & 1) ? rb : ra, (st & 1) ? ra : rb, nitems, row);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));      // warm-up (results: 2 x nstages stages from the initial data)
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(ra, h.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(rb, h.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0, s));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("A  graph of launches, grid %5d          : %7.2f us per stage\n", grid, ms * 1e3 / nstages);
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
  }
  std::vector<double> ref(n), got(n);
  CHECK(hipMemcpy(ref.data(), (nstages & 1) ? rb : ra, n * 8, hipMemcpyDeviceToHost));

  // ---- B / C: one persistent launch --------------------------------------------------------------------------------
  const int nblk = (nitems + BLK - 1) / BLK;
  int *counters, *abortf;
  unsigned long long* wc;
  CHECK(hipMalloc(&counters, (size_t)nstages * nblk * sizeof(int)));
  CHECK(hipMalloc(&abortf, sizeof(int)));
  CHECK(hipMalloc(&wc, 3 * sizeof(unsigned long long)));
  struct Variant { int skew, rows_per_band, grid, fence, own_scope; };
  std::vector<Variant> vars;
  for (int fence : {1, 0}) {
    for (int grid : {resident, resident / 2}) vars.push_back({0, 0, grid, fence, 0});
    for (int skew : {3, 4}) vars.push_back({skew, 4, resident, fence, 0});
  }
  for (int grid : {resident, resident / 2, resident / 4}) vars.push_back({0, 0, grid, 0, 1});
  for (const Variant& v : vars) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(b, h.data(), n * 8, hipMemcpyHostToDevice));
      CHECK(hipMemset(counters, 0, (size_t)nstages * nblk * sizeof(int)));
      CHECK(hipMemset(abortf, 0, sizeof(int)));
      CHECK(hipMemset(wc, 0, 3 * sizeof(unsigned long long)));
      Args A;
      A.buf[0] = a;
      A.buf[1] = b;
      A.nitems = nitems;
      A.row = row;
      A.nstages = nstages;
      A.skew = v.skew;
      A.fence = v.fence;
      A.own_scope = v.own_scope;
      A.band_items = v.rows_per_band > 0 ? v.rows_per_band * row : nitems;
      A.nbands = (nitems + A.band_items - 1) / A.band_items;
      A.counters = counters;
      A.abort_flag = abortf;
      A.wait_cycles = wc;
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(persist_kernel, dim3(v.grid), dim3(256), 0, s, A);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      int ab = 0;
      unsigned long long w3[3];
      CHECK(hipMemcpy(&ab, abortf, sizeof(int), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(w3, wc, sizeof(w3), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(got.data(), (nstages & 1) ? b : a, n * 8, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < n; ++i) bad += got[i] != ref[i];
      if (rep == 1)
        printf("%s skew %d, %2d rows per band, grid %5d, %s : %7.2f us per stage   waited %6.0f + worked %6.0f cycles per item   %s%s\n",
               v.skew ? "C  persistent, skewed," : "B  persistent, stage-major,", v.skew, v.rows_per_band, v.grid,
               v.fence ? "release fence" : (v.own_scope ? "no fence, own rows at workgroup scope" : "no fence     "), ms * 1e3 / nstages,
               w3[2] ? (double)w3[0] / w3[2] : 0.0, w3[2] ? (double)w3[1] / w3[2] : 0.0, ab ? "ABORTED (a poll ran out) " : "",
               bad ? "RESULT DIFFERS" : "bitwise = A");
    }
  }
  return 0;
}
