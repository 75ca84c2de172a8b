// A grid barrier confined to ONE XCD against a kernel boundary (gfx950).  Round 2 measured the agent-scope barrier a
// whole-step kernel would need across the 8 XCDs (L2 write-back + invalidate: 3.5-40 x a kernel boundary,
// tools/ubench_gridbarrier.hip).  Blocks are dealt to the XCDs round-robin (blockIdx % 8), so a launch of 8 nb blocks
// whose blocks with blockIdx % 8 != 0 leave at once keeps nb working blocks on one XCD: they share ONE L2, and a barrier
// among them needs no L2 maintenance - stores complete at the L2 (write-through L1: s_waitcnt vmcnt(0)), one relaxed
// atomic per block at the L2, a poll that bypasses the L1, and an invalidate of the CU's own L1 (buffer_inv sc0).
//   (a) N dependent launches of a tiny kernel that rewrites `bytes` per block (dispatch-to-dispatch time, as a graph),
//   (b) one launch that does the same N times with the XCD-local barrier in between; the result is CHECKED (every
//       round reads what the neighbour block wrote in the round before).
// usage: ubench_xcdbarrier [working blocks] [bytes per block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

#ifndef INV
#define INV "buffer_inv sc1"
#endif
__device__ __forceinline__ void xcd_barrier(unsigned* counter, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores have reached the L2 (write-through L1)
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
      __builtin_amdgcn_s_sleep(2);      // bounded: a block that is not co-resident must not hang the GPU
  }
  __syncthreads();
  asm volatile(INV ::: "memory");                         // drop this CU's L1 lines: the next loads come from the L2
}

__global__ void touch(double* a, long per_block, double s) {
  double* p = a + (long)blockIdx.x * per_block;
  for (long i = threadIdx.x; i < per_block; i += blockDim.x) p[i] = p[i] * s + 1.0;
}

__global__ void touch_xcd(double* a, long per_block, double s, int n, unsigned* counter, unsigned* xccs) {
  if (blockIdx.x & 7) return;
  const int b = blockIdx.x >> 3, nb = gridDim.x >> 3;
  if (threadIdx.x == 0) atomicOr(xccs, 1u << xcc_id());
  for (int k = 0; k < n; ++k) {
    const long src = ((b + 1) % nb) * per_block, dst = (long)b * per_block;
    double* q = a + (k & 1 ? 0 : (long)nb * per_block);
    const double* p = a + (k & 1 ? (long)nb * per_block : 0);
#ifdef AGENT_LOADS   // no invalidate at all: the loads themselves go past the L1 (agent-scope relaxed: global_load ... sc1)
    for (long i = threadIdx.x; i < per_block; i += blockDim.x)
      q[dst + i] = __hip_atomic_load(&p[src + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * s + 1.0;
#else
    for (long i = threadIdx.x; i < per_block; i += blockDim.x) q[dst + i] = p[src + i] * s + 1.0;
#endif
    xcd_barrier(counter, (unsigned)(k + 1) * nb);
  }
}

int main(int argc, char** argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 50;
  long per_block = (argc > 2 ? atol(argv[2]) : 8192) / 8;
  const int n = 600;
  double* a;
  unsigned *counter, *xccs;
  hipMalloc(&a, 2 * nb * per_block * sizeof(double));
  hipMalloc(&counter, sizeof(unsigned));
  hipMalloc(&xccs, sizeof(unsigned));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  hipStream_t st;
  hipStreamCreate(&st);
  // (a) as a graph of n dependent launches
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < n; ++k) hipLaunchKernelGGL(touch, dim3(nb), dim3(256), 0, st, a, per_block, 0.5);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%d blocks x %ld B: graph of %d dependent launches: %.2f us per launch\n", nb, per_block * 8, n, ms * 1e3 / n);
  // (b) one launch, XCD-local barrier
  for (int rep = 0; rep < 2; ++rep) {
    hipMemsetAsync(a, 0, 2 * nb * per_block * sizeof(double), st);
    hipMemsetAsync(counter, 0, sizeof(unsigned), st);
    hipMemsetAsync(xccs, 0, sizeof(unsigned), st);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(touch_xcd, dim3(8 * nb), dim3(256), 0, st, a, per_block, 0.5, n, counter, xccs);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned mask = 0;
  hipMemcpy(&mask, xccs, sizeof(unsigned), hipMemcpyDeviceToHost);
  // expected value: x_{k+1} = x_k / 2 + 1 from 0, the same for every element whichever block wrote it
  double want = 0.0;
  for (int k = 0; k < n; ++k) want = want * 0.5 + 1.0;
  std::vector<double> h((size_t)nb * per_block);
  hipMemcpy(h.data(), a + ((n - 1) & 1 ? 0 : (long)nb * per_block), h.size() * sizeof(double), hipMemcpyDeviceToHost);
  long bad = 0;
  for (double v : h) bad += (v != want);
  printf("%d blocks x %ld B: one launch, %d rounds with an XCD-local barrier: %.2f us per round; XCC mask 0x%x (%s), %ld wrong values\n",
         nb, per_block * 8, n, ms * 1e3 / n, mask, (mask & (mask - 1)) ? "MORE THAN ONE XCD" : "one XCD", bad);
  return 0;
}
