// Device-side grid barrier against a kernel boundary (gfx950).  The launch-bound blocks (config 1: six
// 5-us launches per step) would need a whole-step kernel with a grid barrier between stages; the L2s of
// the 8 XCDs are not coherent inside a launch, so such a barrier needs agent-scope release/acquire
// (L2 write-back + invalidate), which is what a kernel boundary does anyway.  This measures both:
//   (a) N dependent launches of a tiny kernel that touches `bytes` of data (dispatch-to-dispatch time),
//   (b) one cooperative launch that does the same work N times with a grid barrier in between.
// usage: ubench_gridbarrier [blocks] [bytes per block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                       // release: make this block's writes visible device-wide
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();                       // acquire
  }
  __syncthreads();
}

__global__ void touch(double* a, long per_block, double s) {
  double* p = a + (long)blockIdx.x * per_block;
  for (long i = threadIdx.x; i < per_block; i += blockDim.x) p[i] = p[i] * s + 1.0;
}

__global__ void touch_coop(double* a, long per_block, double s, int n, unsigned* counter) {
  for (int k = 0; k < n; ++k) {
    // read the NEXT block's data of the previous round (cross-block dependence, as a stencil stage has)
    const long src = ((blockIdx.x + 1) % gridDim.x) * per_block, dst = (long)blockIdx.x * per_block;
    double* q = a + (k & 1 ? 0 : (long)gridDim.x * per_block);
    const double* p = a + (k & 1 ? (long)gridDim.x * per_block : 0);
    for (long i = threadIdx.x; i < per_block; i += blockDim.x) q[dst + i] = p[src + i] * s + 1.0;
    grid_barrier(counter, (unsigned)(k + 1) * gridDim.x);
  }
}

int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 1024;
  long per_block = (argc > 2 ? atol(argv[2]) : 8192) / 8;
  const int n = 600;
  double* a;
  unsigned* counter;
  hipMalloc(&a, 2 * blocks * per_block * sizeof(double));
  hipMemset(a, 0, 2 * blocks * per_block * sizeof(double));
  hipMalloc(&counter, sizeof(unsigned));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int k = 0; k < n; ++k) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, 0, a, per_block, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%d blocks x %ld B: %d dependent launches: %.2f us per launch\n", blocks, per_block * 8, n, ms * 1e3 / n);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(counter, 0, sizeof(unsigned));
    double s = 0.5;
    int nn = n;
    void* args[] = {&a, &per_block, &s, &nn, &counter};
    hipEventRecord(e0);
    hipError_t err = hipLaunchCooperativeKernel((const void*)touch_coop, dim3(blocks), dim3(256), args, 0, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (err != hipSuccess) {
      printf("cooperative launch failed: %s\n", hipGetErrorString(err));
      return 0;
    }
  }
  printf("%d blocks x %ld B: one cooperative launch, %d rounds with a grid barrier: %.2f us per round\n", blocks,
         per_block * 8, n, ms * 1e3 / n);
  return 0;
}
