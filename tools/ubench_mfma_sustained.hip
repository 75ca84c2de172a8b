// Sustained FP64 matrix rate of the two gfx950 shapes over tens of milliseconds (power management needs about a
// millisecond to react; tools/ubench_mfma4.hip runs 0.2 ms per launch): v_mfma_f64_16x16x4_f64 against
// v_mfma_f64_4x4x4_4b_f64 with the same number of flops, 2 waves per SIMD on every CU, ten back-to-back launches each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_big(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_small(double* out, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same with a fresh A operand from LDS for every matrix instruction (what a kernel built on small tiles would do)
template <int NACC>
__global__ __launch_bounds__(256) void k_small_lds(double* out, int iters) {
  __shared__ double tab[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) tab[i] = 1e-3 * i;
  __syncthreads();
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double b = threadIdx.x * 2e-3 + 1.0;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(tab[((it + i) & 63) * 64 + lane], b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, double flop_per_inst, int nacc, int iters, double* out) {
  const int grid = 256 * 2;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("%-34s", name);
  for (int rep = 0; rep < 10; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf(" %5.1f", (double)iters * nacc * flop_per_inst * grid * 4 / ms / 1e9);
  }
  printf("  TFLOP/s per launch (%.0f ms each)\n", 0.0);
  fflush(stdout);
}

int main() {
  double* out;
  (void)hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
  for (int round = 0; round < 2; ++round) {
    run("16x16x4, 8 chains", k_big<8>, 2048.0, 8, 40000, out);         // ~ 15 ms per launch
    run("4x4x4_4b, 8 chains", k_small<8>, 512.0, 8, 160000, out);      // same flops
    run("4x4x4_4b, 12 chains", k_small<12>, 512.0, 12, 106667, out);
    run("4x4x4_4b, 12 chains, A from LDS", k_small_lds<12>, 512.0, 12, 106667, out);
  }
  return 0;
}
