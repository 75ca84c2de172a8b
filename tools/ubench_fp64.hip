// Microbenchmark: FP64 MFMA (16x16x4) and FP64 VALU FMA issue rates on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o tools/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, unsigned long long* clk) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
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
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) clk[0] = __builtin_amdgcn_s_memtime() - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double x) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
  double y = 1.0 + 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], y, x);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
  unsigned long long* clk;
  hipMalloc(&clk, 8);
  const int iters = 20000;
  for (int wpb : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 thr = 1 wave/SIMD; 2 blocks/CU = 2 waves/SIMD
    int grid = 256 * wpb;
    float ms = timeit([&] { hipLaunchKernelGGL((k_mfma<8>), dim3(grid), dim3(256), 0, 0, out, iters, clk); });
    unsigned long long hc = 0;
    hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
    printf("   shader clock during mfma: %.0f MHz (block0 cycles %llu)\n", hc / (ms * 1e3), hc);
    double n_inst = (double)iters * 8;  // per wave
    double flops = n_inst * 2048.0 * grid * 4;
    printf("mfma_f64_16x16x4  waves/SIMD=%d  %.3f ms  %.2f TFLOP/s  %.1f ns/inst/wave -> cycles@2.4GHz per SIMD-inst: %.1f\n",
           wpb, ms, flops / ms / 1e9, ms * 1e6 / n_inst, ms * 1e6 / n_inst * 2.4 / wpb);
    ms = timeit([&] { hipLaunchKernelGGL((k_fma<16>), dim3(grid), dim3(256), 0, 0, out, iters, 0.5); });
    n_inst = (double)iters * 16;
    flops = n_inst * 128.0 * grid * 4;
    printf("v_fma_f64         waves/SIMD=%d  %.3f ms  %.2f TFLOP/s  cycles@2.4GHz per SIMD-inst: %.2f\n", wpb, ms,
           flops / ms / 1e9, ms * 1e6 / n_inst * 2.4 / wpb);
  }
  return 0;
}
