// How many independent accumulator chains does v_mfma_f64_16x16x4_f64 need? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
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
void run(double* out, int wps) {
  const int iters = 24000 / NACC;
  int grid = 256 * wps;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mfma<NACC>), dim3(grid), dim3(256), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_mfma<NACC>), dim3(grid), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double n_inst = (double)iters * NACC;
  printf("chains=%2d waves/SIMD=%d : %.1f ns per MFMA per wave, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", NACC, wps,
         ms * 1e6 / n_inst, ms * 1e6 / n_inst / wps, n_inst * 2048.0 * grid * 4 / ms / 1e9);
}
int main() {
  double* out; (void)hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
  for (int wps : {1, 2}) {
    run<1>(out, wps); run<2>(out, wps); run<3>(out, wps); run<4>(out, wps); run<6>(out, wps); run<9>(out, wps); run<12>(out, wps);
  }
  return 0;
}
