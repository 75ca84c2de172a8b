// Can FP64 VALU FMAs and FP64 MFMAs run side by side for more than either alone? (gfx950)
// Waves with (wave id % 2) < nv run v_fma_f64 chains, the others v_mfma_f64_16x16x4 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, int valu_waves_of_4) {
  const int wave = threadIdx.x >> 6;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
  double s = 0;
  if (wave < valu_waves_of_4) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
    }
    for (int i = 0; i < 8; ++i) s += x[i];
  } else {
    d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* out; (void)hipMalloc(&out, 256 * 2048 * 8);
  const int iters = 4000;
  for (int nv = 0; nv <= 4; ++nv) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix, dim3(512), dim3(256), 0, 0, out, iters, nv);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix, dim3(512), dim3(256), 0, 0, out, iters, nv);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per block: nv waves x iters x 128 fma x 64 lanes x 2 flop; (4-nv) waves x iters x 8 mfma x 2048 flop
    double fv = 512.0 * nv * iters * 128 * 64 * 2, fm = 512.0 * (4 - nv) * iters * 8 * 2048;
    printf("VALU waves %d/4: %.3f ms  (if all ran for the whole time: VALU %.1f + MFMA %.1f = %.1f TFLOP/s)\n", nv, ms,
           fv / ms / 1e9, fm / ms / 1e9, (fv + fm) / ms / 1e9);
  }
  return 0;
}
