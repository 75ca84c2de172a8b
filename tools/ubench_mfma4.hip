// v_mfma_f64_4x4x4_4b_f64 on gfx950: issue rate next to the 16x16x4 shape, and the lane layout of
// its operands (probed with one-hot A operands and lane-coded B operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int MIX>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters) {
  double acc[NACC];
  d4 big[2] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    if (MIX) {
      big[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, big[0], 0, 0, 0);
      big[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, big[1], 0, 0, 0);
    }
  }
  double s = big[0][0] + big[1][1];
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int MIX>
void run(double* out, int wps) {
  const int iters = 24000 / NACC;
  int grid = 256 * wps;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<NACC, MIX>), dim3(grid), dim3(256), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<NACC, MIX>), dim3(grid), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("4x4x4 chains=%2d mix16=%d waves/SIMD=%d : %.1f ns per loop-iteration per SIMD (%d small%s)\n", NACC, MIX, wps,
         ms * 1e6 / iters / wps, NACC, MIX ? " + 2 big" : "");
}
__global__ void k_probe(const double* a, const double* b, double* d) {
  d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}
int main() {
  double* out; (void)hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
  for (int wps : {1, 2}) {
    run<4, 0>(out, wps); run<8, 0>(out, wps); run<12, 0>(out, wps); run<8, 1>(out, wps);
  }
  double *da, *db, *dd;
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
  std::vector<double> a(64), b(64), d(64);
  for (int l = 0; l < 64; ++l) b[l] = 1 + l;
  (void)hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
  // D[o] = sum over (la, lb) pairs; with A one-hot at la, D[o] = b[lb] -> prints lb+1
  printf("probe: rows = A lane la, columns = output lane o, entry = B lane feeding it (or . if none)\n");
  for (int la = 0; la < 64; ++la) {
    for (int l = 0; l < 64; ++l) a[l] = (l == la);
    (void)hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    printf("la=%2d:", la);
    for (int o = 0; o < 64; ++o)
      if (d[o] != 0) printf(" o%d<-b%d", o, (int)d[o] - 1);
    printf("\n");
  }
  return 0;
}
