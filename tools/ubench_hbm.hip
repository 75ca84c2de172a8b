// Practical HBM3E stream rates on one MI355X for the read:write mixes of the LF4 stages
// (all arrays far larger than the 256 MB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NR, int NW>
__global__ __launch_bounds__(256) void k_mix(const d2* __restrict__ in, d2* __restrict__ out, long n) {
  // n = elements per stream; NR read streams, NW write streams
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    d2 s = {0, 0};
#pragma unroll
    for (int r = 0; r < NR; ++r) s += in[r * n + i];
#pragma unroll
    for (int w = 0; w < NW; ++w) out[w * n + i] = s * (double)(w + 1);
    if (NW == 0 && s[0] == 1.2345e300) out[i] = s;
  }
}
template <int NR, int NW>
void run(const d2* in, d2* out, long n, int grid) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mix<NR, NW>), dim3(grid), dim3(256), 0, 0, in, out, n);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_mix<NR, NW>), dim3(grid), dim3(256), 0, 0, in, out, n);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("reads=%2d writes=%2d grid=%5d : %.3f ms, %.0f GB/s\n", NR, NW, grid, ms, (NR + NW) * n * 16.0 / ms / 1e6);
}
int main() {
  const long n = 24L << 20;  // 384 MiB per stream
  d2 *in, *out;
  (void)hipMalloc(&in, 15 * n * 16); (void)hipMalloc(&out, 6 * n * 16);
  (void)hipMemset(in, 0, 15 * n * 16);
  for (int grid : {2048, 8192, 65536}) {
    run<1, 0>(in, out, n, grid);
    run<6, 0>(in, out, n, grid);
    run<0, 1>(in, out, n, grid);
    run<1, 1>(in, out, n, grid);
    run<3, 6>(in, out, n, grid);   // G0
    run<6, 3>(in, out, n, grid);   // F0
    run<12, 3>(in, out, n, grid);  // U1
    run<15, 6>(in, out, n, grid);  // S1
  }
  return 0;
}
