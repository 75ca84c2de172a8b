// Why do 4-row FP64 MFMAs run slower inside the stage kernels than in tools/ubench_mfma_lds.hip?  The volume phase of
// G rebuilt step by step: 27 v_mfma_f64_4x4x4_4b per "k-step" (9 operator fragments x 3 components), 2 waves per SIMD.
//   mode 0: fragments batched from LDS at the top of the k-step (ubench_mfma_lds mode 1)
//   mode 1: + the three B operands of every k-step come from global memory, requested 4 k-steps ahead; WINDOW k-steps
//           of data per wave are re-read over and over (8: L2-resident like the cells' own rows that G reads three
//           times; 40000: a pure HBM stream, 9 flop per byte)
//   mode 2: + after every 9 k-steps the 27 accumulators are folded into 54 running sums (VALU) and cleared
//   mode 3: mode 2 with the fragments streamed through a ring of 4 registers, 3 fragments ahead (the rebuilt kernels)
//   mode 4: mode 2 with v_mfma_f64_16x16x4 on the same data volume (2 large + 1 small tile per k-step and component:
//           what the shipped kernels do)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int WINDOW>
__global__ __launch_bounds__(256, 2) void k(const double* __restrict__ src, double* out, int ksteps, long stride) {
  __shared__ double tab[9 * 16 * 16];
  for (int i = threadIdx.x; i < 9 * 16 * 16; i += 256) tab[i] = 1e-3 * (i % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int pos = (lane >> 4) * 4 + (lane & 3);
  const double* p = src + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * stride + lane;
  double acc[27], sum[54];
  d4 big[6];
  for (int i = 0; i < 27; ++i) acc[i] = 0;
  for (int i = 0; i < 54; ++i) sum[i] = 0;
  for (int i = 0; i < 6; ++i) big[i] = d4{0, 0, 0, 0};
  double bq[4][3];
  for (int s = 0; s < 4; ++s)
    for (int c = 0; c < 3; ++c) bq[s][c] = (MODE >= 1) ? p[(s * 3 + c) * 64] : 1.0 + c;
  double ar[4];
  for (int j = 0; j < 3; ++j) ar[j] = tab[j * 16 + pos];
  for (int ks = 0; ks < ksteps; ks += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = ks + u;
      double b[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) b[c] = bq[u][c];
      if (MODE >= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) bq[u][c] = p[((long)((kk + 4) % WINDOW) * 3 + c) * 64];
      }
      const int base = (kk & 15) * 144;
      if (MODE == 4) {
        const double a0 = tab[base + lane], a1 = tab[base + 64 + lane], a2 = tab[base + 128 + pos];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          big[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[c], big[c], 0, 0, 0);
          big[3 + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[c], big[3 + c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b[c], acc[c], 0, 0, 0);
        }
      } else if (MODE == 3) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int j = u * 9 + t;   // 36 fragments per unrolled body: ring index is compile-time
          ar[(j + 3) % 4] = tab[((kk * 9 + t + 3) % 144) * 16 + pos];
          const double a = ar[j % 4];
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[3 * t + c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[c], acc[3 * t + c], 0, 0, 0);
        }
      } else {
        double a[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) a[t] = tab[base + t * 16 + pos];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[3 * t + c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t], b[c], acc[3 * t + c], 0, 0, 0);
      }
    }
    if (MODE >= 2 && (ks % 8) == 4) {   // every 8 k-steps (the kernels: 9)
      if (MODE == 4) {
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sum[c * 4 + r] += 0.3 * big[c][r];
            sum[24 + c * 4 + r] += 0.7 * big[c][r];
            big[c][r] = 0;
          }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          sum[48 + c] += 0.3 * acc[c];
          acc[c] = 0;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 27; ++i) {
          sum[i] += 0.3 * acc[i];
          sum[27 + i] += 0.7 * acc[i];
          acc[i] = 0;
        }
      }
#pragma unroll
      for (int i = 0; i < 54; ++i) asm volatile("" : "+v"(sum[i]));
    }
  }
  double s = 0;
  for (int i = 0; i < 27; ++i) s += acc[i];
  for (int i = 0; i < 54; ++i) s += sum[i];
  for (int i = 0; i < 6; ++i) s += big[i][0] + big[i][3];
  for (int u = 0; u < 4; ++u) s += bq[u][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + ar[0];
}

template <int MODE, int WINDOW>
void run(const char* name, const double* src, double* out) {
  const int grid = 512, ksteps = 40000;
  const long stride = (long)(ksteps + 8) * 3 * 64;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("%-78s", name);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WINDOW>), dim3(grid), dim3(256), 0, 0, src, out, ksteps, stride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (MODE == 4) ? (6 * 2048.0 + 3 * 512.0) : 27 * 512.0;
    printf(" %5.1f", (double)ksteps * flop * grid * 4 / ms / 1e9);
  }
  printf("  TFLOP/s (issued)\n");
  fflush(stdout);
}

int main() {
  const long n = 2048L * (40008L * 3 * 64);
  double *src, *out;
  (void)hipMalloc(&src, n * sizeof(double));
  (void)hipMemset(src, 0, n * sizeof(double));
  (void)hipMalloc(&out, 512 * 256 * sizeof(double));
  run<0, 8>("0: 27 small MFMAs per k-step, fragments batched from LDS", src, out);
  run<1, 8>("1: + B operands from global memory (L2-resident window), 4 k-steps ahead", src, out);
  run<2, 8>("2: + fold into 54 running sums every 8 k-steps", src, out);
  run<3, 8>("3: as 2, fragments through a ring of 4 registers", src, out);
  run<4, 8>("4: as 2 with 16-row tiles (2 large + 1 small per k-step and component)", src, out);
  run<2, 40000>("2': as 2, B operands a pure HBM stream", src, out);
  run<4, 40000>("4': as 4, B operands a pure HBM stream", src, out);
  return 0;
}
