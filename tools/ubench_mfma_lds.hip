// How to feed v_mfma_f64_4x4x4_4b with a fresh operator fragment every 3 instructions (what the 3-D stage kernels
// do: one A fragment serves the three components).  A small-tile A operand has 16 distinct values replicated over
// the four quads of every 16-lane row.  Variants: (0) fragments stay in registers (ceiling); (1) ds_read_b64 by all
// 64 lanes; (2) ds_read_b128 = two fragments per read; (3) ds_read_b64 by the first quad of every row only, then two
// DPP row_shr steps per dword to fill the other three quads.  27 accumulators, 2 waves per SIMD, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  __shared__ double tab[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 1e-3 * i;
  __syncthreads();
  double acc[27];
  for (int i = 0; i < 27; ++i) acc[i] = 0;
  const int lane = threadIdx.x & 63;
  const int pos = (lane >> 4) * 4 + (lane & 3);
  const bool first_quad = (lane & 12) == 0;
  double b0 = threadIdx.x * 2e-3 + 1.0, b1 = b0 + 1, b2 = b0 + 2;
  double a[9];
  for (int t = 0; t < 9; ++t) a[t] = tab[t * 16 + pos];
  for (int it = 0; it < iters; ++it) {
    const int base = (it & 15) * 256;
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t) a[t] = tab[base + t * 16 + pos];
    } else if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
        const double2 v = *reinterpret_cast<const double2*>(&tab[base + t * 16 + pos * 2]);
        a[t] = v.x;
        a[t + 1] = v.y;
      }
      a[8] = tab[base + 8 * 16 + pos];
    } else if (MODE == 3) {
      if (first_quad) {
#pragma unroll
        for (int t = 0; t < 9; ++t) a[t] = tab[base + t * 16 + pos];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        int lo = __double2loint(a[t]), hi = __double2hiint(a[t]);
        // row_shr:4 into bank 1 (lanes 4-7 of each row), then row_shr:8 into banks 2, 3
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0xf, 0x2, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xf, 0x2, false);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x118, 0xf, 0xc, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x118, 0xf, 0xc, false);
        a[t] = __hiloint2double(hi, lo);
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[3 * t + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t], b0, acc[3 * t + 0], 0, 0, 0);
      acc[3 * t + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t], b1, acc[3 * t + 1], 0, 0, 0);
      acc[3 * t + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t], b2, acc[3 * t + 2], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < 27; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double* out) {
  const int grid = 512, iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("%-62s", name);
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf(" %5.1f", (double)iters * 27 * 512.0 * grid * 4 / ms / 1e9);
  }
  printf("  TFLOP/s\n");
  fflush(stdout);
}

int main() {
  double* out;
  (void)hipMalloc(&out, 512 * 256 * sizeof(double));
  run<0>("fragments in registers", out);
  run<1>("ds_read_b64, all lanes, 1 per 3 MFMA", out);
  run<2>("ds_read_b128 (2 fragments), all lanes", out);
  run<3>("ds_read_b64 by 16 lanes + 4 DPP moves per fragment", out);
  return 0;
}
