// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for this code's access pattern on gfx950:
// 8 bytes per lane, a wave touching four 128-byte lines [(node, comp)][16 cells] - the pattern
// of the MFMA stage kernels' operand loads and result stores.  Known byte counts:
//   read_kernel  reads  N doubles once, write_kernel writes N doubles once.
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_kernel(const double* __restrict__ p, long nlines, double* out) {
  // one wave reads lines 4i .. 4i+3 (quad q -> line 4i+q, lane w -> word w)
  const int lane = threadIdx.x & 63, q = lane >> 4, w = lane & 15;
  long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  double s = 0;
  for (long i = wave; i * 4 + 3 < nlines; i += nwaves) s += p[(i * 4 + q) * 16 + w];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void write_kernel(double* __restrict__ p, long nlines) {
  const int lane = threadIdx.x & 63, q = lane >> 4, w = lane & 15;
  long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long i = wave; i * 4 + 3 < nlines; i += nwaves) p[(i * 4 + q) * 16 + w] = (double)i;
}
int main() {
  const long nbytes = 4L << 30;  // 4 GiB, far beyond the 256 MiB Infinity Cache
  const long nlines = nbytes / 128;
  double *p, *out;
  if (hipMalloc(&p, nbytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  (void)hipMemset(p, 0, nbytes);
  hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, p, nlines);
  hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, p, nlines, out);
  (void)hipDeviceSynchronize();
  printf("each kernel moved %ld bytes = %.1f KB\n", nbytes, nbytes / 1024.0);
  return 0;
}
