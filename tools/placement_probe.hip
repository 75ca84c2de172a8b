// Where and when do the blocks of a persistent grid (2 blocks of 66 KB LDS per CU) start on a
// CU-masked stream?  Per XCD: blocks placed, distinct CUs used, blocks that started late.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k_persist(unsigned long long* out) {
  __shared__ double lds[66 * 128];  // 66 KB: two blocks per CU
  unsigned long long t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  lds[threadIdx.x] = (double)t0;
  if (threadIdx.x == 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    out[2 * blockIdx.x] = t0;
    out[2 * blockIdx.x + 1] = ((unsigned long long)xcc << 16) | hwid;
  }
  for (int i = 0; i < 1500; ++i) __builtin_amdgcn_s_sleep(127);   // ~5 ms at 100 MHz realtime / a few ms of core clocks
  if (lds[threadIdx.x] < 0) out[0] = 0;
}
static void run(hipStream_t s, int nblk, const char* name) {
  unsigned long long* d;
  (void)hipMalloc(&d, nblk * 16);
  hipLaunchKernelGGL(k_persist, dim3(nblk), dim3(256), 0, s, d);
  (void)hipStreamSynchronize(s);
  std::vector<unsigned long long> h(2 * nblk);
  (void)hipMemcpy(h.data(), d, nblk * 16, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (int b = 0; b < nblk; ++b) tmin = h[2 * b] < tmin ? h[2 * b] : tmin;
  printf("%s, %d blocks:\n", name, nblk);
  for (int x = 0; x < 8; ++x) {
    std::map<unsigned, int> cus;
    int nb = 0, late = 0, lab_mismatch = 0;
    for (int b = 0; b < nblk; ++b) {
      const unsigned v = (unsigned)h[2 * b + 1];
      if (((v >> 16) & 15) != (unsigned)x) continue;
      ++nb;
      cus[(v >> 8) & 0xff]++;
      if (h[2 * b] - tmin > 10000) ++late;   // > 100 us after the first block (100 MHz counter)
      if (b % 8 != x) ++lab_mismatch;
    }
    int mx = 0;
    for (auto& kv : cus) mx = kv.second > mx ? kv.second : mx;
    int labels[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nblk; ++b)
      if ((((unsigned)h[2 * b + 1] >> 16) & 15) == (unsigned)x) labels[b % 8] += 1;
    printf("  XCD %d holds blockIdx %% 8 =", x);
    for (int l = 0; l < 8; ++l)
      if (labels[l]) printf(" %d (x%d)", l, labels[l]);
    printf("\n");
    printf("  XCD %d: %3d blocks on %2zu CUs (max %d per CU), %d started late, %d with blockIdx %% 8 != XCD\n", x, nb,
           cus.size(), mx, late, lab_mismatch);
  }
  (void)hipFree(d);
}
int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  hipStream_t plain, masked;
  (void)hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
  std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
  for (int b = 0; b < ncu; ++b) mask[b / 32] |= 1u << (b % 32);
  for (int x = 0; x < 8; ++x) {
    const int b = (ncu / 8) * x + (31 - x);
    mask[b / 32] &= ~(1u << (b % 32));
  }
  (void)hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data());
  run(plain, 512, "no mask");
  run(masked, 496, "one CU per XCD masked off");
  run(masked, 448, "one CU per XCD masked off");
  return 0;
}
