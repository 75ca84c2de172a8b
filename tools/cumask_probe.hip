// Which compute units does a CU-masked stream use?  Prints, for the mask sg_create builds for
// reserve_cus = 1 (api.cpp) and for no mask, the number of distinct CUs seen per XCD by a grid
// that oversubscribes the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out) {
  if (threadIdx.x == 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID[15:0]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // XCC_ID[3:0]
    out[blockIdx.x] = (xcc << 16) | hwid;
  }
  // stay a little so blocks spread over all CUs
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);
}
static void run(hipStream_t s, const char* name) {
  const int nb = 4096;
  unsigned* d;
  (void)hipMalloc(&d, nb * 4);
  hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, s, d);
  (void)hipStreamSynchronize(s);
  std::vector<unsigned> h(nb);
  (void)hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
  std::set<unsigned> per[16];
  for (unsigned v : h) {
    const unsigned xcc = (v >> 16) & 15, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7;
    per[xcc].insert((se << 5) | (sh << 4) | cu);
  }
  printf("%s: distinct CUs per XCD:", name);
  int tot = 0;
  for (int x = 0; x < 8; ++x) { printf(" %zu", per[x].size()); tot += (int)per[x].size(); }
  printf("  total %d\n", tot);
  (void)hipFree(d);
}
int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  hipStream_t plain, masked, contiguous;
  (void)hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
  std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
  for (int b = 0; b < ncu; ++b) mask[b / 32] |= 1u << (b % 32);
  std::vector<uint32_t> m2 = mask;
  for (int x = 0; x < 8; ++x) {
    const int b = (ncu / 8) * x + (31 - x);
    mask[b / 32] &= ~(1u << (b % 32));
  }
  for (int b = ncu - 8; b < ncu; ++b) m2[b / 32] &= ~(1u << (b % 32));
  (void)hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data());
  (void)hipExtStreamCreateWithCUMask(&contiguous, (uint32_t)m2.size(), m2.data());
  printf("device CUs: %d\n", ncu);
  run(plain, "no mask");
  run(masked, "one bit per 32-block, staggered (sg_create)");
  run(contiguous, "last 8 bits");
  return 0;
}
