// Probe: which XCDs / CUs does a CU-masked HIP stream run on?  (hipExtStreamCreateWithCUMask; mask bit -> CU mapping on MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
__global__ void where_kernel(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  // hold the CU for a while so that the grid spreads over every CU the stream may use
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
}
int main(int argc, char** argv) {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs %d\n", ncu);
  const int nblk = 2048;
  unsigned* d; hipMalloc(&d, nblk * 8);
  std::vector<unsigned> h(nblk * 2);
  const char* names[] = {"none", "low128", "high128", "even", "low32", "bits 0-7", "mod8<4", "mod16<8"};
  for (int pat = 0; pat < 8; ++pat) {
    uint32_t mask[8]; memset(mask, 0, sizeof(mask));
    for (int b = 0; b < 256; ++b) {
      bool on = pat == 0 ? true : pat == 1 ? b < 128 : pat == 2 ? b >= 128 : pat == 3 ? (b % 2 == 0) : pat == 4 ? b < 32 : pat == 5 ? b < 8 : pat == 6 ? (b % 8 < 4) : (b % 16 < 8);
      if (on) mask[b / 32] |= 1u << (b % 32);
    }
    hipStream_t s;
    hipError_t e = pat == 0 ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("%s: create failed %s\n", names[pat], hipGetErrorString(e)); continue; }
    hipMemsetAsync(d, 0xff, nblk * 8, s);
    hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(1024), 0, s, d, 200000);
    e = hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost);
    int xcc_hist[16] = {0}; std::vector<int> cu_seen(16 * 4096, 0); int distinct = 0;
    for (int i = 0; i < nblk; ++i) {
      unsigned x = h[2 * i] & 0xf, id = h[2 * i + 1];
      xcc_hist[x]++;
      unsigned key = x * 4096 + ((id >> 8) & 0xf) + 16 * ((id >> 12) & 0x1) + 32 * ((id >> 13) & 0x7);
      if (!cu_seen[key]) { cu_seen[key] = 1; distinct++; }
    }
    printf("%-8s sync=%s distinct CUs %3d | per-XCC workgroups:", names[pat], hipGetErrorString(e), distinct);
    for (int x = 0; x < 8; ++x) printf(" %4d", xcc_hist[x]);
    printf(" | first blocks' xcc:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
