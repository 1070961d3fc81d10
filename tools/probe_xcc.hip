// Calibration probe: which XCD (XCC_ID), shader engine and CU each workgroup of a 1-D grid lands on, as a function of blockIdx.x, for the
// workgroup shape of the FP64 product (256 threads, 64 KiB LDS -> 2 per CU) while the chip is full.
// hipcc --offload-arch=gfx950 -O3 tools/probe_xcc.hip -o tools/bin/probe_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_where(unsigned* out, int spin) {
  extern __shared__ double sm[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = xcc; out[blockIdx.x * 4 + 1] = hwid; out[blockIdx.x * 4 + 2] = (unsigned)t0; sm[0] = 1.0; }
}
int main() {
  const int n = 2048;
  unsigned* d; hipMalloc(&d, n * 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_where), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 65536, 0, d, 20000);     // 200 us each: the first 512 fill the chip, the rest replace them
  hipDeviceSynchronize();
  std::vector<unsigned> h(n * 4); hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  printf("blockIdx: xcc (XCC_ID & 15)  se  cu   [HW_ID: cu_id = bits 11:8, sh_id = 12, se_id = 15:13]\n");
  for (int i = 0; i < n; ++i)
    if (i < 80 || (i >= 512 && i < 560) || i % 256 == 0)
      printf("%5d: xcc %2u  se %u  cu %2u  start %u\n", i, h[i * 4] & 15u, (h[i * 4 + 1] >> 13) & 7u, (h[i * 4 + 1] >> 8) & 15u, h[i * 4 + 2]);
  int agree = 0; for (int i = 0; i < n; ++i) agree += (h[i * 4] & 15u) == (unsigned)(i & 7);
  printf("xcc == blockIdx %% 8 for %d of %d workgroups\n", agree, n);
  int hist[16] = {0}; for (int i = 0; i < 512; ++i) hist[h[i * 4] & 15u]++;
  printf("first 512 workgroups per xcc:"); for (int x = 0; x < 8; ++x) printf(" %d", hist[x]); printf("\n");
  return 0;
}
