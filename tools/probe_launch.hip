// Calibration probe: cost of back-to-back dependent launches on one stream as a function of the workgroup shape (threads, LDS) and of
// what the kernel touches.  hipcc --offload-arch=gfx950 -O3 tools/probe_launch.hip -o tools/bin/probe_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_empty(double* buf, int touch) {
  extern __shared__ double sm[];
  if (touch) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (int r = 0; r < touch; ++r) buf[i + (size_t)r * 262144] += 1.0; }
  if (threadIdx.x == 0) sm[0] = 1.0;
}
int main() {
  double* buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20);
  const int shapes[][3] = {{256, 256, 0}, {256, 512, 0}, {256, 512, 65536}, {512, 512, 65536}, {256, 512, 65536}, {256, 512, 65536}};
  const int touch[] = {0, 0, 0, 0, 1, 16};
  for (int s = 0; s < 6; ++s) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(shapes[s][0]), dim3(shapes[s][1]), shapes[s][2], 0, buf, touch[s]);
      hipDeviceSynchronize();
      auto t1 = std::chrono::steady_clock::now();
      if (rep) printf("%3d WGs x %3d threads, %5d B LDS, touch %2d x 2 MiB : %6.2f us per launch\n", shapes[s][0], shapes[s][1], shapes[s][2], touch[s],
                      std::chrono::duration<double, std::micro>(t1 - t0).count() / 200);
    }
  }
  return 0;
}
