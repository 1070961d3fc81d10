// Where does k_gemm_f64 (psf_gemm_kernels.hpp) spend a K chunk?  Times the product of the left-looking Cholesky's update shape (M rows x 128 columns, K long,
// leading dimension of a C3 key) as shipped and with parts of the loop removed:
//   -DGM_PROBE=1  no global loads after the first chunk (operands stay in LDS: matrix pipe + LDS reads + barrier only)
//   -DGM_PROBE=2  no MFMAs (loads, LDS traffic and barriers only)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I tools_amd/csrc [-DGM_PROBE=n] tools/probe_gemm.hip -o tools/bin/probe_gemm[n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "psf_gemm_kernels.hpp"
using namespace psf;
__global__ void k_fillr(double* p, size_t n) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = (g + 1) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[g] = ((double)(x >> 11) * 0x1.0p-53 - 0.5);
  }
}
int main(int argc, char** argv) {
  const size_t ld = argc > 1 ? atol(argv[1]) : 30801;
  gemm_prepare();
  double *A, *C; hipMalloc(&A, ld * ld * 8); hipMalloc(&C, ld * ld * 8);
  hipLaunchKernelGGL(k_fillr, dim3(4096), dim3(256), 0, 0, A, ld * ld);
  GemmWorkspace w; w.bytes = (size_t)900 * 128 * 128 * 8; hipMalloc(&w.ws, w.bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Shape { size_t M, N, K; const char* what; };
  const Shape shapes[] = {{15360, 128, 15360, "update at the middle of C3 (120 row tiles, 2 K splits)"}, {28160, 128, 2560, "early panel (220 tiles, K = 2560)"},
                          {2560, 128, 28160, "late panel (20 tiles, 12 K splits)"}, {128, 128, 15360, "diagonal tile (1 tile, 64 K splits)"},
                          {15360, 15360, 128, "square, K = 128 (the right-looking update's shape)"}, {8192, 8192, 8192, "square 8192^3"}};
  for (const Shape& sh : shapes) {
    if (sh.M + 128 > ld || sh.K > ld || sh.N > ld) continue;
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(e0, 0);
      // A operand: rows 128.., B operand: rows 0..N (as in the factorisation: both are row blocks of the same matrix)
      launch_gemm<true>(nullptr, GemmArgs{A + 128 * ld, ld, A, ld, C, sh.N > 128 ? sh.N : 128, sh.M, sh.N, sh.K, -1.0, 0.0, nullptr, nullptr, 0}, w);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-62s M %6zu N %6zu K %6zu: %8.3f ms  %6.2f TFLOP/s  %s\n", sh.what, sh.M, sh.N, sh.K, best, 2.0 * sh.M * sh.N * sh.K / best * 1e-9, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
