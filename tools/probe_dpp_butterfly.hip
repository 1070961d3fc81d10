#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ inline double xor_sum_ref(double acc) { for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off); return acc; }
__device__ inline double mk(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ inline double wave_xor_sum(double x) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  uint32_t lo = (uint32_t)__double2loint(x), hi = (uint32_t)__double2hiint(x);
  { u2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); u2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = mk(a.x, b.x) + mk(a.y, b.y); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); u2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = mk(a.x, b.x) + mk(a.y, b.y); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { uint32_t pl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xf, 0xf, false), ph = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, false);
    x = x + mk(pl, ph); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { int pl = __builtin_amdgcn_update_dpp(0, (int)lo, 0x104, 0xf, 0x5, false); pl = __builtin_amdgcn_update_dpp(pl, (int)lo, 0x114, 0xf, 0xa, false);
    int ph = __builtin_amdgcn_update_dpp(0, (int)hi, 0x104, 0xf, 0x5, false); ph = __builtin_amdgcn_update_dpp(ph, (int)hi, 0x114, 0xf, 0xa, false);
    x = x + mk((uint32_t)pl, (uint32_t)ph); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { uint32_t pl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4e, 0xf, 0xf, false), ph = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4e, 0xf, 0xf, false);
    x = x + mk(pl, ph); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { uint32_t pl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xb1, 0xf, 0xf, false), ph = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xb1, 0xf, 0xf, false);
    x = x + mk(pl, ph); }
  return x;
}
__global__ void k(const double* in, double* o1, double* o2) { const double v = in[threadIdx.x + blockIdx.x * blockDim.x]; o1[threadIdx.x + blockIdx.x * blockDim.x] = xor_sum_ref(v); o2[threadIdx.x + blockIdx.x * blockDim.x] = wave_xor_sum(v); }
int main() {
  const int N = 256 * 64; double* h = new double[N]; unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < N; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = ((double)(long long)st) * 1e-9 * ((i % 7) ? 1.0 : 1e-12); }
  double *d, *a, *b; hipMalloc(&d, N * 8); hipMalloc(&a, N * 8); hipMalloc(&b, N * 8); hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, 0, d, a, b); double* ha = new double[N]; double* hb = new double[N];
  hipMemcpy(ha, a, N * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, b, N * 8, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < N; ++i) if (memcmp(&ha[i], &hb[i], 8)) { if (bad < 5) printf("mismatch %d: %.17g vs %.17g\n", i, ha[i], hb[i]); ++bad; }
  printf("DPP_BUTTERFLY %s (%d mismatches of %d)\n", bad ? "FAIL" : "OK", bad, N); return bad != 0;
}
