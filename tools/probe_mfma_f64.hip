// Hardware probe (not product code): numerics + rate of v_mfma_f64_16x16x4_f64 on gfx950,
// and IEEE-correctness of f64 division / sqrt as compiled by hipcc.  Drives design choices in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma_once(const double* A, const double* B, const double* C, double* D, int nk) {
  // A: 16 x (4*nk) row-major, B: (4*nk) x 16 row-major, C/D: 16x16 row-major
  int l = threadIdx.x;
  int K = 4 * nk;
  d4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
  for (int s = 0; s < nk; ++s) {
    double a = A[(l & 15) * K + 4 * s + (l >> 4)];
    double b = B[(4 * s + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

__global__ void k_rate(double* out, int iters) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}


// the TRMM inner loop without global traffic: 16 accumulators (64 x 64 wave tile), per 16 MFMAs eight ds_read_b64 of fresh
// operands from LDS (conflict-free, lane-contiguous) -- what the matrix cores deliver when fed from LDS
__global__ __launch_bounds__(256, 2) void k_rate_lds(double* out, int iters) {
  __shared__ double buf[8192];                       // 64 KiB
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 8192; e += 256) buf[e] = 1.0 + (e & 1023) * 1e-6;
  __syncthreads();
  d4 c[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const double* p = buf + ((it & 7) * 1024) + (w & 1) * 256;
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = p[i * 64 + l]; b[i] = p[512 + i * 64 + l]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], c[i][j], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// variant: two k-steps per LDS read (ds_read_b128), 8 reads per 32 MFMAs
__global__ __launch_bounds__(256, 2) void k_rate_lds128(double* out, int iters) {
  __shared__ __attribute__((aligned(16))) double buf[8192];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 8192; e += 256) buf[e] = 1.0 + (e & 1023) * 1e-6;
  __syncthreads();
  d4 c[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; it += 2) {
    const double2* p = reinterpret_cast<const double2*>(buf + ((it & 6) * 1024)) + (w & 1) * 256;
    double2 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = p[i * 64 + l]; b[i] = p[512 + i * 64 + l]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, c[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, c[i][j], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// variant: explicit register double buffering (operands of step it+1 requested before the MFMAs of step it)
__global__ __launch_bounds__(256, 2) void k_rate_lds_pipe(double* out, int iters) {
  __shared__ double buf[8192];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 8192; e += 256) buf[e] = 1.0 + (e & 1023) * 1e-6;
  __syncthreads();
  d4 c[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = d4{0, 0, 0, 0};
  double a[4], b[4], an[4], bn[4];
  { const double* p = buf + (w & 1) * 256;
    for (int i = 0; i < 4; ++i) { a[i] = p[i * 64 + l]; b[i] = p[512 + i * 64 + l]; } }
  for (int it = 0; it < iters; ++it) {
    const double* p = buf + (((it + 1) & 7) * 1024) + (w & 1) * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) { an[i] = p[i * 64 + l]; bn[i] = p[512 + i * 64 + l]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], c[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = an[i]; b[i] = bn[i]; }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_vfma_rate(double* out, int iters) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  double c[16];
  for (int j = 0; j < 16; ++j) c[j] = j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = __builtin_fma(c[j], a, b);
  }
  double s = 0;
  for (int j = 0; j < 16; ++j) s += c[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_divsqrt(const double* x, const double* y, double* q, double* r, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { q[i] = x[i] / y[i]; r[i] = sqrt(fabs(x[i])); }
}

static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  // ---------- 1. numerics ----------
  for (int nk : {1, 3}) {
    int K = 4 * nk;
    long mism[5] = {0, 0, 0, 0, 0}; long total = 0;
    for (int trial = 0; trial < 200; ++trial) {
      std::vector<double> A(16 * K), B(K * 16), C(256), D(256);
      for (auto& v : A) v = U(rng) * std::ldexp(1.0, (int)(rng() % 40) - 20);
      for (auto& v : B) v = U(rng) * std::ldexp(1.0, (int)(rng() % 40) - 20);
      for (auto& v : C) v = U(rng) * std::ldexp(1.0, (int)(rng() % 40) - 20);
      double *dA, *dB, *dC, *dD;
      CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dC, 2048)); CK(hipMalloc(&dD, 2048));
      CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, nk);
      CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
      CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dD));
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double c0 = C[i * 16 + j];
        double asc = c0, desc = c0, unf = c0; long double ex = c0;
        for (int k = 0; k < K; ++k) asc = std::fma(A[i * K + k], B[k * 16 + j], asc);
        // descending inside each 4-block, blocks ascending
        for (int s = 0; s < nk; ++s) for (int k = 3; k >= 0; --k) desc = std::fma(A[i * K + 4 * s + k], B[(4 * s + k) * 16 + j], desc);
        for (int k = 0; k < K; ++k) { volatile double p = A[i * K + k] * B[k * 16 + j]; unf = unf + p; }
        // block-exact: each 4-block dot computed in long double then added
        double blk = c0;
        for (int s = 0; s < nk; ++s) { long double t = blk; for (int k = 0; k < 4; ++k) t += (long double)A[i * K + 4 * s + k] * (long double)B[(4 * s + k) * 16 + j]; blk = (double)t; }
        for (int k = 0; k < K; ++k) ex += (long double)A[i * K + k] * (long double)B[k * 16 + j];
        double g = D[i * 16 + j];
        ++total;
        if (bits(g) != bits(asc)) ++mism[0];
        if (bits(g) != bits(desc)) ++mism[1];
        if (bits(g) != bits(unf)) ++mism[2];
        if (bits(g) != bits(blk)) ++mism[3];
        if (bits(g) != bits((double)ex)) ++mism[4];
      }
    }
    printf("numerics nk=%d total=%ld mismatches: fma_asc=%ld fma_desc_in_block=%ld unfused_asc=%ld block_longdouble=%ld full_longdouble=%ld\n",
           nk, total, mism[0], mism[1], mism[2], mism[3], mism[4]);
  }
  // ---------- 2. rate ----------
  {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;          // 250000 = about 55 ms per launch (sustained clocks)
    for (int wpb : {256, 512, 256, 512}) {
      int blocks = prop.multiProcessorCount * (wpb == 256 ? 2 : 1);
      for (int rep = 0; rep < 2; ++rep) {
      double* out; CK(hipMalloc(&out, (size_t)blocks * wpb * 8));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(wpb), 0, 0, out, 100);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(wpb), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)blocks * (wpb / 64) * iters * 4.0 * (2.0 * 16 * 16 * 4);
      printf("mfma_f64 rate: blocks=%d threads=%d  %.3f ms  %.2f TFLOP/s\n", blocks, wpb, ms, flops / ms * 1e-9);
      CK(hipFree(out));
      }
    }
    {  // MFMA fed from LDS (TRMM inner loop shape), 2 workgroups of 4 waves per CU; three feeding patterns
      int blocks = prop.multiProcessorCount * 2, wpb = 256; int it = iters / 4;
      for (int variant = 0; variant < 3; ++variant)
      for (int rep = 0; rep < 2; ++rep) {
        double* out; CK(hipMalloc(&out, (size_t)blocks * wpb * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&](int n) {
          if (variant == 0) hipLaunchKernelGGL(k_rate_lds, dim3(blocks), dim3(wpb), 0, 0, out, n);
          else if (variant == 1) hipLaunchKernelGGL(k_rate_lds128, dim3(blocks), dim3(wpb), 0, 0, out, n);
          else hipLaunchKernelGGL(k_rate_lds_pipe, dim3(blocks), dim3(wpb), 0, 0, out, n);
        };
        launch(100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch(it);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)blocks * (wpb / 64) * it * 16.0 * (2.0 * 16 * 16 * 4);
        printf("mfma_f64 fed from LDS (%s): %.3f ms  %.2f TFLOP/s\n", variant == 0 ? "ds_read_b64 per k-step" : variant == 1 ? "ds_read_b128, two k-steps" : "b64, register double buffer", ms, flops / ms * 1e-9);
        CK(hipFree(out));
      }
    }
    {
      int blocks = prop.multiProcessorCount * 4, wpb = 256; int it = 20000;
      double* out; CK(hipMalloc(&out, (size_t)blocks * wpb * 8));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k_vfma_rate, dim3(blocks), dim3(wpb), 0, 0, out, 100);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_vfma_rate, dim3(blocks), dim3(wpb), 0, 0, out, it);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)blocks * wpb * it * 16.0 * 2.0;
      printf("v_fma_f64 rate: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
    }
  }
  // ---------- 3. div / sqrt ----------
  {
    int n = 1 << 22;
    std::vector<double> x(n), y(n), q(n), r(n);
    for (int i = 0; i < n; ++i) { x[i] = U(rng) * std::ldexp(1.0, (int)(rng() % 200) - 100); y[i] = U(rng) * std::ldexp(1.0, (int)(rng() % 200) - 100); if (y[i] == 0) y[i] = 1; }
    double *dx, *dy, *dq, *dr;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dy, n * 8)); CK(hipMalloc(&dq, n * 8)); CK(hipMalloc(&dr, n * 8));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_divsqrt, dim3(n / 256), dim3(256), 0, 0, dx, dy, dq, dr, n);
    CK(hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost));
    long md = 0, ms_ = 0;
    for (int i = 0; i < n; ++i) { if (bits(q[i]) != bits(x[i] / y[i])) ++md; if (bits(r[i]) != bits(std::sqrt(std::fabs(x[i])))) ++ms_; }
    printf("div mismatches=%ld / %d   sqrt mismatches=%ld / %d\n", md, n, ms_, n);
  }
  return 0;
}
