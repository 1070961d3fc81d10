// Calibration probe: cost of one Philox4x32-10 block and of a few scalar/vector idioms on gfx950, and the ratio of s_memtime
// (clock64) to the 100 MHz wall clock while a latency-bound single-wave-per-SIMD kernel runs.
// hipcc --offload-arch=gfx950 -O3 -I tools_amd/csrc tools/probe_philox.hip -o tools/bin/probe_philox && tools/bin/probe_philox
#include <hip/hip_runtime.h>
#include <cstdio>
#include "psf_rng.hpp"
using namespace psf;

__global__ void k_probe(int mode, int iters, unsigned long long* out, unsigned* sink) {
  __shared__ unsigned lds[1024];
  const unsigned long long w0 = wall_clock64();
  const long long c0 = clock64();
  unsigned acc = threadIdx.x;
  double d = 1.0 + threadIdx.x;
  float f = 1.0f + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) { const U4 w = philox(12345, acc, i, 7, 9); acc ^= w.x ^ w.w; }
    else if (mode == 1) { d = fma(d, 1.0000001, 0.5); }                       // dependent f64 fma chain
    else if (mode == 2) { f = __builtin_amdgcn_exp2f(-f * 0.001f) + 1.0f; }   // dependent exp chain
    else if (mode == 3) { d = ceil(d * 1.0000001) + 0.25; }                   // ceil + mul + add (f64)
    else if (mode == 4) { const unsigned long long m = __ballot(acc & 1); acc += (unsigned)__builtin_ctzll(m | 1) + (unsigned)__builtin_amdgcn_readlane((int)acc, (int)(i & 63)); }
    else if (mode == 5) { acc = (unsigned)(((unsigned long long)acc * 0xD2511F53u) >> 32) ^ i; }   // dependent v_mad_u64_u32 chain
    else if (mode == 6) { d = fma(d, 1.0000001, 0.5); __builtin_amdgcn_s_sleep(4); }                // mostly asleep: does the shader clock follow?
    else if (mode == 7) { lds[threadIdx.x] = acc; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); acc = lds[(threadIdx.x + 1) & 255] + i; }   // LDS round trip
  }
  const long long c1 = clock64();
  const unsigned long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned long long)(c1 - c0); out[1] = w1 - w0; }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + (unsigned)d + (unsigned)f;
}

int main() {
  unsigned long long* out; unsigned* sink;
  hipMalloc(&out, 16); hipMalloc(&sink, 1024 * 256 * 4);
  const char* names[] = {"philox block", "f64 fma (dependent)", "exp2f + add (dependent)", "ceil+mul+add f64", "ballot+ctz+readlane", "v_mad_u64_u32 (dependent)", "f64 fma + s_sleep 4", "LDS write + read (dependent)"};
  for (int waves = 1; waves <= 4; waves *= 4)
    for (int mode = 0; mode < 8; ++mode) {
      const int iters = 20000;
      hipLaunchKernelGGL(k_probe, dim3(256), dim3(256 * waves), 0, 0, mode, iters, out, sink);   // one (or four) wave(s) per SIMD
      hipDeviceSynchronize();
      unsigned long long h[2];
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("%d wave/SIMD  %-28s %8.1f clock64 ticks / iteration   %7.2f ns / iteration   clock64 rate %.2f GHz\n", waves, names[mode], (double)h[0] / iters,
             (double)h[1] * 10.0 / iters, (double)h[0] / ((double)h[1] * 10.0));
    }
  return 0;
}
