// Calibration probe: when do workgroups of one XCD that stream the same addresses share the fetches in that XCD's L2?
// 64 workgroups on XCD 0 (blockIdx % 8 == 0; the other workgroups exit at once) each read the same buffer of S bytes front to back in 16 KiB chunks,
// one chunk every `pace` ticks of the 100 MHz clock (the FP64 product reads one 16 KiB chunk per operand about every 3.4 us); workgroup t starts
// stagger[t] ticks late.  FETCH_SIZE of the dispatch (rocprofv3 --pmc FETCH_SIZE) / S = how many times the buffer crossed the fabric: 1 = every fetch shared.
// hipcc --offload-arch=gfx950 -O3 tools/probe_l2share.hip -o tools/bin/probe_l2share ;  rocprofv3 --pmc FETCH_SIZE ... -- tools/bin/probe_l2share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k_stream(const double* __restrict__ buf, size_t chunks, int pace, const int* __restrict__ stagger, int nwg, double* out) {
  if (blockIdx.x % 8 != 0) return;
  const int t = blockIdx.x / 8;
  if (t >= nwg) return;
  const unsigned long long t0 = wall_clock64() + (unsigned long long)stagger[t];
  double s = 0;
  for (size_t c = 0; c < chunks; ++c) {
    while (wall_clock64() < t0 + (unsigned long long)c * pace) {}
    const double2* p = reinterpret_cast<const double2*>(buf + c * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const double2 v = p[i * 256 + threadIdx.x]; s += v.x + v.y; }
  }
  if (s == 123.456) out[threadIdx.x] = s;
}
// two streams per workgroup, as in the FP64 product: workgroup t = (r, c) reads row stream r and column stream c (8 + 8 streams of S bytes);
// order 0: r = t / 8, c = t % 8 (consecutive workgroups share the row stream), order 1: r = t % 8, c = t / 8
__global__ __launch_bounds__(256) void k_stream2(const double* __restrict__ rows, const double* __restrict__ cols, size_t S8, size_t chunks, int pace, int order, int lds64k, double* out) {
  extern __shared__ double sm[];
  if (blockIdx.x % 8 != 0) return;
  const int t = blockIdx.x / 8;
  const int r = order ? t % 8 : t / 8, c = order ? t / 8 : t % 8;
  const unsigned long long t0 = wall_clock64();
  double s = 0;
  for (size_t k = 0; k < chunks; ++k) {
    while (wall_clock64() < t0 + (unsigned long long)k * pace) {}
    const double2* p = reinterpret_cast<const double2*>(rows + (size_t)r * S8 + k * 2048);
    const double2* q = reinterpret_cast<const double2*>(cols + (size_t)c * S8 + k * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const double2 v = p[i * 256 + threadIdx.x], w = q[i * 256 + threadIdx.x]; s += v.x + v.y + w.x + w.y; }
  }
  if (s == 123.456) out[threadIdx.x] = s;
  if (lds64k && threadIdx.x == 0) sm[0] = s;
}
int main(int argc, char** argv) {
  const size_t S = 64ull << 20, chunks = S / 16384;
  double *buf, *out; hipMalloc(&buf, S); hipMalloc(&out, 4096); hipMemset(buf, 0, S);
  int* dst; hipMalloc(&dst, 64 * 4);
  struct Case { const char* name; int nwg, pace; int stag_mode; int d; };
  // stag_mode 0: all together; 1: workgroup t starts t * d ticks late; 2: workgroups in groups of 8 consecutive, group g starts g * d late; 3: t % 8 * d late
  const Case cases[] = {
    {"64 WGs together, one chunk / 3 us", 64, 300, 0, 0},
    {"64 WGs, t x 0.1 us late", 64, 300, 1, 10},
    {"64 WGs, t x 1 us late", 64, 300, 1, 100},
    {"64 WGs, t x 5 us late", 64, 300, 1, 500},
    {"64 WGs, groups of 8 consecutive, group g x 3 us late", 64, 300, 2, 300},
    {"64 WGs, groups of 8 consecutive, group g x 20 us late", 64, 300, 2, 2000},
    {"64 WGs, (t % 8) x 3 us late", 64, 300, 3, 300},
    {"8 WGs together", 8, 300, 0, 0},
    {"8 WGs, t x 5 us late", 8, 300, 1, 500},
    {"8 WGs, t x 50 us late", 8, 300, 1, 5000},
    {"2 WGs, 200 us apart", 2, 300, 1, 20000},
    {"64 WGs together, as fast as they can", 64, 0, 0, 0},
    {"64 WGs, t x 1 us late, as fast as they can", 64, 0, 1, 100},
  };
  for (const Case& c : cases) {
    std::vector<int> st(64, 0);
    for (int t = 0; t < 64; ++t) st[t] = c.stag_mode == 1 ? t * c.d : c.stag_mode == 2 ? (t / 8) * c.d : c.stag_mode == 3 ? (t % 8) * c.d : 0;
    hipMemcpy(dst, st.data(), 64 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_stream, dim3(512), dim3(256), 0, 0, buf, chunks, c.pace, dst, c.nwg, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("case: %-60s %8.2f ms   (S = %zu MiB per workgroup, %d workgroups)\n", c.name, ms, S >> 20, c.nwg);
  }
  {
    const size_t S2 = 32ull << 20, ch2 = S2 / 16384;
    double *rows, *cols; hipMalloc(&rows, 8 * S2); hipMalloc(&cols, 8 * S2); hipMemset(rows, 0, 8 * S2); hipMemset(cols, 0, 8 * S2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream2), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int paces[] = {340, 340, 100, 0};
    for (int pi = 0; pi < 4; ++pi)
      for (int order = 0; order < 2; ++order) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_stream2, dim3(512), dim3(256), pi == 1 ? 65536 : 0, 0, rows, cols, S2 / 8, ch2, paces[pi], order, pi == 1, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("case: two streams (8 row + 8 column streams of 32 MiB = 512 MiB if every fetch is shared), order %d, pace %d ticks%s %8.2f ms\n", order, paces[pi], pi == 1 ? ", 64 KiB LDS" : "", ms);
      }
  }
  return 0;
}
