// occupy.hip -- test helper (not product code): a kernel that HOLDS compute-unit slots for a given time on a stream of its own, so that the tests can call the
// library on a GPU that is not idle (tests/test_gpu_walk_contention.py).  Built by __graft_entry__.build() into tests/cpp/liboccupy.so.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void k_occupy(uint64_t ticks, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  const uint64_t t0 = wall_clock64();                      // constant 100 MHz counter
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); acc += smem[threadIdx.x & 63]; }
  if (acc == 0xffffffffu) *sink = acc;
}

extern "C" int occupy_start(int device, int ms, int blocks, int threads, int lds_bytes, void** stream_out) {
  if (hipSetDevice(device) != hipSuccess) return 1;
  hipStream_t st;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 2;
  if (lds_bytes > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 3;
  static unsigned* sink = nullptr;
  if (!sink && hipMalloc(&sink, sizeof(unsigned)) != hipSuccess) return 4;
  hipLaunchKernelGGL(k_occupy, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, st, (uint64_t)ms * 100000ull, sink);
  if (hipGetLastError() != hipSuccess) return 5;
  *stream_out = st;
  return 0;
}
extern "C" int occupy_done(void* stream) { return hipStreamQuery((hipStream_t)stream) == hipSuccess ? 1 : 0; }
extern "C" int occupy_wait(void* stream) {
  const int rc = hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : 1;
  (void)hipStreamDestroy((hipStream_t)stream);
  return rc;
}
