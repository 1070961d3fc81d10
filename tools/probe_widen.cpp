// Measurement harness (not product code): the host-side widening of a C3 batch of int32 rows into int64 rows (psfp.hip, widen_rows) -- ordinary stores against
// streaming stores, N threads over 16 MiB chunks.   clang++ -O3 -std=c++17 -pthread tools/probe_widen.cpp -o tools/bin/probe_widen ; tools/bin/probe_widen [threads]
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
#include <chrono>
#include <atomic>
static void widen_rows(int64_t* __restrict__ dst, const int32_t* __restrict__ src, size_t cnt) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef int v2i __attribute__((ext_vector_type(2)));
  typedef long long v2l __attribute__((ext_vector_type(2)));
  size_t i = 0;
  while (i < cnt && (reinterpret_cast<uintptr_t>(dst + i) & 15)) { dst[i] = (int64_t)src[i]; ++i; }
  for (; i + 4 <= cnt; i += 4) {
    v4i x;
    std::memcpy(&x, src + i, sizeof(x));
    const v2i a = __builtin_shufflevector(x, x, 0, 1), b = __builtin_shufflevector(x, x, 2, 3);
    __builtin_nontemporal_store(__builtin_convertvector(a, v2l), reinterpret_cast<v2l*>(dst + i));
    __builtin_nontemporal_store(__builtin_convertvector(b, v2l), reinterpret_cast<v2l*>(dst + i + 2));
  }
  for (; i < cnt; ++i) dst[i] = (int64_t)src[i];
  std::atomic_thread_fence(std::memory_order_seq_cst);
}
static void plain(int64_t* __restrict__ dst, const int32_t* __restrict__ src, size_t cnt) { for (size_t i = 0; i < cnt; ++i) dst[i] = (int64_t)src[i]; }
int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 4;
  const size_t N = (size_t)126160896;   // C3 batch
  std::vector<int32_t> src(N); std::vector<int64_t> dst(N + 1, 1);
  for (size_t i = 0; i < N; ++i) src[i] = (int32_t)(i * 2654435761u);
  for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    const size_t CE = (size_t)4 << 20, nch = (N + CE - 1) / CE;
    for (int w = 0; w < nt; ++w) th.emplace_back([&, w] { for (size_t c = w; c < nch; c += nt) { size_t b0 = c * CE, cnt = N - b0 < CE ? N - b0 : CE; (mode ? widen_rows : plain)(dst.data() + 1 + b0, src.data() + b0, cnt); } });
    for (auto& t : th) t.join();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    size_t bad = 0; for (size_t i = 0; i < N; i += 9973) bad += dst[1 + i] != (int64_t)src[i];
    bad += dst[N] != (int64_t)src[N - 1];
    printf("%s %d threads: %.1f ms  (bad %zu)\n", mode ? "streaming" : "plain    ", nt, ms, bad);
  }
}
