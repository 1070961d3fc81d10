// Measurement harness (not product code) for the small-batch streaming product x = sqrt(Sigma_2) d at the C3 shape with random operands:
// what bounds k_trmm_stream -- ring depth, the B-operand loads, the MFMAs, the load width / layout, the tail of the longest chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe_stream.hip -o tools/bin/probe_stream
//   tools/bin/probe_stream [m=30801] [reps=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "../tools_amd/csrc/psf_rng.hpp"
#include "../tools_amd/csrc/psf_kernels.hpp"
#include "../tools_amd/csrc/psf_stream_kernels.hpp"
using namespace psf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void k_fill(double* p, size_t n, uint64_t salt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = (i + salt) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = (double)(int64_t)(x >> 11) * 0x1.0p-52 - 1.0;
  }
}

typedef double d2 __attribute__((ext_vector_type(2)));
template <int N> __device__ inline void pw_wait2(d2& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }
__device__ inline void pw_touch2(d2& x) { asm volatile("" : "+v"(x)); }

// LAYOUT 0: the key's chunk stream, one global_load_dwordx2 per k-step and operand (the shipped form).  MODE 0 full, 1 no B loads, 2 loads only (no MFMA).
// W8: 1 = eight-wave workgroups of (long, short) pairs, 0 = four-wave workgroups in descending order.
template <int NB, int PD, int MODE, int W8>
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 2 : 1) void k_probe0(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X, StreamGeom g,
                                                                      size_t nkb, size_t ldx, unsigned long long* __restrict__ tlog) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int task;
  if (W8) {
    const int slot = 4 * (int)blockIdx.x + (wave & 3), mirror = g.ntask - 1 - slot;
    task = wave < 4 ? slot : mirror;
    if (wave < 4 ? slot > mirror : mirror <= slot) return;
  } else {
    task = 4 * (int)blockIdx.x + wave;
    if (task >= g.ntask) return;
  }
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  const int tg = g.ntile - 1 - task / g.ncg, cg = task % g.ncg;
  const int t0 = tg, nsteps = 4 * (t0 + 1);
  const double* gA = Lt + tr_rowblock_base((size_t)(t0 >> 3)) * TR_CHUNK + (size_t)(t0 & 7) * 64;
  const int cf0 = cg * NB;
  const double* gB = Dt + (size_t)(cf0 >> 3) * nkb * TR_CHUNK + (size_t)(cf0 & 7) * 64;
  const uint32_t voff = (uint32_t)lane * 8u;
  d4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  double a[PD], b[PD][NB];
  auto issue = [&](double& av, double (&bv)[NB], int s) {
    const double* pa = gA + (size_t)s * 512;
    const double* pb = gB + (size_t)s * 512;
    ts_load<0>(av, voff, pa);
    if (MODE != 1) ts_for<0, NB>([&](auto J) { ts_load<decltype(J)::value * 512>(bv[decltype(J)::value], voff, pb); });
  };
  constexpr int PER = MODE == 1 ? 1 : 1 + NB;
  static_assert((PD - 1) * PER <= 63, "vmcnt");
  if (MODE == 1) {
#pragma unroll
    for (int u = 0; u < PD; ++u)
#pragma unroll
      for (int j = 0; j < NB; ++j) b[u][j] = 0.001 * (lane + j + u);
  }
#pragma unroll
  for (int u = 0; u < PD; ++u) issue(a[u], b[u], u < nsteps ? u : nsteps - 1);
  for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (s0 + u < nsteps) {
        ts_wait<(PD - 1) * PER>(a[u]);
        if (MODE != 1) {
#pragma unroll
          for (int j = 0; j < NB; ++j) ts_touch(b[u][j]);
        }
        if (MODE != 2) {
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][j], acc[j], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j][0] += a[u] + b[u][j];
        }
      }
      int sn = s0 + u + PD;
      sn = sn < nsteps ? sn : nsteps - 1;
      issue(a[u], b[u], sn);
    }
  }
  ts_wait<0>(a[0]);
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    ts_touch(a[u]);
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < NB; ++j) ts_touch(b[u][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) X[((size_t)t0 * 16 + (lane >> 4) + 4 * r) * ldx + (size_t)(cf0 + j) * 16 + (lane & 15)] = acc[j][r];
  if (tlog && lane == 0) { tlog[2 * (size_t)task] = t_begin; tlog[2 * (size_t)task + 1] = __builtin_amdgcn_s_memrealtime(); }
}

// LAYOUT 1: a per-tile stream [tile][k-step pair][lane][2]: one global_load_dwordx4 brings two k-steps of a tile (1 KiB per wave-instruction); the B operand
// likewise [column fragment][pair][lane][2].  Tile t starts at 128 t (t + 1) doubles.
template <int NB, int PD, int MODE>
__global__ __launch_bounds__(512, 2) void k_probe1(const double* __restrict__ Ls, const double* __restrict__ Ds, double* __restrict__ X, StreamGeom g, size_t dstride,
                                                   size_t ldx, unsigned long long* __restrict__ tlog) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = 4 * (int)blockIdx.x + (wave & 3), mirror = g.ntask - 1 - slot;
  const int task = wave < 4 ? slot : mirror;
  if (wave < 4 ? slot > mirror : mirror <= slot) return;
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  const int tg = g.ntile - 1 - task / g.ncg, cg = task % g.ncg;
  const int t0 = tg, npairs = 2 * (t0 + 1);
  const double* gA = Ls + (size_t)128 * t0 * (t0 + 1);
  const int cf0 = cg * NB;
  const uint32_t voff = (uint32_t)lane * 16u;
  d4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  d2 a[PD], b[PD][NB];
  auto issue = [&](d2& av, d2 (&bv)[NB], int p) {
    const double* pa = gA + (size_t)p * 128;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(av) : "v"(voff), "s"(pa) : "memory");
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const double* pb = Ds + (size_t)(cf0 + j) * dstride + (size_t)p * 128;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bv[j]) : "v"(voff), "s"(pb) : "memory");
      }
    }
  };
  constexpr int PER = MODE == 1 ? 1 : 1 + NB;
  static_assert((PD - 1) * PER <= 63, "vmcnt");
  if (MODE == 1) {
#pragma unroll
    for (int u = 0; u < PD; ++u)
#pragma unroll
      for (int j = 0; j < NB; ++j) b[u][j] = d2{0.001 * (lane + j + u), 0.002 * (lane + u)};
  }
#pragma unroll
  for (int u = 0; u < PD; ++u) issue(a[u], b[u], u < npairs ? u : npairs - 1);
  for (int s0 = 0; s0 < npairs; s0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (s0 + u < npairs) {
        pw_wait2<(PD - 1) * PER>(a[u]);
        if (MODE != 1) {
#pragma unroll
          for (int j = 0; j < NB; ++j) pw_touch2(b[u][j]);
        }
        if (MODE != 2) {
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, b[u][j].x, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].y, b[u][j].y, acc[j], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j][0] += a[u].x + a[u].y + b[u][j].x + b[u][j].y;
        }
      }
      int sn = s0 + u + PD;
      sn = sn < npairs ? sn : npairs - 1;
      issue(a[u], b[u], sn);
    }
  }
  pw_wait2<0>(a[0]);
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    pw_touch2(a[u]);
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < NB; ++j) pw_touch2(b[u][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) X[((size_t)t0 * 16 + (lane >> 4) + 4 * r) * ldx + (size_t)(cf0 + j) * 16 + (lane & 15)] = acc[j][r];
  if (tlog && lane == 0) { tlog[2 * (size_t)task] = t_begin; tlog[2 * (size_t)task + 1] = __builtin_amdgcn_s_memrealtime(); }
}

int main(int argc, char** argv) {
  const size_t m = argc > 1 ? std::atol(argv[1]) : 30801;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
  const size_t M_pad = (m + 127) / 128 * 128, nbi = M_pad / 128, nkb = M_pad / 16;
  const size_t nL = tr_total_chunks(nbi) * TR_CHUNK;
  const int ntile = (int)((m + 15) / 16);
  const size_t nLs = (size_t)128 * ntile * (ntile + 1);                  // tile streams
  const size_t ld = 256, dstride = (size_t)ntile * 2 * 128 + 128;
  double *L, *Ls, *D, *Ds, *X;
  unsigned long long* tlog;
  CK(hipMalloc(&L, (nL + TS_SLACK_DOUBLES) * 8)); CK(hipMalloc(&Ls, nLs * 8)); CK(hipMalloc(&D, (2 * nkb * TR_CHUNK + TS_SLACK_DOUBLES) * 8)); CK(hipMalloc(&Ds, 8 * dstride * 8)); CK(hipMalloc(&X, M_pad * ld * 8));
  CK(hipMalloc(&tlog, (size_t)ntile * 8 * 2 * 8));
  k_fill<<<4096, 256>>>(L, nL, 1); k_fill<<<4096, 256>>>(Ls, nLs, 2); k_fill<<<1024, 256>>>(D, 2 * nkb * TR_CHUNK, 3); k_fill<<<256, 256>>>(Ds, 8 * dstride, 4);
  CK(hipDeviceSynchronize());
  const double bytes_tri = (double)ntile * (ntile + 1) / 2 * 4 * 512;     // bytes of the factor a launch must read (tiles to their diagonals)
  std::printf("m=%zu tiles=%d chunk stream %.2f GB, bytes to the diagonals %.3f GB\n", m, ntile, nL * 8 / 1e9, bytes_tri / 1e9);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, int ntask_log) {
    launch(nullptr); launch(nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0)); launch(nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
    }
    CK(hipGetLastError());
    if (!ntask_log) { std::printf("%-44s %7.3f ms best %7.3f avg  %6.0f GB/s\n", name, best, sum / reps, bytes_tri / (best * 1e-3) / 1e9); std::fflush(stdout); return; }
    // timeline of one launch: when did the tasks end (100 MHz ticks)
    CK(hipMemset(tlog, 0, (size_t)ntile * 8 * 2 * 8));
    launch(tlog); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> tl((size_t)ntask_log * 2);
    CK(hipMemcpy(tl.data(), tlog, tl.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < ntask_log; ++i) if (tl[2 * i]) { t0 = std::min(t0, tl[2 * i]); t1 = std::max(t1, tl[2 * i + 1]); }
    auto endof = [&](int i) { return (double)(tl[2 * i + 1] - t0) / 100.0; };   // us
    std::printf("%-44s %7.3f ms best %7.3f avg  %6.0f GB/s | span %6.1f us; end of task 0 (longest) %6.1f, N/4 %6.1f, N/2 %6.1f, 3N/4 %6.1f, last %6.1f us; start spread %5.1f us\n", name, best,
                sum / reps, bytes_tri / (best * 1e-3) / 1e9, (double)(t1 - t0) / 100.0, endof(0), endof(ntask_log / 4), endof(ntask_log / 2), endof(3 * ntask_log / 4), endof(ntask_log - 1),
                [&] { unsigned long long mx = 0; for (int i = 0; i < ntask_log; ++i) mx = std::max(mx, tl[2 * i]); return (double)(mx - t0) / 100.0; }());
    std::fflush(stdout);
  };
#define P0(NB, PD, MODE, W8) { StreamGeom g{ntile, 1, ntile}; run("chunk stream dwordx2 NB=" #NB " PD=" #PD " MODE=" #MODE " W8=" #W8, [&](unsigned long long* tl) { \
    hipLaunchKernelGGL((k_probe0<NB, PD, MODE, W8>), dim3((unsigned)((g.ntask + (W8 ? 7 : 3)) / (W8 ? 8 : 4))), dim3(W8 ? 512 : 256), 0, 0, L, D, X, g, nkb, ld, tl); }, g.ntask); }
#define P1(NB, PD, MODE) { StreamGeom g{ntile, 1, ntile}; run("tile stream  dwordx4 NB=" #NB " PD(pairs)=" #PD " MODE=" #MODE, [&](unsigned long long* tl) { \
    hipLaunchKernelGGL((k_probe1<NB, PD, MODE>), dim3((unsigned)((g.ntask + 7) / 8)), dim3(512), 0, 0, Ls, Ds, X, g, dstride, ld, tl); }, g.ntask); }
#define PS(RT, NB, PD, HALF) PSC(RT, NB, PD, HALF, 1)
#define PSC(RT, NB, PD, HALF, NCG) { StreamGeom g{(ntile + RT - 1) / RT, NCG, (ntile + RT - 1) / RT * NCG}; run("k_trmm_stream RT=" #RT " NB=" #NB " PD=" #PD " HALF=" #HALF " NCG=" #NCG, [&](unsigned long long* tl) { \
    hipLaunchKernelGGL((k_trmm_stream<RT, NB, PD, HALF, 0>), dim3((unsigned)((g.ntask + 2 * HALF - 1) / (2 * HALF))), dim3(128 * HALF), 0, 0, L, D, X, g, nkb, ld, M_pad); }, 0); }
  PS(2, 1, 16, 2) PS(2, 1, 20, 2) PS(2, 1, 12, 2)
  std::printf("-- 32 preimages\n");
  PSC(2, 1, 16, 2, 2) PSC(2, 1, 16, 4, 2) PSC(1, 1, 16, 4, 2) PSC(2, 2, 8, 2, 1) PSC(2, 2, 16, 2, 1) PSC(4, 1, 12, 2, 2) PSC(4, 1, 12, 4, 2)
  std::printf("-- 64 preimages\n");
  PSC(2, 1, 16, 2, 4) PSC(2, 1, 16, 4, 4) PSC(2, 2, 16, 2, 2) PSC(2, 2, 16, 4, 2) PSC(4, 1, 12, 4, 4) PSC(4, 2, 10, 4, 2) PSC(4, 2, 10, 2, 2) PSC(2, 4, 8, 4, 1) PSC(1, 4, 8, 4, 1)
  std::printf("-- 128 preimages\n");
  PSC(2, 2, 16, 4, 4) PSC(4, 2, 10, 4, 4) PSC(2, 4, 8, 4, 2) PSC(4, 4, 8, 4, 2) PSC(4, 4, 8, 2, 2) PSC(8, 2, 6, 4, 4)
  return 0;
  P0(1, 32, 0, 1) P0(1, 16, 0, 1) P0(1, 8, 0, 1) P0(1, 32, 0, 0)
  P0(1, 32, 1, 1) P0(1, 64, 1, 1) P0(1, 32, 2, 1)
  P1(1, 16, 0) P1(1, 24, 0) P1(1, 32, 1) P1(1, 24, 2)
  P0(2, 22, 0, 1) P0(4, 13, 0, 1) P0(4, 13, 1, 1) P0(4, 13, 2, 1)
  P1(2, 16, 0) P1(4, 8, 0) P1(4, 12, 0)
  std::printf("-- 256 preimages\n");
  PSC(2, 4, 8, 4, 4) PSC(4, 2, 10, 4, 8) PSC(4, 4, 6, 4, 4) PSC(4, 4, 5, 4, 4)
  return 0;
}
