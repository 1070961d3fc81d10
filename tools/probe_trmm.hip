// Measurement harness (not product code) for the FP64 triangular product at the C3 shape with random operands: the shipped kernel and experimental
// variants of its K loop, each timed alone.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DTRMM_CLOCK_PROBE tools/probe_trmm.hip -o tools/bin/probe_trmm
//   tools/bin/probe_trmm [nbi=240] [nbj=32] [reps=3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "../tools_amd/csrc/psf_rng.hpp"
#include "../tools_amd/csrc/psf_kernels.hpp"
using namespace psf;

__global__ void k_fill(double* p, size_t n, uint64_t salt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = (i + salt) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = (double)(int64_t)(x >> 11) * 0x1.0p-52 - 1.0;
  }
}

// MODE 0: register-streamed as shipped (TR_PD steps in flight); 1: same, every load re-reads the first eight steps (cache-resident: what the issue pattern
// alone delivers); 2: no loads inside the loop at all (operands of the prologue reused)
template <int MODE, int PD, int AGPR = 0>
__global__ __launch_bounds__(256, 2) void k_reg_variant(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X, int nbi, int nbj, size_t nkb,
                                                        size_t ldx, int GR, int GC) {
  int bi, bj;
  tr_map_block(blockIdx.x, nbi, nbj, GR, GC, &bi, &bj);
  if (bi < 0 || bi >= nbi || bj >= nbj) return;
  TR_CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nsteps = TR_KB_PER_BLOCK * (bi + 1) * (TR_BK / 4);
  const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + (size_t)(wr * 4) * 64;
  const double* gB = Dt + (size_t)bj * nkb * TR_CHUNK + (size_t)(wc * 4) * 64;
  const uint32_t voff = (uint32_t)lane * 8u;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  double a[PD][4], b[PD][4];
  auto issue = [&](double (&av)[4], double (&bv)[4], int s) {
    if (MODE == 1) s &= 7;
    const double* pa = gA + (size_t)s * 512;
    const double* pb = gB + (size_t)s * 512;
    TR_LOAD8(av[0], voff, pa, 0); TR_LOAD8(bv[0], voff, pb, 0);
    TR_LOAD8(av[1], voff, pa, 512); TR_LOAD8(bv[1], voff, pb, 512);
    TR_LOAD8(av[2], voff, pa, 1024); TR_LOAD8(bv[2], voff, pb, 1024);
    TR_LOAD8(av[3], voff, pa, 1536); TR_LOAD8(bv[3], voff, pb, 1536);
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) issue(a[u], b[u], u);
  if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (s0 + u < nsteps) {
        if (MODE != 2) {
          if (PD == 4) TR_WAIT(24, a[u], b[u]);
          if (PD == 6) TR_WAIT(40, a[u], b[u]);
          if (PD == 3) TR_WAIT(16, a[u], b[u]);
          if (PD == 2) TR_WAIT(8, a[u], b[u]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (AGPR) asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[u][i]), "v"(b[u][j]));      // accumulators in AccVGPRs
            else acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
          }
      }
      if (MODE != 2) {
        int sn = s0 + u + PD;
        sn = sn < nsteps ? sn : nsteps - 1;
        issue(a[u], b[u], sn);
      }
    }
  }
  // the re-reads behind the last step are never consumed: their destination registers must stay allocated until they have landed
#pragma unroll
  for (int u = 0; u < PD; ++u) TR_WAIT(0, a[u], b[u]);
  if (AGPR) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");    // the hazard recogniser does not see the MFMAs inside asm statements
  const size_t row0 = (size_t)bi * TR_BM + wr * 64, col0 = (size_t)bj * TR_BN + wc * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) X[(row0 + i * 16 + (lane >> 4) + 4 * r) * ldx + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
  TR_CLK_END
}

// One workgroup per CU: 4 waves, each alone on its SIMD with a 128 x 64 tile (256 AccVGPRs of accumulators, MFMAs as asm statements), workgroup tile 256 x 128 = two
// row-blocks x one column block.  A wave streams all 8 A fragments of its row-block and 4 B fragments per k-step (12 loads for 32 MFMAs: 25 % fewer operand bytes per flop than
// two 128 x 128 workgroups per CU), PD steps in flight.  No co-resident partner: nothing hides a stall, but nothing competes either (every workgroup runs at the same pace).
#define TR_WAIT12(n, A, Bv) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), \
                                         "+v"(Bv[0]), "+v"(Bv[1]), "+v"(Bv[2]), "+v"(Bv[3]))
template <int PD, int MODE>
__global__ __launch_bounds__(256, 1) void k_big_variant(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X, int nbi, int nbj, size_t nkb,
                                                        size_t ldx, int GR, int GC) {
  int bt, bj;
  const int nbt = nbi / 2;                                           // row tiles of 256 (the probe's nbi is even)
  tr_map_block(blockIdx.x, nbt, nbj, GR, GC, &bt, &bj);
  if (bt < 0 || bt >= nbt || bj >= nbj) return;
  TR_CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int bi = 2 * bt + wr;
  const int nsteps = TR_KB_PER_BLOCK * (bi + 1) * (TR_BK / 4);
  const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK;
  const double* gB = Dt + (size_t)bj * nkb * TR_CHUNK + (size_t)(wc * 4) * 64;
  const uint32_t voff = (uint32_t)lane * 8u;
  d4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  double a[PD][8], b[PD][4];
  auto issue = [&](double (&av)[8], double (&bv)[4], int s) {
    if (MODE == 1) s &= 7;
    const double* pa = gA + (size_t)s * 512;
    const double* pb = gB + (size_t)s * 512;
#define TR_LOAD8NT(dst, voff, base, imm) asm volatile("global_load_dwordx2 %0, %1, %2 offset:" #imm " nt" : "=v"(dst) : "v"(voff), "s"(base) : "memory")
    if (MODE == 2 || MODE == 4) {
      TR_LOAD8NT(av[0], voff, pa, 0); TR_LOAD8NT(av[1], voff, pa, 512); TR_LOAD8NT(av[2], voff, pa, 1024); TR_LOAD8NT(av[3], voff, pa, 1536);
      TR_LOAD8NT(av[4], voff, pa, 2048); TR_LOAD8NT(av[5], voff, pa, 2560); TR_LOAD8NT(av[6], voff, pa, 3072); TR_LOAD8NT(av[7], voff, pa, 3584);
    } else {
    TR_LOAD8(av[0], voff, pa, 0); TR_LOAD8(av[1], voff, pa, 512); TR_LOAD8(av[2], voff, pa, 1024); TR_LOAD8(av[3], voff, pa, 1536);
    TR_LOAD8(av[4], voff, pa, 2048); TR_LOAD8(av[5], voff, pa, 2560); TR_LOAD8(av[6], voff, pa, 3072); TR_LOAD8(av[7], voff, pa, 3584);
    }
    if (MODE == 3 || MODE == 4) { TR_LOAD8NT(bv[0], voff, pb, 0); TR_LOAD8NT(bv[1], voff, pb, 512); TR_LOAD8NT(bv[2], voff, pb, 1024); TR_LOAD8NT(bv[3], voff, pb, 1536); }
    else { TR_LOAD8(bv[0], voff, pb, 0); TR_LOAD8(bv[1], voff, pb, 512); TR_LOAD8(bv[2], voff, pb, 1024); TR_LOAD8(bv[3], voff, pb, 1536); }
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) issue(a[u], b[u], u);
  for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (s0 + u < nsteps) {
        if (PD == 4) TR_WAIT12(36, a[u], b[u]);
        if (PD == 5) TR_WAIT12(48, a[u], b[u]);
        if (PD == 6) TR_WAIT12(60, a[u], b[u]);
        if (PD == 3) TR_WAIT12(24, a[u], b[u]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[u][i]), "v"(b[u][j]));
      }
      int sn = s0 + u + PD;
      sn = sn < nsteps ? sn : nsteps - 1;
      issue(a[u], b[u], sn);
    }
  }
#pragma unroll
  for (int u = 0; u < PD; ++u) TR_WAIT12(0, a[u], b[u]);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const size_t row0 = (size_t)bi * TR_BM, col0 = (size_t)bj * TR_BN + wc * 64;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) X[(row0 + i * 16 + (lane >> 4) + 4 * r) * ldx + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
  TR_CLK_END
}

// Experiment kept for the record (profiles/r02_notes.md): the register-streamed kernel in two phases -- phase 1 the chunks left of the row group's
// diagonal super-block (the same count for all 8 row-blocks of a super-tile), phase 2 the ragged rest continuing the chains from X -- with optional
// meeting points of the 64 workgroups of a super-tile in phase 1 (a counter per super-tile and meeting, 60 us timeout, no meetings after a timeout).
__device__ unsigned* g_probe_wait;
// the super-tile a workgroup belongs to and how many of its GR x GC positions hold a tile (edges of the block grid)
__device__ inline void tr_map_group(unsigned id, int nbi, int nbj, int GR, int GC, unsigned* group, unsigned* members) {
  const int ncg = (nbj + GC - 1) / GC, nrg = (nbi + GR - 1) / GR;
  const unsigned slot = id >> 3;
  const unsigned g = (slot / (GR * GC)) * 8u + (id & 7u);
  const int rg = nrg - 1 - (int)(g / ncg), cg = (int)(g % ncg);
  const int rows = nbi - rg * GR < GR ? nbi - rg * GR : GR, cols = nbj - cg * GC < GC ? nbj - cg * GC : GC;
  *group = g;
  *members = (unsigned)(rows * cols);
}
__host__ inline unsigned tr_group_count(int nbi, int nbj, int GR = 8, int GC = 8) {
  const int ncg = (nbj + GC - 1) / GC, nrg = (nbi + GR - 1) / GR;
  return (unsigned)((ncg * nrg + 7) / 8 * 8);
}
constexpr int TR_SYNC_STEPS = 512;      // k-steps (of 4 coordinates) between two meeting points of a super-tile's workgroups: 128 chunks, ~0.45 ms
constexpr int TR_SYNC_SLOTS = 16;       // 240 row-blocks x 32 steps / 512 = 15 meeting points at most for the supported sizes; beyond that no more meetings
__global__ __launch_bounds__(256, 2) void k_reg_phase_meet(const double* __restrict__ Lt, const double* __restrict__ Dt,
                                                         double* __restrict__ X, int nbi, int nbj, size_t nkb, size_t ldx, int GR, int GC, size_t row_hi, int phase,
                                                         unsigned* __restrict__ meet) {
  int bi, bj;
  tr_map_block(blockIdx.x, nbi, nbj, GR, GC, &bi, &bj);
  if (bi < 0 || bi >= nbi || bj >= nbj) return;
  TR_CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int s_split = TR_KB_PER_BLOCK * (bi - bi % GR) * (TR_BK / 4);      // phases: see k_trmm_f64
  const int sb = phase == 2 ? s_split : 0;
  const int nsteps = phase == 1 ? s_split : TR_KB_PER_BLOCK * (bi + 1) * (TR_BK / 4);
  if (sb >= nsteps) return;
  const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + (size_t)(wr * 4) * 64;      // wave-uniform
  const double* gB = Dt + (size_t)bj * nkb * TR_CHUNK + (size_t)(wc * 4) * 64;
  const uint32_t voff = (uint32_t)lane * 8u;
  const size_t row0 = (size_t)bi * TR_BM + wr * 64, col0 = (size_t)bj * TR_BN + wc * 64;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = sb > 0 ? X[(row0 + i * 16 + (lane >> 4) + 4 * r) * ldx + col0 + j * 16 + (lane & 15)] : 0.0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the vmcnt arithmetic below counts the operand loads only
  double a[TR_PD][4], b[TR_PD][4];
  auto issue = [&](double (&av)[4], double (&bv)[4], int s) {
    const double* pa = gA + (size_t)s * 512;
    const double* pb = gB + (size_t)s * 512;
    TR_LOAD8(av[0], voff, pa, 0); TR_LOAD8(bv[0], voff, pb, 0);
    TR_LOAD8(av[1], voff, pa, 512); TR_LOAD8(bv[1], voff, pb, 512);
    TR_LOAD8(av[2], voff, pa, 1024); TR_LOAD8(bv[2], voff, pb, 1024);
    TR_LOAD8(av[3], voff, pa, 1536); TR_LOAD8(bv[3], voff, pb, 1536);
  };
#pragma unroll
  for (int u = 0; u < TR_PD; ++u) issue(a[u], b[u], sb + u);               // at least 32 steps in every phase
  // Meeting points (phase 1 only: there every workgroup of a super-tile runs the same number of steps).  The 64 workgroups of a super-tile are the
  // resident set of one XCD; they drift apart by ~1.5 % of the run time (the older workgroup of a CU wins the MFMA arbitration) and every new
  // super-tile inherits the stagger of the one before, and once two sharers of a chunk are more than the L2's few steps of history apart each fetches
  // it on its own (measured: D is then fetched once per row-block, 4x the traffic).  A meeting is a counter per (super-tile, meeting index): arrive,
  // then wait for the others -- but never longer than 60 us, and never again after one timeout (a workgroup whose mates are not resident, e.g. on a
  // partitioned or shared device, just runs on): it cannot deadlock.
  unsigned group = 0, members = 0;
  bool meeting = meet != nullptr && phase == 1;
  if (meeting) tr_map_group(blockIdx.x, nbi, nbj, GR, GC, &group, &members);
  int next_meet = sb, meet_idx = 0;
  for (int s0 = sb; s0 < nsteps; s0 += TR_PD) {
    if (meeting && s0 >= next_meet) {
      if (tid == 0) {
        unsigned* c = meet + (size_t)group * TR_SYNC_SLOTS + meet_idx;
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t_in = wall_clock64();
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members) {
          if (wall_clock64() - t_in > 6000ull) { meet[(size_t)group * TR_SYNC_SLOTS + TR_SYNC_SLOTS - 1] = ~0u; break; }      // last slot: "gave up" mark, read below
          __builtin_amdgcn_s_sleep(16);
        }
        if (g_probe_wait) g_probe_wait[(size_t)blockIdx.x * 16 + meet_idx] = (unsigned)(wall_clock64() - t_in) + 1u;
      }
      __syncthreads();
      next_meet += TR_SYNC_STEPS;
      if (++meet_idx >= TR_SYNC_SLOTS - 1 || __hip_atomic_load(meet + (size_t)group * TR_SYNC_SLOTS + TR_SYNC_SLOTS - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ~0u) meeting = false;
    }
#pragma unroll
    for (int u = 0; u < TR_PD; ++u) {
      if (s0 + u < nsteps) {                                           // the number of k-steps is a multiple of 32, not of TR_PD
        TR_WAIT(40, a[u], b[u]);                                       // 8 (TR_PD - 1)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
      }
      int sn = s0 + u + TR_PD;
      sn = sn < nsteps ? sn : nsteps - 1;                              // past the end: re-read the last step (never consumed)
      issue(a[u], b[u], sn);
    }
  }
  // the re-reads behind the last step are never consumed: their destination registers must stay allocated until they have landed
  // (a load that lands in a register the compiler has reused since would corrupt it)
#pragma unroll
  for (int u = 0; u < TR_PD; ++u) TR_WAIT(0, a[u], b[u]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
        const size_t row = row0 + i * 16 + (lane >> 4) + 4 * r;
        if (row < row_hi) X[row * ldx + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
      }
  TR_CLK_END
}


// LDS-fed with a five-deep ring of half chunks (8 coordinates: A 8 KiB | B 8 KiB), fragments double-buffered in registers, the barrier in the middle of
// a half chunk's MFMAs (nothing is requested from LDS right behind it), LDS-DMA three half chunks ahead.
#define DS_RD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n, A, Bv) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(Bv[0]), "+v"(Bv[1]), "+v"(Bv[2]), "+v"(Bv[3]))
template <int FAKE>
__global__ __launch_bounds__(256, 2) void k_ring_variant(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X, int nbi, int nbj, size_t nkb,
                                                         size_t ldx, int GR, int GC) {
  extern __shared__ __attribute__((aligned(16))) double smem[];      // 5 x 2048 doubles
  int bi, bj;
  tr_map_block(blockIdx.x, nbi, nbj, GR, GC, &bi, &bj);
  if (bi < 0 || bi >= nbi || bj >= nbj) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nh = 2 * TR_KB_PER_BLOCK * (bi + 1);                     // half chunks (two k-steps each)
  const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + lane * 2;
  const double* gB = Dt + (size_t)bj * nkb * TR_CHUNK + lane * 2;
  // half chunk h of a stream = doubles [1024 h, 1024 h + 1024): k-step 2h at +0 (8 fragments of 64), k-step 2h+1 at +512
  auto dma = [&](int h, int slot) {
    const int hh = FAKE ? (h & 7) : h;
    double* ls = smem + slot * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                    // 8 pieces of 128 doubles per operand half chunk, two per wave
      const int piece = wave * 2 + i;
      __builtin_amdgcn_global_load_lds(gA + (size_t)hh * 1024 + piece * 128, (lds_void_ptr)(ls + piece * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gB + (size_t)hh * 1024 + piece * 128, (lds_void_ptr)(ls + 1024 + piece * 128), 16, 0, 0);
    }
  };
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  const uint32_t lbase = (uint32_t)(uintptr_t)(lds_void_ptr)smem;
  const uint32_t la0 = lbase + (uint32_t)((wr * 4) * 64 + lane) * 8u, lb0 = lbase + (uint32_t)(1024 + (wc * 4) * 64 + lane) * 8u;
  double a0[4], b0[4], a1[4], b1[4];
  auto rd = [&](double (&av)[4], double (&bv)[4], int slot, int ks) {
    const uint32_t pa = la0 + (uint32_t)(slot * 2048 + ks * 512) * 8u, pb = lb0 + (uint32_t)(slot * 2048 + ks * 512) * 8u;
    DS_RD(av[0], pa, 0); DS_RD(bv[0], pb, 0); DS_RD(av[1], pa, 512); DS_RD(bv[1], pb, 512);
    DS_RD(av[2], pa, 1024); DS_RD(bv[2], pb, 1024); DS_RD(av[3], pa, 1536); DS_RD(bv[3], pb, 1536);
  };
  auto mm = [&](double (&av)[4], double (&bv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
  };
  // prologue: half chunks 0..3 in flight (nh >= 16), 0 visible
  dma(0, 0); dma(1, 1); dma(2, 2); dma(3, 3);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __syncthreads();
  rd(a0, b0, 0, 0);
  int slot = 0;
  for (int h = 0; h < nh; ++h) {
    const int nslot = slot == 4 ? 0 : slot + 1;
    rd(a1, b1, slot, 1);
    LGKM_WAIT(8, a0, b0);
    mm(a0, b0);
    // half chunk h+1 has landed for this wave (two younger half chunks may be in flight); behind the barrier it is visible to all, and every wave has
    // left half chunk h-1, whose slot takes half chunk h+4
    if (h + 3 < nh) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (h + 4 < nh) dma(h + 4, slot == 0 ? 4 : slot - 1);
    if (h + 1 < nh) rd(a0, b0, nslot, 0);
    LGKM_WAIT(8, a1, b1);                                             // with no new read behind it (last half chunk) this is conservative: lgkmcnt(0) would be exact
    if (h + 1 >= nh) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mm(a1, b1);
    slot = nslot;
  }
  const size_t row0 = (size_t)bi * TR_BM + wr * 64, col0 = (size_t)bj * TR_BN + wc * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) X[(row0 + i * 16 + (lane >> 4) + 4 * r) * ldx + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
}

int main(int argc, char** argv) {
  const int nbi = argc > 1 ? atoi(argv[1]) : 240, nbj = argc > 2 ? atoi(argv[2]) : 32, reps = argc > 3 ? atoi(argv[3]) : 3;
  const unsigned mask = argc > 4 ? (unsigned)strtoul(argv[4], nullptr, 0) : 0xffffffffu;      // bit i: run variant i (variant 0 always runs: it is the reference)
  const size_t nkb = (size_t)nbi * TR_KB_PER_BLOCK + (argc > 7 ? atoi(argv[7]) : 0), ldx = (size_t)nbj * 128;      // argv[7]: chunks of padding between the D streams
  const size_t nL = tr_total_chunks(nbi) * TR_CHUNK, nD = (size_t)nbj * nkb * TR_CHUNK, nX = (size_t)nbi * 128 * ldx;
  double *L, *D, *X, *Xref;
  hipMalloc(&L, (nL + 65536) * 8); hipMalloc(&D, (nD + 65536) * 8); hipMalloc(&X, nX * 8); hipMalloc(&Xref, nX * 8);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, L, nL, 1ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, D, nD, 77ull);
  hipDeviceSynchronize();
  double flops = 0;
  for (int bi = 0; bi < nbi; ++bi) flops += 2.0 * 128 * 128 * 128 * (bi + 1) * nbj;
  const int GR = argc > 5 ? atoi(argv[5]) : 8, GC = argc > 6 ? atoi(argv[6]) : 8;      // super-tile shape, GR * GC = 64
  const unsigned grid = tr_grid_size(nbi, nbj, GR, GC);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_f64), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TR_CHUNK * sizeof(double));
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring_variant<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 2048 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring_variant<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 2048 * 8);
  unsigned* meet; hipMalloc(&meet, (size_t)tr_group_count(nbi, nbj, GR, GC) * TR_SYNC_SLOTS * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> href(1 << 16), hx(1 << 16);
  auto run = [&](const char* name, int which, bool check) {
    if (which && !((mask >> which) & 1u)) return;
    fprintf(stderr, "running %s\n", name);
    hipMemset(X, 0, nX * 8);
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
      hipEventRecord(e0, 0);
      switch (which) {
        case 0: hipLaunchKernelGGL(k_trmm_f64, dim3(grid), dim3(256), 4 * TR_CHUNK * sizeof(double), 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC, (size_t)nbi * 128); break;
        case 19: hipLaunchKernelGGL((k_big_variant<4, 0>), dim3(tr_grid_size(nbi / 2, nbj, 4, 8)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 4, 8); break;
        case 20: hipLaunchKernelGGL((k_big_variant<5, 0>), dim3(tr_grid_size(nbi / 2, nbj, 4, 8)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 4, 8); break;
        case 21: hipLaunchKernelGGL((k_big_variant<4, 1>), dim3(tr_grid_size(nbi / 2, nbj, 4, 8)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 4, 8); break;
        case 22: hipLaunchKernelGGL((k_big_variant<4, 0>), dim3(tr_grid_size(nbi / 2, nbj, 8, 4)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 8, 4); break;
        case 23: hipLaunchKernelGGL((k_big_variant<4, 0>), dim3(tr_grid_size(nbi / 2, nbj, 2, 16)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 2, 16); break;
        case 24: hipLaunchKernelGGL((k_big_variant<4, 0>), dim3(tr_grid_size(nbi / 2, nbj, 16, 2)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 16, 2); break;
        case 25: hipLaunchKernelGGL((k_big_variant<4, 0>), dim3(tr_grid_size(nbi / 2, nbj, 32, 1)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 32, 1); break;
        case 26: hipLaunchKernelGGL((k_big_variant<4, 0>), dim3(tr_grid_size(nbi / 2, nbj, 1, 32)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 1, 32); break;
        case 27: hipLaunchKernelGGL((k_big_variant<6, 0>), dim3(tr_grid_size(nbi / 2, nbj, 8, 4)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 8, 4); break;
        case 28: hipLaunchKernelGGL(k_trmm_f64_big, dim3(tr_grid_size((nbi + 1) / 2, nbj, 8, 4)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 8, 4, (size_t)nbi * 128); break;
        case 29: hipLaunchKernelGGL((k_big_variant<4, 2>), dim3(tr_grid_size(nbi / 2, nbj, 8, 4)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 8, 4); break;
        case 30: hipLaunchKernelGGL((k_big_variant<4, 3>), dim3(tr_grid_size(nbi / 2, nbj, 8, 4)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 8, 4); break;
        case 31: hipLaunchKernelGGL((k_big_variant<4, 4>), dim3(tr_grid_size(nbi / 2, nbj, 8, 4)), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, 8, 4); break;
        case 12: hipLaunchKernelGGL(k_trmm_f64_reg, dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC, (size_t)nbi * 128); break;
        case 13: for (int ph = 1; ph <= 2; ++ph) hipLaunchKernelGGL(k_reg_phase_meet, dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC, (size_t)nbi * 128, ph, (unsigned*)nullptr); break;
        case 15: hipMemsetAsync(meet, 0, (size_t)tr_group_count(nbi, nbj, GR, GC) * TR_SYNC_SLOTS * 4, 0);
                 for (int ph = 1; ph <= 2; ++ph) hipLaunchKernelGGL(k_reg_phase_meet, dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC, (size_t)nbi * 128, ph, meet); break;
        case 1: hipLaunchKernelGGL((k_reg_variant<0, 4>), dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
        case 2: hipLaunchKernelGGL((k_reg_variant<1, 4>), dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
        case 3: hipLaunchKernelGGL((k_reg_variant<2, 4>), dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
        case 4: hipLaunchKernelGGL((k_reg_variant<0, 6>), dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
        case 5: hipLaunchKernelGGL((k_reg_variant<0, 3>), dim3(grid), dim3(256), 60000, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;   // 60000 B of unused LDS: two workgroups per CU
        case 10: hipLaunchKernelGGL((k_reg_variant<0, 3>), dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;      // 179 VGPRs: three workgroups per CU fit
        case 8: hipLaunchKernelGGL((k_reg_variant<2, 4, 1>), dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
        case 9: hipLaunchKernelGGL((k_reg_variant<0, 6, 1>), dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
        case 6: hipLaunchKernelGGL((k_ring_variant<0>), dim3(grid), dim3(256), 5 * 2048 * 8, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
        case 7: hipLaunchKernelGGL((k_ring_variant<1>), dim3(grid), dim3(256), 5 * 2048 * 8, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC); break;
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best; sum += ms;
    }
    const hipError_t err = hipGetLastError();
    unsigned long long clk[4] = {0, 0, 0, 0};
#ifdef TRMM_CLOCK_PROBE
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_trmm_clk), sizeof(clk));
    { unsigned long long z[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_trmm_clk), z, sizeof(z)); }
#endif
    const double mhz = clk[1] ? (double)clk[0] / (double)clk[1] * 100.0 : 0.0;
    long bad = -1;
    if (which == 0) hipMemcpy(Xref, X, nX * 8, hipMemcpyDeviceToDevice);
    else if (check) {                                                 // bitwise against the shipped kernel on a strided sample
      bad = 0;
      for (size_t off = 0; off + (1 << 16) <= nX; off += nX / 37 / 8 * 8 + 8) {
        hipMemcpy(href.data(), Xref + off, href.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), X + off, hx.size() * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < href.size(); ++i) bad += href[i] != hx[i];
      }
    }
    printf("%-62s best %8.3f ms  avg %8.3f ms  %6.2f TFLOP/s (best)  mismatches vs shipped: %ld  clock %.0f MHz (peak there %.2f)  %s\n", name, best, sum / reps, flops / best * 1e-9, bad, mhz, 256 * 4 * 32 * mhz * 1e-6, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
  };
  run("k_trmm_f64 (shipped: LDS-DMA, 2 stages of 16 coordinates)", 0, false);
  run("registers, 4 k-steps in flight", 1, true);
  run("registers, 4 in flight, loads re-read 8 steps (cache hits)", 2, false);
  run("registers, no loads in the loop (issue pattern only)", 3, false);
  run("registers, 6 k-steps in flight", 4, true);
  run("registers, 3 k-steps in flight, 2 workgroups / CU", 5, true);
  run("LDS ring of 5 half chunks, mid-MFMA barrier, frag double buffer", 6, true);
  run("same ring, loads re-read 8 half chunks (cache hits)", 7, false);
  run("registers, no loads in the loop, accumulators in AccVGPRs", 8, false);
  run("registers, 6 in flight, accumulators in AccVGPRs", 9, true);
  run("k_trmm_f64 (shipped) again", 0, false);
#ifdef TRMM_CLOCK_PROBE
  if ((mask >> 14) & 1u) {      // timeline of the shipped LDS kernel: when do the 64 workgroups of a super-tile start and end?
    unsigned long long* dlog; hipMalloc(&dlog, (size_t)grid * 16); hipMemset(dlog, 0, (size_t)grid * 16);
    hipMemcpyToSymbol(HIP_SYMBOL(g_trmm_log), &dlog, sizeof(dlog));
    hipLaunchKernelGGL(k_trmm_f64, dim3(grid), dim3(256), 4 * TR_CHUNK * sizeof(double), 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC, (size_t)nbi * 128);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hl((size_t)grid * 2); hipMemcpy(hl.data(), dlog, (size_t)grid * 16, hipMemcpyDeviceToHost);
    unsigned long long* nul = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(g_trmm_log), &nul, sizeof(nul));
    unsigned long long base = ~0ull; for (unsigned i = 0; i < grid; ++i) if (hl[2 * i] && hl[2 * i] < base) base = hl[2 * i];
    for (int gen = 0; gen < 3; ++gen) {       // XCD 0: super-tiles 0, 8, 16 (slots 64 gen .. 64 gen + 63)
      printf("XCD 0, super-tile generation %d: start / end in us relative to the first workgroup of the launch, by position t = 8 r + c inside the super-tile\n", gen);
      for (int t = 0; t < 64; ++t) {
        const unsigned id = (unsigned)((gen * 64 + t) * 8);
        printf("  t=%2d %9.1f %9.1f%s", t, (hl[2 * id] - base) * 0.01, (hl[2 * id + 1] - base) * 0.01, t % 4 == 3 ? "\n" : "");
      }
    }
  }
#endif
  run("k_trmm_f64_reg (library, 6 in flight)", 12, true);
  run("k_trmm_f64_big (library default: one workgroup / CU, 8 x 4 super-tiles)", 28, true);
  run("one workgroup / CU, wave tile 128 x 64 in AccVGPRs, 4 in flight, 4 x 8 super-tiles", 19, true);
  run("same, 5 in flight", 20, true);
  run("same, 4 in flight, loads re-read 8 steps (cache hits)", 21, false);
  run("same, 4 in flight, 8 x 4 super-tiles", 22, true);
  run("same, 4 in flight, 2 x 16 super-tiles", 23, true);
  run("same, 4 in flight, 16 x 2 super-tiles", 24, true);
  run("same, 4 in flight, 32 x 1 super-tiles", 25, true);
  run("same, 4 in flight, 1 x 32 super-tiles", 26, true);
  run("same, 6 in flight, 8 x 4 super-tiles", 27, true);
  run("same, 4 in flight, 8 x 4, A loads non-temporal", 29, true);
  run("same, 4 in flight, 8 x 4, B loads non-temporal", 30, true);
  run("same, 4 in flight, 8 x 4, all loads non-temporal", 31, true);
  run("k_trmm_f64_reg in two phases", 13, true);
  run("k_trmm_f64_reg in two phases, meeting points in phase 1", 15, true);
#ifdef TRMM_CLOCK_PROBE
  if ((mask >> 16) & 1u) {      // phase 1 with meeting points: did any workgroup give up, and how far apart do the workgroups of a super-tile start and end?
    unsigned long long* dlog; hipMalloc(&dlog, (size_t)grid * 16); hipMemset(dlog, 0, (size_t)grid * 16);
    hipMemcpyToSymbol(HIP_SYMBOL(g_trmm_log), &dlog, sizeof(dlog));
    const unsigned ng = tr_group_count(nbi, nbj, GR, GC);
    hipMemset(meet, 0, (size_t)ng * TR_SYNC_SLOTS * 4);
    unsigned* dwait; hipMalloc(&dwait, (size_t)grid * 64); hipMemset(dwait, 0, (size_t)grid * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wait), &dwait, sizeof(dwait));
    hipLaunchKernelGGL(k_reg_phase_meet, dim3(grid), dim3(256), 0, 0, L, D, X, nbi, nbj, nkb, ldx, GR, GC, (size_t)nbi * 128, 1, meet);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hl((size_t)grid * 2); hipMemcpy(hl.data(), dlog, (size_t)grid * 16, hipMemcpyDeviceToHost);
    std::vector<unsigned> hm((size_t)ng * TR_SYNC_SLOTS); hipMemcpy(hm.data(), meet, hm.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long* nul = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(g_trmm_log), &nul, sizeof(nul));
    { std::vector<unsigned> hw((size_t)grid * 16); hipMemcpy(hw.data(), dwait, hw.size() * 4, hipMemcpyDeviceToHost);
      unsigned* nulw = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wait), &nulw, sizeof(nulw));
      for (int gen = 0; gen < 2; ++gen)
        for (int idx = 0; idx < 3; ++idx) {
          printf("XCD 0 generation %d meeting %d: us waited by t = 0..63:", gen, idx);
          for (int t = 0; t < 64; ++t) { const unsigned id = (unsigned)((gen * 64 + t) * 8); const unsigned w = hw[(size_t)id * 16 + idx]; if (w) printf(" %.0f", (w - 1) * 0.01); else printf(" -"); }
          printf("\n");
        } }
    int gave_up = 0; for (unsigned g = 0; g < ng; ++g) gave_up += hm[(size_t)g * TR_SYNC_SLOTS + TR_SYNC_SLOTS - 1] == ~0u;
    printf("meeting points: %d of %u super-tiles had a workgroup give up; counters of super-tile 0:", gave_up, ng);
    for (int i = 0; i < TR_SYNC_SLOTS; ++i) printf(" %u", hm[i]); printf("\n");
    unsigned long long base = ~0ull; for (unsigned i = 0; i < grid; ++i) if (hl[2 * i] && hl[2 * i] < base) base = hl[2 * i];
    for (int gen = 0; gen < 6; ++gen) {
      double s0 = 1e30, s1 = 0, e0 = 1e30, e1 = 0;
      for (int t = 0; t < 64; ++t) { const unsigned id = (unsigned)((gen * 64 + t) * 8); if (id >= grid || !hl[2 * id]) continue;
        const double a = (hl[2 * id] - base) * 0.01, b = (hl[2 * id + 1] - base) * 0.01; s0 = a < s0 ? a : s0; s1 = a > s1 ? a : s1; e0 = b < e0 ? b : e0; e1 = b > e1 ? b : e1; }
      printf("XCD 0 generation %d: starts %.1f .. %.1f us, ends %.1f .. %.1f us\n", gen, s0, s1, e0, e1);
    }
  }
#endif
  run("registers, 3 k-steps in flight, occupancy not limited", 10, true);
  return 0;
}
