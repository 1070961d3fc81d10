#pragma once
// psf_stream_kernels.hpp -- the SINGLE-CALL / SMALL-BATCH form of x = sqrt(Sigma_2) d  (mp_perturbation.rs:315; psf.rs:48-80: one `samp_p` call is
// one preimage, and that is what benches/psf.rs:38,63-65,90-92 time).
//
// With a handful of preimages the product is a matrix-VECTOR product: every byte of the factor is used once, 16 B flop per byte at most, so the
// bound is reading the 3.8 GB chunk stream (C3) from HBM once -- not the FP64 pipe that bounds k_trmm_f64_big at batch 4096.  k_trmm_f64_big
// cannot get there: its 256 x 128 tiles give 121 workgroups for any batch <= 128 and the bottom one walks 7 712 k-steps x 32 MFMAs alone (7 ms).
//
// k_trmm_stream: the unit of work is ONE 16-row tile of the factor (one MFMA accumulator row) x NB column fragments of 16 preimages, owned by ONE
// wave from k = 0 to the diagonal -- every element of X is still one ascending-k fma chain from +0 (v_mfma_f64_16x16x4_f64 is an ascending chain,
// profiles/r01_probe_mfma_f64.log), NO split-K, so the bits are those of the oracle and of the three batch kernels.  The chunk stream is already
// fragment-ordered per 16-row tile: k-step s of tile t is the 512 bytes at (rowblock_base(t / 8) * 2048 + s * 512 + (t % 8) * 64 + lane) doubles, one
// global_load_dwordx2 per lane.  A tile stops at its diagonal (4 (t + 1) k-steps; the zero fragments behind it inside the diagonal chunk are skipped --
// they would add exact zeros).
//
// Balance without split-K: the 16-row tiles have lengths 4, 8, ..., m / 4 k-steps.  Task i (descending length) and task N - 1 - i add up to the same
// length for every i, so a workgroup of eight waves takes four (long, short) pairs: waves w and w + 4 of a 512-thread workgroup share a SIMD
// (MI355X_MICROARCH.md, LDS section: waves are dealt to SIMDs cyclically), every SIMD of every workgroup gets the same number of MFMAs, and all tiles
// start together: the launch lasts as long as reading the factor takes, plus the tail of the longest chain.
// PD k-steps of operands are in flight per wave in a register ring (loads and waits are asm volatile: hipcc drains an unrolled ring with vmcnt(0));
// the 6-bit vmcnt allows 63 loads in flight, so PD <= 63 / (RT + NB) + 1.  Eight waves per CU with eight k-steps each already saturate what a CU
// takes from HBM (tools/probe_stream.hip: PD 8 / 16 / 32 within 4 %); what a wave must not do is spend issue cycles per k-step beside its loads.
constexpr size_t TS_SLACK_DOUBLES = 2 * 16 * 128 * 64;      // doubles (2 MiB) the host allocates behind the factor's and the normals' streams: up to 16 k-steps of over-read (2 PD of k_trmm_stream, the ring of k_trmm_stream_wg), 128 fragments x 512 B each for the compact normals stream at 2048 preimages = 1 MiB
#include <type_traits>
#include "psf_kernels.hpp"

namespace psf {

template <int N> __device__ inline void ts_wait(double& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }
__device__ inline void ts_touch(double& x) { asm volatile("" : "+v"(x)); }
template <int OFF> __device__ inline void ts_load(double& dst, uint32_t voff, const double* base) {
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
#ifndef PSF_TS_NT
#define PSF_TS_NT 1      /* -DPSF_TS_NT=0: plain loads (A/B builds) */
#endif
// the same load with the non-temporal hint, for bytes that are read ONCE per call (the factor with one column group): 0.632-0.665 -> 0.605-0.629 ms for the product of one
// preimage at C3, same box (round 6).  Measured and not kept elsewhere: non-temporal LDS-DMA of the factor's pieces in k_trmm_stream_wg32 / _wg (0.69 -> 0.82 ms at 32,
// 0.92-0.96 -> 0.99-1.00 at 64 preimages), non-temporal loads of A in k_round_syndrome_small (28 -> 29-30 us).
template <int OFF> __device__ inline void ts_load_nt(double& dst, uint32_t voff, const double* base) {
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");      // (with sc1 / sc0 sc1 / sc0 beside nt: within the noise)
}
template <int I, int N, class F> __device__ inline void ts_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ts_for<I + 1, N>(f); }
}

struct StreamGeom {
  int ntile;        // groups of RT 16-row tiles (ceil(rows / (16 RT)))
  int ncg;          // column groups of 16 NB preimages
  int ntask;        // ntile * ncg
  int bc;           // CD == 2: preimages the dense normals stream stores per k-step (1, 2, 4, 8 or 16)
  int ncf = 0;      // k_trmm_stream_wg / _wg32 only: fragments per k-step of the compact normals stream when it is NOT ncg x (the kernel's own group width) -- a batch of
  int cf_base = 0;  // 65 ... 96 preimages is one launch of each (64 + 32 columns) over one stream of 6 fragments; cf_base = the launch's first fragment
};

// ---- rounding and the syndrome of ONE or TWO preimages in one launch (round 6) -----------------------------------------------------------------------------
// A call of one preimage spent 61 us behind the product in two launches: p <- D_{Z,r,x} (25 us for 30 801 draws) and v = u - A p (36 us for one pass over A).
// k_round_syndrome_small does both: a wave owns RT 16-row tiles of x, rounds them and adds its share A[:, rows] p[rows] of the syndrome, so p never travels and one
// launch disappears (26 + 10 us for the pair; the call 0.825-0.848 -> 0.828-0.833 ms median, 0.814-0.819 -> 0.789-0.791 fastest, same box: tools/fused_skip_sweep2.sh).
//   rounding: 4 lanes per sample, lane s of a quad evaluates the attempt groups s, s + 4, ... of the sample's own Philox stream (sz_group4 / sz_group4_narrow: the
//             expressions of k_perturb_round_lean), the lowest accepting group of a round wins -- the first accepted attempt of the stream, i.e. the value every
//             other sampler of the library returns.  16 samples per pass.
//   syndrome: lane l owns the rows 8 l ... 8 l + 7 of A (n <= 512 per 64 lanes, more in further rounds), read from the TRANSPOSED compact copy A32T[coordinate][row]
//             (32 bytes per lane and coordinate, sixteen coordinates in flight), products with p in 64 bits (|a p| < 2^32 2^23), one residue per (row, preimage)
//             into part[row][b][task]; k_zq_combine_wave<true> sums the tasks and subtracts from u, as it does for the K splits of the other forms.  Exact
//             integers: any order, same v.
// MEASURED AND NOT KEPT: the same epilogue INSIDE the streaming product (every wave rounding its own rows as its chain ends).  It made the product 60-70 us slower
// (0.67 -> 0.73-0.74 ms) whether or not the longest tasks were excused from it: the epilogues of the short tasks disturb the stream they run beside (profiles/r06_notes.md).
struct StreamFuse {
  uint64_t seed, first_index;
  size_t m;                       // coordinates (rows of x that exist)
  SampleZParams sp;               // D_{Z, r, .}
  int32_t* P; size_t ldp;         // p as [coordinate][preimage]
  const uint32_t* A32T; size_t n; // A transposed, [coordinate][row], n rows
  uint64_t q;
  uint64_t* part;                 // [row][bc][task]
  int* fail;
};

// the epilogue of one task of the fused tail: sx holds the task's RT * 16 rows of x for its bc preimages ((local row) * bc + b); rounding, then the share of A p
template <int RT>
__device__ __forceinline__ void fused_tail_epilogue(const StreamFuse& F, int ntask, int bc, int t0, int tg, double* sx, int32_t* spv, int lane) {
    const int nrow = RT * 16, npos = nrow * bc;                       // samples of this wave: (local row rr, preimage b) at rr * bc + b
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int f = 0;
    const int sub = lane & 3, quad0 = lane & ~3;
    const float inv_s_f = (float)F.sp.inv_s;
    for (int p0 = 0; p0 < npos; p0 += 16) {
      const int pos = p0 + (lane >> 2);
      const int rr = pos / bc, b = pos - rr * bc;
      const uint32_t coord = (uint32_t)(t0 * 16 + rr);
      const bool live = pos < npos && (size_t)coord < F.m;
      const double c = live ? sx[pos] : 0.0;
      const uint64_t index = F.first_index + (uint64_t)b;
      const uint32_t idx_lo = (uint32_t)index, tw = tag_word(TAG_PERTURB, index);
      const SzRange rg = sz_range(c, F.sp);
      const float c_rel = (float)((double)rg.lo - c);
      const bool generic = !(fabs(c) < 0x1.0p40);
      bool found = !live;
      long long x = 0;
      for (uint32_t t = (uint32_t)sub; ; t += 4) {
        if (!__builtin_amdgcn_ballot_w64(!found)) break;
        bool acc1 = false;
        long long xl = 0;
        if (!found) {
          if (t >= kMaxAttempts / 4) { acc1 = true; f = 1; xl = (long long)floor(c + 0.5); }      // (every lane of the quad gets here in the same round: the lowest takes it)
          else acc1 = generic ? sz_group4(F.seed, coord, idx_lo, tw, t, rg, c, F.sp.inv_s, &xl)
                              : sz_group4_narrow(F.seed, coord, idx_lo, tw, t, rg, c, F.sp.inv_s, c_rel, inv_s_f, &xl);
        }
        const uint32_t qm = (uint32_t)(__builtin_amdgcn_ballot_w64(acc1) >> quad0) & 0xfu;
        const int src = qm ? quad0 + __builtin_ctz(qm) : lane;
        const long long xs = __shfl(xl, src);
        if (!found && qm) { x = xs; found = true; }
      }
      if (live && sub == 0) {
        if (x > kDigitRangeP || x < -kDigitRangeP) f = 1;             // the syndrome product needs |p| < 2^23
        F.P[(size_t)coord * F.ldp + (size_t)b] = (int32_t)x;
      }
      if (pos < npos && sub == 0) spv[pos] = live ? (int32_t)x : 0;
    }
    if (f) atomicOr(F.fail, 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // this task's share of A p: rows 8 lane ... of A (rounds of 512 rows), coordinates t0 * 16 ... + nrow - 1
    const size_t c0 = (size_t)t0 * 16;
    const int ncoord = (int)(c0 + nrow <= F.m ? (size_t)nrow : (F.m > c0 ? F.m - c0 : 0));
    uint64_t* mypart = F.part + (size_t)tg;                           // part[(row * bc + b) * ntask + task]: the tasks of one output side by side for the combine
    const bool pow2 = (F.q & (F.q - 1)) == 0;
    for (size_t j0 = 0; j0 < F.n; j0 += 512) {
      const size_t j = j0 + (size_t)lane * 8;
      long long sum[2][8];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int u = 0; u < 8; ++u) sum[b][u] = 0;
      if (j < F.n) {                                                  // (n is a multiple of 8 here: the host checks)
        // sixteen coordinates' loads in flight at once: under the product's own stream a dependent round trip to memory costs microseconds, so a loop that waited for
        // one coordinate at a time made this epilogue 110 us long (measured); two batches make it two round trips
        constexpr int CB = 16;
        for (int cb = 0; cb < nrow; cb += CB) {
          uint4 a0[CB], a1[CB];
#pragma unroll
          for (int u = 0; u < CB; ++u) {
            const int ci = cb + u < ncoord ? cb + u : (ncoord > 0 ? ncoord - 1 : 0);      // (clamped: a row past the end is multiplied by p = 0 below)
            const uint4* src = reinterpret_cast<const uint4*>(F.A32T + (c0 + (size_t)ci) * F.n + j);
            a0[u] = src[0]; a1[u] = src[1];
          }
#pragma unroll
          for (int u = 0; u < CB; ++u) {
            const bool in = cb + u < ncoord;
            const long long pa = in ? (long long)spv[(cb + u) * bc] : 0ll;
            const long long pb = (in && bc > 1) ? (long long)spv[(cb + u) * bc + 1] : 0ll;
            const uint32_t av[8] = {a0[u].x, a0[u].y, a0[u].z, a0[u].w, a1[u].x, a1[u].y, a1[u].z, a1[u].w};
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) {
              sum[0][w8] += (long long)av[w8] * pa;
              if (bc > 1) sum[1][w8] += (long long)av[w8] * pb;
            }
          }
        }
        for (int b = 0; b < bc && b < 2; ++b)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            long long r;
            if (pow2) r = (long long)((uint64_t)sum[b][u] & (F.q - 1));
            else { r = sum[b][u] % (long long)F.q; if (r < 0) r += (long long)F.q; }
            mypart[((j + (size_t)u) * (size_t)bc + (size_t)b) * (size_t)ntask] = (uint64_t)r;
          }
      }
    }
}

// task (descending length) -> (tile group, column group); column groups of one tile group are neighbours in the order
template <int RT, int NB, int PD, int HALF, int CD>
__device__ __forceinline__ void trmm_stream_body(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X,
                                                 const StreamGeom& g, size_t nkb, size_t ldx, size_t row_hi) {
  static_assert((PD - 1) * (RT + NB) <= 63, "vmcnt is a 6-bit counter");
  static_assert(CD != 2 || NB == 1, "the dense normals stream holds one fragment");
  static_assert(8 % RT == 0 && 8 % NB == 0, "a tile group stays inside one row block, a column group inside one column block");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = HALF * (int)blockIdx.x + (wave % HALF);           // waves [0, HALF): the long tasks, [HALF, 2 HALF): their mirrors
  const int mirror = g.ntask - 1 - slot;
  const int task = wave < HALF ? slot : mirror;
  if (wave < HALF ? slot > mirror : mirror <= slot) return;          // the middle of an odd count belongs to the long half (no barrier in here)
  const int tg = g.ntile - 1 - task / g.ncg, cg = task % g.ncg;
  const int t0 = tg * RT;                                            // first 16-row tile of the group
  const int nsteps = 4 * (t0 + RT);                                  // k-steps to the diagonal of the group's last tile
  const int bi = t0 >> 3, tl = t0 & 7;
  const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + (size_t)tl * 64;                 // wave-uniform
  const int cf0 = cg * NB;                                           // first column fragment (16 preimages each)
  // fragments of one column block (8 per 128 preimages) are 64 doubles apart, blocks nkb chunks apart; NB divides 8: a group stays inside its block
  // CD: the COMPACT normals stream of small batches, [k-step][column fragment][lane] over the g.ncg * NB fragments in use only -- consecutive k-steps of
  // a fragment are ncf * 512 B apart instead of 4 KiB (where every wave of the chip reads the same two L2 channels: tools/probe_stream.hip)
  // CD == 2 (one fragment, NB = 1, ncg = 1): the DENSE stream of a call with bc = 1, 2, 4, 8 or 16 preimages, [k-step][preimage < bc][k % 4] -- 32 bc
  // bytes per k-step instead of 512: with one preimage the whole stream is 0.25 MB, resident in L1 / L2, and the normals kernel draws m_pad x bc
  // positions instead of m_pad x 16.  (The product itself did not get faster: it sits at the ~10 B / clock / CU an HBM stream delivers, profiles/r04_notes.md.)
  // Lanes whose column holds no preimage load a neighbour's value and replace it by zero before the MFMA.
  const int bc = CD == 2 ? g.bc : 16;
  const size_t strideB = CD == 2 ? (size_t)4 * bc : CD ? (size_t)g.ncg * NB * 64 : 512;       // doubles per k-step
  const double* gB = CD == 2 ? Dt : CD ? Dt + (size_t)cf0 * 64 : Dt + (size_t)(cf0 >> 3) * nkb * TR_CHUNK + (size_t)(cf0 & 7) * 64;
  const bool bact = CD != 2 || (lane & 15) < bc;
  const int colB = (lane & 15) < bc ? (lane & 15) : bc - 1;           // lanes without a preimage read a neighbour's value and discard it (no exec-masked asm)
  const uint32_t laneB = CD == 2 ? (uint32_t)((colB * 4 + (lane >> 4)) * 8) : (uint32_t)lane * 8u;
  // One k-step of a tile is 4 KiB further down the stream (8 tiles x 512 B), beyond the 12-bit immediate: ring slot u carries its own lane offset
  // (lane * 8 + u * 4096) and the two wave-uniform bases advance once per round of PD steps -- a k-step costs the wave its wait, its MFMAs and its
  // loads, nothing else.  (Measured with per-step scalar address arithmetic and guards: the LONGEST chain's issue time, not HBM, set the launch time.)
  // Loads run up to 2 PD k-steps past the diagonal instead of being clamped: those bytes exist (the rest of the diagonal chunk, the next row block, or
  // the slack the host allocates behind both streams) and are never consumed; every round issues the same number of loads, so the wait count is exact.
  uint32_t voff[PD], voffB[PD];
#pragma unroll
  for (int u = 0; u < PD; ++u) { voff[u] = (uint32_t)lane * 8u + (uint32_t)u * 4096u; voffB[u] = CD ? laneB + (uint32_t)u * (uint32_t)strideB * 8u : voff[u]; }

  d4 acc[RT][NB];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  double a[PD][RT], b[PD][NB];
  const double* pa = gA;
  const double* pb = gB;
  auto issue = [&](double (&av)[RT], double (&bv)[NB], uint32_t vo, uint32_t vob) {
    // NB == 1 <=> at most 16 preimages <=> one column group: every byte of the factor is read once -- non-temporal; with several column groups the neighbour groups want
    // the factor's lines in L2
    ts_for<0, RT>([&](auto I) {
      if constexpr (PSF_TS_NT != 0 && NB == 1) ts_load_nt<decltype(I)::value * 512>(av[decltype(I)::value], vo, pa);
      else ts_load<decltype(I)::value * 512>(av[decltype(I)::value], vo, pa);
    });
    ts_for<0, NB>([&](auto J) { ts_load<decltype(J)::value * 512>(bv[decltype(J)::value], vob, pb); });
  };
  auto consume = [&](double (&av)[RT], double (&bv)[NB]) {
    ts_wait<(PD - 1) * (RT + NB)>(av[0]);                            // all but the PD - 1 newest k-steps have landed
#pragma unroll
    for (int i = 1; i < RT; ++i) ts_touch(av[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) ts_touch(bv[j]);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], (CD == 2 && !bact) ? 0.0 : bv[j], acc[i][j], 0, 0, 0);
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) issue(a[u], b[u], voff[u], voffB[u]);
  const int nfull = nsteps / PD;
  for (int r = 0; r < nfull; ++r) {
    pa += (size_t)PD * 512;
    pb += (size_t)PD * strideB;
#pragma unroll
    for (int u = 0; u < PD; ++u) { consume(a[u], b[u]); issue(a[u], b[u], voff[u], voffB[u]); }
  }
  {  // the last, partial round (nsteps is a multiple of 4, PD need not divide it); its refills only keep the count
    const int rest = nsteps - nfull * PD;
    pa += (size_t)PD * 512;
    pb += (size_t)PD * strideB;
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (u < rest) { consume(a[u], b[u]); issue(a[u], b[u], voff[u], voffB[u]); }      // wave-uniform
    }
  }
  // the unconsumed loads keep their registers until they have landed (see k_trmm_f64_reg)
  ts_wait<0>(a[0][0]);
#pragma unroll
  for (int u = 0; u < PD; ++u) {
#pragma unroll
    for (int i = 0; i < RT; ++i) ts_touch(a[u][i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) ts_touch(b[u][j]);
  }
  // C/D map of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t row = (size_t)(t0 + i) * 16 + (lane >> 4) + 4 * r;
        if (row < row_hi) X[row * ldx + (size_t)(cf0 + j) * 16 + (lane & 15)] = acc[i][j][r];
      }
}

template <int RT, int NB, int PD, int HALF = 4, int CD = 0>
__global__ __launch_bounds__(128 * HALF, (HALF + 1) / 2) void k_trmm_stream(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X,
                                                        StreamGeom g, size_t nkb, size_t ldx, size_t row_hi) {
  trmm_stream_body<RT, NB, PD, HALF, CD>(Lt, Dt, X, g, nkb, ldx, row_hi);
}
// ---- 33 ... 1024 preimages: the operands SHARED by the four waves of a 64 x 64 tile through LDS (round 6) ---------------------------------------------------
// k_trmm_stream at 64 and at 128 preimages runs at the same ~19.5 bytes per clock and CU of operand fetches (one 512-byte fragment per MFMA at 64 preimages: 1.28 ms
// where the matrix pipe needs 0.77; 0.75 fragments per MFMA at 128: 1.87 against 1.54): what bounds it is neither HBM nor the matrix pipe but what a CU can pull
// through its L1.  Here four waves (2 x 2, each 2 x 2 MFMA tiles) own 64 rows x 64 preimages and fetch every fragment of the factor and of the normals ONCE per
// workgroup half, by LDS-DMA: 4 KiB per k-step for 16 MFMAs = a quarter of the fetches.  Every accumulator is still one ascending-k chain from +0 (same bits); a
// tile group runs to the diagonal of its LAST 16-row tile, the shorter tiles add the stream's explicit zeros behind their diagonals (exact).
//   ring: NBUF buffers of H k-steps per half; round r: wait for the own pieces of round r (vmcnt), ONE barrier (everybody's pieces of round r have landed, and
//   everybody is done reading round r - 1), refill the buffer of round r - 1 with round r + NBUF - 1, consume.  Loads run NBUF - 1 rounds ahead of the MFMAs and,
//   as in k_trmm_stream, up to that far past the diagonal (bytes that exist and are never consumed).
//   balance: waves 0-3 take the long task `slot`, waves 4-7 its mirror (the two lengths add up to the same for every slot), wave w and w + 4 share a SIMD; the
//   short half leaves when it is done (s_barrier counts the surviving waves only).
constexpr int TSW_H = 4, TSW_NBUF = 4;                                // k-steps per round, rounds in the ring
#ifndef PSF_TSW64_H
#define PSF_TSW64_H 4
#define PSF_TSW64_NBUF 4
#endif
constexpr int TSW64_H = PSF_TSW64_H, TSW64_NBUF = PSF_TSW64_NBUF;      // the 64 x 64 tiles' rounds (A/B builds: -DPSF_TSW64_H=2 -DPSF_TSW64_NBUF=8: same LDS, shorter rounds, deeper ring)
constexpr size_t TSW_LDS = (size_t)2 * TSW_NBUF * TSW_H * 4096;       // 128 KiB: one workgroup per CU
constexpr size_t TSW128_LDS = (size_t)2 * TSW_NBUF * 2 * 6144;            // 96 KiB (halves of eight waves: 6 KiB per k-step, rounds of two k-steps)
constexpr size_t TSW32_LDS = (size_t)2 * TSW_NBUF * TSW_H * 3072;     // 96 KiB (k_trmm_stream_wg32: 3 KiB per k-step)
// one piece (64 lanes x 16 bytes) of a k-step from global memory into LDS at `lds` + lane * 16: wave-uniform base, a per-lane 32-bit offset -- no vector
// arithmetic per load (an FP64 MFMA holds the SIMD's vector pipe; scalar and memory instructions issue beside it)
__device__ __forceinline__ void tsw_dma(uint32_t lds, uint32_t voff, const void* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds), "v"(voff), "s"(base) : "memory");      // (M0 is reserved: the compiler sets it before every use of its own and has none in these kernels)
}
// WCN = 2: halves of four waves, 64 rows x 64 preimages (33 ... 64 preimages); WCN = 4: halves of EIGHT waves (2 x 4), 64 rows x 128 preimages: 6 KiB per k-step for 32
// MFMAs -- 12 bytes per clock and CU at the matrix peak instead of 16, which is what the four-wave tiles needed beside a fetch ceiling of ~19.5 (65 ... 1024 preimages;
// 1024 threads, the waves 6 and 7 of a half bring nothing)
template <int H, int NBUF, int CD, int WCN = 2>
__global__ __launch_bounds__(256 * WCN, 1) void k_trmm_stream_wg(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X,
                                                           StreamGeom g, size_t nkb, size_t ldx, size_t row_hi) {
  static_assert((H == 4 || H == 2) && (NBUF == 4 || NBUF == 8) && (NBUF - 2) * H <= 63, "rounds of four or two k-steps in a ring of four or eight");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NP = 2 + WCN;                                        // 1 KiB pieces per k-step of a half: 2 of the factor (4 fragments) | WCN of the normals (2 WCN fragments)
  constexpr int KS_D = NP * 128;                                     // doubles per k-step of a half
  constexpr int NWH = 2 * WCN;                                       // waves per half
  constexpr int G = H / 2;                                           // k-steps per register group
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wave / NWH, w4 = wave % NWH;
  const bool brings = w4 < NP;                                       // wave-uniform
  // workgroup -> slot: the column groups of one tile group (consecutive slots) stay on one XCD, so that the factor's fragments they all read cross that XCD's L2 once:
  // workgroups are dealt to the XCDs round-robin, XCD x takes the slots [x per, (x + 1) per)
  const int nwg = (g.ntask + 1) / 2, per = (nwg + 7) / 8;
  const int slot = g.ncg > 1 ? (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (slot >= nwg || (g.ncg > 1 && (int)(blockIdx.x >> 3) >= per)) return;
  const int mirror = g.ntask - 1 - slot;
  if (half ? mirror <= slot : slot > mirror) return;                 // the middle of an odd count belongs to the long half
  const int task = half ? mirror : slot;
  const int tg = g.ntile - 1 - task / g.ncg, cg = task % g.ncg;      // g.ntile: groups of four 16-row tiles, g.ncg: groups of four column fragments
  const int t0 = tg * 4;
  const int nround = (t0 + 4) * 4 / H;                               // 4 (t0 + 4) k-steps to the diagonal of the group's last tile; a multiple of NBUF
  const int bi = t0 >> 3, tl = t0 & 7;
  const int cf0 = g.cf_base + cg * 2 * WCN;
  const size_t strideB = CD ? (size_t)(g.ncf ? g.ncf : g.ncg * 2 * WCN) * 64 : 512;      // doubles per k-step of the normals stream (compact: [k-step][fragment][lane] over the fragments in use)
  // piece w4 of a k-step: 0, 1 = the factor's four fragments (2 KiB, contiguous in the chunk stream), 2 ... = the normals' 2 WCN fragments (contiguous)
  const double* src = w4 < 2 ? Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + (size_t)tl * 64 + (size_t)w4 * 128
                             : (CD ? Dt + (size_t)cf0 * 64 : Dt + (size_t)(cf0 >> 3) * nkb * TR_CHUNK + (size_t)(cf0 & 7) * 64) + (size_t)(w4 - 2) * 128;
  const size_t sstride = w4 < 2 ? (size_t)512 : strideB;
  uint32_t voff[H];
#pragma unroll
  for (int u = 0; u < H; ++u) voff[u] = (uint32_t)lane * 16u + (uint32_t)u * (uint32_t)sstride * 8u;
  double* ring = smem + (size_t)half * (NBUF * H * KS_D);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void_ptr)ring + (uint32_t)w4 * 1024u;      // this wave's piece of k-step 0 of buffer 0
  const char* sbase = reinterpret_cast<const char*>(src);            // the wave's piece of the round the next fill brings
  const size_t round_bytes = (size_t)H * sstride * 8;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  const int wr = w4 / WCN, wc = w4 % WCN;
  const double* rdA = ring + (wr * 2) * 64 + lane;                   // fragment wr * 2 of the factor, this lane's element; the next fragment 64 doubles on
  const double* rdB = ring + 256 + (wc * 2) * 64 + lane;
#pragma unroll
  for (int r = 0; r < NBUF; ++r) {
#pragma unroll
    for (int u = 0; u < H; ++u) if (brings) tsw_dma(lds0 + (uint32_t)((r * H + u) * KS_D * 8), voff[u], sbase);
    sbase += round_bytes;
  }
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NBUF - 1) * H) : "memory");      // round 0 is there (a wave that brings nothing passes the wait at once)
  double a0[G][2], b0[G][2], a1[G][2], b1[G][2];                     // the first and the second G k-steps of a round: [k-step][fragment]
  auto mm = [&](double x, double y, int i, int j) { acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i][j], 0, 0, 0); };
#pragma unroll
  for (int u = 0; u < G; ++u) { a0[u][0] = rdA[u * KS_D]; a0[u][1] = rdA[u * KS_D + 64]; b0[u][0] = rdB[u * KS_D]; b0[u][1] = rdB[u * KS_D + 64]; }
  // One round: [a0 / b0 hold the first G k-steps]; the second G are read while the first 4 G MFMAs run; behind them ONE barrier says that round r + 1 has landed for
  // everybody and that nobody reads buffer r any more; the refills of buffer r (round r + NBUF) and the reads of round r + 1 issue between the last 4 G MFMAs
  // (MFMA e of those: the refill of k-step e for e < H, the (e - 2 G)-th read pair of the next round from e = 2 G on).
  auto round = [&](auto BUFC) {
    constexpr int BUF = decltype(BUFC)::value;
    constexpr int cur = BUF * H * KS_D, nxt = ((BUF + 1) % NBUF) * H * KS_D;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      a1[u][0] = rdA[cur + (G + u) * KS_D]; a1[u][1] = rdA[cur + (G + u) * KS_D + 64];
      b1[u][0] = rdB[cur + (G + u) * KS_D]; b1[u][1] = rdB[cur + (G + u) * KS_D + 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; ++u) { mm(a0[u][0], b0[u][0], 0, 0); mm(a0[u][0], b0[u][1], 0, 1); mm(a0[u][1], b0[u][0], 1, 0); mm(a0[u][1], b0[u][1], 1, 1); }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((NBUF - 2) * H) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    ts_for<0, 4 * G>([&](auto EC) {
      constexpr int e = decltype(EC)::value, u = e / 4, i = (e % 4) / 2, j = e % 2;
      mm(a1[u][i], b1[u][j], i, j);
      if constexpr (e < H) {
        if (brings) tsw_dma(lds0 + (uint32_t)((BUF * H + e) * KS_D * 8), voff[e], sbase);
        if constexpr (e == H - 1) sbase += round_bytes;
      }
      if constexpr (e >= 2 * G) {
        constexpr int rp = e - 2 * G, ru = rp / 2;                   // read pair rp: the factor's (even) or the normals' (odd) two fragments of k-step ru
        if constexpr (rp % 2 == 0) { a0[ru][0] = rdA[nxt + ru * KS_D]; a0[ru][1] = rdA[nxt + ru * KS_D + 64]; }
        else { b0[ru][0] = rdB[nxt + ru * KS_D]; b0[ru][1] = rdB[nxt + ru * KS_D + 64]; }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  for (int r = 0; r < nround; r += NBUF) ts_for<0, NBUF>([&](auto BC) { round(BC); });      // (nround is a multiple of NBUF: 4 (tg + 1) rounds of four, 8 (tg + 1) of two k-steps)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // the over-read pieces land before the workgroup's LDS is released
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t row = (size_t)(t0 + wr * 2 + i) * 16 + (lane >> 4) + 4 * r;
        if (row < row_hi) X[row * ldx + (size_t)(cf0 + wc * 2 + j) * 16 + (lane & 15)] = acc[i][j][r];
      }
}

// 17 ... 32 preimages: the same ring for tiles of 64 rows x 32 preimages -- wave w of a half owns the 16-row tile w and both column fragments (two MFMAs per
// k-step), a k-step is 2 KiB of the factor (pieces 0, 1: waves 0, 1) and 1 KiB of the normals (piece 2: wave 2; wave 3 brings nothing).  Here the launch is bound by
// reading the factor once (0.85 ms with the one-wave tasks, whose operand fetches are three fragments per two MFMAs).
template <int H, int NBUF, int CD>
__global__ __launch_bounds__(512, 1) void k_trmm_stream_wg32(const double* __restrict__ Lt, const double* __restrict__ Dt, double* __restrict__ X,
                                                             StreamGeom g, size_t nkb, size_t ldx, size_t row_hi) {
  static_assert(H == 4 && NBUF == 4, "rounds of four k-steps in a ring of four");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS_D = 384;                                          // doubles per k-step of a half: 4 fragments of the factor | 2 of the normals
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wave >> 2, w4 = wave & 3;
  const int slot = (int)blockIdx.x, mirror = g.ntask - 1 - slot;
  if (half ? mirror <= slot : slot > mirror) return;                 // the middle of an odd count belongs to the long half
  const int task = half ? mirror : slot;
  const int tg = g.ntile - 1 - task / g.ncg, cg = task % g.ncg;      // g.ntile: groups of four 16-row tiles, g.ncg: groups of TWO column fragments
  const int t0 = tg * 4;
  const int nround = (t0 + 4) * 4 / H;
  const int bi = t0 >> 3, tl = t0 & 7;
  const int cf0 = g.cf_base + cg * 2;
  const size_t strideB = CD ? (size_t)(g.ncf ? g.ncf : g.ncg * 2) * 64 : 512;
  const double* src = w4 < 2 ? Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + (size_t)tl * 64 + (size_t)w4 * 128
                             : (CD ? Dt + (size_t)cf0 * 64 : Dt + (size_t)(cf0 >> 3) * nkb * TR_CHUNK + (size_t)(cf0 & 7) * 64);
  const size_t sstride = w4 < 2 ? (size_t)512 : strideB;
  uint32_t voff[H];
#pragma unroll
  for (int u = 0; u < H; ++u) voff[u] = (uint32_t)lane * 16u + (uint32_t)u * (uint32_t)sstride * 8u;
  double* ring = smem + (size_t)half * (NBUF * H * KS_D);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void_ptr)ring + (uint32_t)w4 * 1024u;
  const char* sbase = reinterpret_cast<const char*>(src);
  const size_t round_bytes = (size_t)H * sstride * 8;
  const bool brings = w4 < 3;                                        // wave-uniform
  d4 acc[2];
  acc[0] = d4{0.0, 0.0, 0.0, 0.0}; acc[1] = d4{0.0, 0.0, 0.0, 0.0};
  const double* rdA = ring + w4 * 64 + lane;
  const double* rdB = ring + 256 + lane;
#pragma unroll
  for (int r = 0; r < NBUF; ++r) {
#pragma unroll
    for (int u = 0; u < H; ++u) if (brings) tsw_dma(lds0 + (uint32_t)((r * H + u) * KS_D * 8), voff[u], sbase);
    sbase += round_bytes;
  }
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NBUF - 1) * H) : "memory");      // round 0 is there (a wave without loads passes the wait at once)
  // the round of k_trmm_stream_wg with two MFMAs per k-step: k-steps 2, 3 are read while the first four MFMAs run, ONE barrier behind them, then each of the last four
  // MFMAs is followed by one refill of buffer r (round r + NBUF) and one read of round r + 1
  double a0[2], b0[2][2], a1[2], b1[2][2];
  auto mm = [&](double x, double y, int j) { acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[j], 0, 0, 0); };
#pragma unroll
  for (int u = 0; u < 2; ++u) { a0[u] = rdA[u * KS_D]; b0[u][0] = rdB[u * KS_D]; b0[u][1] = rdB[u * KS_D + 64]; }
  auto round = [&](auto BUFC) {
    constexpr int BUF = decltype(BUFC)::value;
    constexpr int cur = BUF * H * KS_D, nxt = ((BUF + 1) % NBUF) * H * KS_D;
#pragma unroll
    for (int u = 0; u < 2; ++u) { a1[u] = rdA[cur + (2 + u) * KS_D]; b1[u][0] = rdB[cur + (2 + u) * KS_D]; b1[u][1] = rdB[cur + (2 + u) * KS_D + 64]; }
    __builtin_amdgcn_sched_barrier(0);
    mm(a0[0], b0[0][0], 0); mm(a0[0], b0[0][1], 1); mm(a0[1], b0[1][0], 0); mm(a0[1], b0[1][1], 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((NBUF - 2) * H) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    ts_for<0, 4>([&](auto EC) {
      constexpr int e = decltype(EC)::value, u = e / 2, j = e % 2;
      mm(a1[u], b1[u][j], j);
      if (brings) tsw_dma(lds0 + (uint32_t)((BUF * H + e) * KS_D * 8), voff[e], sbase);
      if constexpr (e == 3) sbase += round_bytes;
      if constexpr (j == 0) a0[u] = rdA[nxt + u * KS_D];
      else { b0[u][0] = rdB[nxt + u * KS_D]; b0[u][1] = rdB[nxt + u * KS_D + 64]; }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  for (int r = 0; r < nround; r += NBUF) {
    round(std::integral_constant<int, 0>{});
    round(std::integral_constant<int, 1>{});
    round(std::integral_constant<int, 2>{});
    round(std::integral_constant<int, 3>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = (size_t)(t0 + w4) * 16 + (lane >> 4) + 4 * r;
      if (row < row_hi) X[row * ldx + (size_t)(cf0 + j) * 16 + (lane & 15)] = acc[j][r];
    }
}

// rounding + syndrome shares of one or two preimages in ONE launch behind the product: one wave per RT 16-row tiles of x
template <int RT>
__global__ __launch_bounds__(256) void k_round_syndrome_small(const double* __restrict__ X, size_t ldx, size_t row_hi, StreamGeom g, StreamFuse fz) {
  __shared__ double s_x[4][RT * 16 * 2];
  __shared__ int32_t s_p[4][RT * 16 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int task = (int)blockIdx.x * 4 + wave;
  if (task >= g.ntask) return;                                       // (wave-uniform; no barrier in here)
  const int tg = task, t0 = tg * RT, bc = g.bc;
  double* sx = s_x[wave];
  for (int e = lane; e < RT * 16 * bc; e += 64) {
    const size_t row = (size_t)t0 * 16 + (size_t)(e / bc);
    sx[e] = row < row_hi ? X[row * ldx + (size_t)(e % bc)] : 0.0;
  }
  fused_tail_epilogue<RT>(fz, g.ntask, bc, t0, tg, sx, s_p[wave], lane);
}
// A32T[coordinate][row] = A[row][coordinate] (the compact copy the fused tail reads): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_transpose_A32(const uint64_t* __restrict__ A, size_t n, size_t m, uint32_t* __restrict__ A32T) {
  __shared__ uint32_t tile[32][33];
  const size_t i0 = (size_t)blockIdx.y * 32, j0 = (size_t)blockIdx.x * 32;      // rows of A, coordinates
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = (i0 + r < n && j0 + tx < m) ? (uint32_t)A[(i0 + r) * m + j0 + tx] : 0u;
  __syncthreads();
  for (int r = ty; r < 32; r += 8) if (j0 + r < m && i0 + tx < n) A32T[(j0 + r) * n + i0 + tx] = tile[tx][r];
}

// ---- gadget nearest plane for a SINGLE CALL (mp_perturbation.rs:173-191; gadget_classical.rs:169-229 for the digits) ------------------------------
// k_gadget_queue keeps 64 lanes busy with 128 problems per wave; a problem is a chain of k dependent draws, so with the n problems of one preimage
// (psf.rs:48-80) the launch lasts one chain of ~3.4 k sampling iterations whatever the wave count (0.2 ms at C3).  Here ONE WAVE OWNS ONE PROBLEM:
// lane r holds c_r, the 64 lanes evaluate attempts t0 .. t0 + 63 of the current draw at once (exact rule, sz_attempt) and the lowest accepted attempt
// is taken -- the first accepted attempt of the draw's own Philox stream, i.e. the value every other sampler of the library returns.
// A draw costs one round (64 attempts miss with probability (11/12)^64 = 0.4 %) and the chain k of them.  Every lane pays an exact attempt, so this
// form only pays while there are about as many problems as SIMDs; the host uses it up to n B = 2048 (psfp.hip; k_gadget_quad beyond).
// Checks mirror k_gadget_queue (|z_i| <= 16000, c in int16): the failure flag is raised by the same inputs.
// one problem (row j of v, preimage index `index`, value v = v_j mod q) on ONE WAVE; out(row r, z_r) is called by lane r < k.  Returns the failure flag.
template <class Out>
__device__ __forceinline__ int gadget_wave_problem(uint64_t seed, uint64_t index, uint32_t j, uint32_t k, uint64_t base, uint64_t v, const GadgetTablesQ& tb, Out&& out) {
  const int lane = threadIdx.x & 63;
  int f = 0;
  int c = 0;
  {  // digit `lane` of v_j (find_solution_gadget_vec): c = -x
    uint64_t d = 0;
    if (base == 2) d = lane < 64 ? (v >> lane) & 1 : 0;
    else for (int t = 0; t <= lane && t < (int)k; ++t) { d = v % base; v = (v - d) / base; }
    if (lane < (int)k) c = -(int)d;
  }
  const uint32_t tw = tag_word(TAG_GADGET, index), idx_lo = (uint32_t)index;
  // Nothing the chain waits for comes from memory: lane i holds the per-step scalars of step i (norm, SampleZ parameters, row ranges), read by
  // v_readlane when step i runs; column i of S_k and of the Gram-Schmidt matrix (lane r: row r) is loaded one step ahead.
  const int li = lane < (int)k ? lane : 0;
  const double my_norm2 = tb.norm2[li];
  const SampleZParams my_sz = tb.sz[li];
  const int my_glo = tb.rng[li], my_ghi = tb.rng[k + li];
  auto bcast_d = [&](double x, int src) -> double {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(__double2loint(x), src), hi = (uint32_t)__builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double((int)hi, (int)lo);
  };
  auto bcast_ll = [&](long long x, int src) -> long long {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), src);
    return (long long)(((uint64_t)hi << 32) | lo);
  };
  double gcol = tb.gso[(size_t)li * k + (k - 1)];
  int skcol = tb.Sk[(size_t)li * k + (k - 1)];
  for (int i = (int)k - 1; i >= 0; --i) {
    const double g_now = gcol;
    const int sk_now = skcol;
    if (i > 0) { gcol = tb.gso[(size_t)li * k + (i - 1)]; skcol = tb.Sk[(size_t)li * k + (i - 1)]; }      // in flight during this step
    // centre <c, b~_i> / ||b~_i||^2 over the non-zero rows: ONE ascending fma chain, evaluated by every lane alike
    double dot = 0.0;
    const int ghi = __builtin_amdgcn_readlane(my_ghi, i);
    for (int r = __builtin_amdgcn_readlane(my_glo, i); r <= ghi; ++r) dot = fma((double)__builtin_amdgcn_readlane(c, r), bcast_d(g_now, r), dot);
    const double cen = dot / bcast_d(my_norm2, i);
    SampleZParams sp;
    sp.inv_s = bcast_d(my_sz.inv_s, i); sp.c6 = bcast_ll(my_sz.c6, i); sp.f6 = bcast_ll(my_sz.f6, i);
    sp.n_int = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.n_int, i); sp.thr_int = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.thr_int, i);
    sp.thr_frac = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.thr_frac, i); sp.sh = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.sh, i);
    const SzRange rg = sz_range(cen, sp);
    const uint32_t coord = j * k + (uint32_t)i;
    long long x = 0;
    bool found = false;
    for (uint32_t t0 = 0; t0 < kMaxAttempts && !found; t0 += 64) {
      const uint32_t t = t0 + (uint32_t)lane;
      uint32_t wa, wb;
      sz_attempt_words(seed, coord, idx_lo, tw, t, rg.sh, &wa, &wb);
      long long xl = 0;
      const bool acc = sz_attempt(seed, coord, idx_lo, tw, t, wa, wb, rg, cen, sp.inv_s, &xl);
      const uint64_t mask = __ballot(acc);
      if (mask) { x = bcast_ll(xl, __builtin_ctzll(mask)); found = true; }
    }
    if (!found) { f = 1; x = (long long)floor(cen + 0.5); }
    if (x > 16000 || x < -16000) f = 1;
    if (lane < (int)k) {
      const int nv = c - (int)x * sk_now;
      if (nv > 32767 || nv < -32768) f = 1;
      c = nv;
    }
  }
  if (lane < (int)k) out(lane, -c);
  return f;
}

__global__ __launch_bounds__(256) void k_gadget_wave(uint64_t seed, uint64_t first_index, uint32_t n, uint32_t k, uint64_t q, uint64_t base, size_t B, size_t ld,
                                                     const uint64_t* __restrict__ V, GadgetTablesQ tb, int8_t* __restrict__ Zlo, int8_t* __restrict__ Zhi,
                                                     int* __restrict__ fail) {
  const size_t pid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pid >= (size_t)n * B) return;                             // wave-uniform; no barrier in this kernel
  const uint32_t j = (uint32_t)(pid / B);
  const size_t b = pid % B;
  int anyhi = 0;
  const int f = gadget_wave_problem(seed, first_index + b, j, k, base, V[(size_t)j * ld + b] % q, tb, [&](int r, int zz) {
    const int32_t zl = (int32_t)(int8_t)(zz & 0xff);
    const int32_t zh = (zz - zl) >> 8;
    const size_t cc = (size_t)j * k + (size_t)r;
    const size_t addr = ((cc >> 4) * ld + b) * 16 + (cc & 15);
    Zlo[addr] = (int8_t)zl;
    Zhi[addr] = (int8_t)zh;
    if (zh) anyhi = 1;
  });
  if (f) atomicOr(fail, 1);
  if (anyhi) atomicOr(fail + 1, 1);
}

#ifdef PSF_EXPERIMENTS   /* round 4's sixteen-lane kernel with exact attempts: replaced by k_gadget_quad<., 16> (round 6); comparison arm of the experiments build */
// ---- the same walk with SIXTEEN lanes per problem (four problems per wave): a handful to a few hundred preimages ------------------------------------
// k_gadget_wave spends a whole wave's instruction stream on one problem; from ~1 k problems on that stream is what bounds the launch.  Here a DPP row
// (16 lanes) owns a problem: lane n of the row holds c_r for r = n, n + 16, n + 32, n + 48; the 16 lanes evaluate attempts t0 .. t0 + 15 of the row's
// current draw (exact rule), a draw takes 1.33 rounds on average (attempts miss with probability (11/12)^16), and the four rows of a wave share every
// instruction, the ascending centre chain included -- its terms come from v_mov_dpp row_share (lane r % 16 of each row to the whole row), the
// coefficients <-, b~_i> from v_readlane.  The DPP control is an immediate, so the chain is unrolled over the 64 possible rows behind wave-uniform
// guards on the support of b~_i.  Same checks, same values as k_gadget_wave / k_gadget_queue.
template <int N> __device__ __forceinline__ int ts_row_share(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, false); }

__global__ __launch_bounds__(256) void k_gadget_wave16(uint64_t seed, uint64_t first_index, uint32_t n, uint32_t k, uint64_t q, uint64_t base, size_t B, size_t ld,
                                                       const uint64_t* __restrict__ V, GadgetTablesQ tb, int8_t* __restrict__ Zlo, int8_t* __restrict__ Zhi,
                                                       int* __restrict__ fail) {
  const int lane = threadIdx.x & 63, row16 = lane >> 4, ln = lane & 15;
  const size_t total = (size_t)n * B;
  const size_t pid0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (pid0 >= total) return;                                    // wave-uniform; no barrier in this kernel
  const size_t pid = pid0 + (size_t)row16;
  const bool active = pid < total;
  const uint32_t j = active ? (uint32_t)(pid / B) : 0;
  const size_t b = active ? pid % B : 0;
  int f = 0;
  int c[4] = {0, 0, 0, 0};
  {  // digits of v_j: row r = slot * 16 + ln
    const uint64_t v0 = V[(size_t)j * ld + b] % q;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int r = sl * 16 + ln;
      uint64_t v = v0, d = 0;
      if (base == 2) d = (v >> r) & 1;
      else for (int t = 0; t <= r && t < (int)k; ++t) { d = v % base; v = (v - d) / base; }
      if (r < (int)k) c[sl] = -(int)d;
    }
  }
  const uint64_t index = first_index + b;
  const uint32_t tw = tag_word(TAG_GADGET, index), idx_lo = (uint32_t)index;
  // per-step scalars and the coefficient column: held wave-wide, lane L <-> step / row L (k <= 64), read by v_readlane
  const int li = lane < (int)k ? lane : 0;
  const double my_norm2 = tb.norm2[li];
  const SampleZParams my_sz = tb.sz[li];
  const int my_glo = tb.rng[li], my_ghi = tb.rng[k + li];
  auto bcast_d = [&](double x, int src) -> double {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(__double2loint(x), src), hi = (uint32_t)__builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double((int)hi, (int)lo);
  };
  auto bcast_ll = [&](long long x, int src) -> long long {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), src);
    return (long long)(((uint64_t)hi << 32) | lo);
  };
  double gcol = tb.gso[(size_t)li * k + (k - 1)];
  int skc[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) { const int r = sl * 16 + ln; skc[sl] = r < (int)k ? tb.Sk[(size_t)r * k + (k - 1)] : 0; }
  for (int i = (int)k - 1; i >= 0; --i) {
    const double g_now = gcol;
    int sk_now[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) sk_now[sl] = skc[sl];
    if (i > 0) {                                                 // in flight during this step
      gcol = tb.gso[(size_t)li * k + (i - 1)];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) { const int r = sl * 16 + ln; skc[sl] = r < (int)k ? tb.Sk[(size_t)r * k + (i - 1)] : 0; }
    }
    // centre <c, b~_i> / ||b~_i||^2: ONE ascending fma chain over the rows 0 .. k - 1 (zeros outside the support of b~_i), per DPP row its own c
    double dot = 0.0;
    const int glo = __builtin_amdgcn_readlane(my_glo, i), ghi = __builtin_amdgcn_readlane(my_ghi, i);      // support of b~_i (zeros outside: skipped, exact)
    ts_for<0, 64>([&](auto R) {
      constexpr int r = decltype(R)::value;
      if (r >= glo && r <= ghi) dot = fma((double)ts_row_share<r % 16>(c[r / 16]), bcast_d(g_now, r), dot);      // (wave-uniform guard)
    });
    const double cen = dot / bcast_d(my_norm2, i);
    SampleZParams sp;
    sp.inv_s = bcast_d(my_sz.inv_s, i); sp.c6 = bcast_ll(my_sz.c6, i); sp.f6 = bcast_ll(my_sz.f6, i);
    sp.n_int = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.n_int, i); sp.thr_int = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.thr_int, i);
    sp.thr_frac = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.thr_frac, i); sp.sh = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.sh, i);
    const SzRange rg = sz_range(cen, sp);
    const uint32_t coord = j * k + (uint32_t)i;
    long long x = 0;
    bool found = !active;
    for (uint32_t t0 = 0; t0 < kMaxAttempts; t0 += 16) {           // t0 advances for every row alike; a row that has found its draw idles
      if (!__ballot(!found)) break;
      const uint32_t t = t0 + (uint32_t)ln;
      long long xl = 0;
      bool acc = false;
      if (!found) {
        uint32_t wa, wb;
        sz_attempt_words(seed, coord, idx_lo, tw, t, rg.sh, &wa, &wb);
        acc = sz_attempt(seed, coord, idx_lo, tw, t, wa, wb, rg, cen, sp.inv_s, &xl);
      }
      const uint32_t gm = (uint32_t)(__ballot(acc) >> (16 * row16)) & 0xffffu;
      const int fl = gm ? (16 * row16 + __builtin_ctz(gm)) : lane;
      const long long xs = __shfl(xl, fl);
      if (!found && gm) { x = xs; found = true; }
    }
    if (!found) { f = 1; x = (long long)floor(cen + 0.5); }
    if (active && (x > 16000 || x < -16000)) f = 1;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int nv = c[sl] - (int)x * sk_now[sl];
      if (active && sl * 16 + ln < (int)k && (nv > 32767 || nv < -32768)) f = 1;
      c[sl] = nv;
    }
  }
  int anyhi = 0;
  if (active) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int r = sl * 16 + ln;
      if (r >= (int)k) continue;
      const int32_t zz = -c[sl];
      const int32_t zl = (int32_t)(int8_t)(zz & 0xff);
      const int32_t zh = (zz - zl) >> 8;
      const size_t cc = (size_t)j * k + (size_t)r;
      const size_t addr = ((cc >> 4) * ld + b) * 16 + (cc & 15);
      Zlo[addr] = (int8_t)zl;
      Zhi[addr] = (int8_t)zh;
      if (zh) anyhi = 1;
    }
  }
  if (f) atomicOr(fail, 1);
  if (anyhi) atomicOr(fail + 1, 1);
}
#endif

// ---- the same walk with FOUR lanes per problem (sixteen problems per wave): tens to a few hundred preimages (round 6) ---------------------------------------------
// k_gadget_wave16 spends 2 600 issue cycles per step on four problems (sixteen exact attempts per problem and round, the centre chain repeated by sixteen lanes):
// from ~8 k problems on the instruction stream bounds it (0.32 ms at 64 preimages of C3), and the queue kernel, whose lanes never wait, has a floor of 0.23 ms however
// few the problems (a step is three trips through its LDS queues and a projection with dependent LDS reads).  Here a QUAD owns a problem: lane s of the quad holds
// c_r for r = s, s + 4, ...; it evaluates the attempt groups s, s + 4, ... of the current draw (sz_group4_narrow: one Philox block and four fp32 screens for four
// attempts, the exact decision only inside the 0.1 % band), sixteen attempts per problem and round as before, the lowest accepting group wins = the first accepted
// attempt of the draw's own Philox stream.  The centre chain takes its terms by v_mov_dpp quad_perm, the coefficients by v_readlane (lane L <-> row L), unrolled
// behind wave-uniform guards on the support of b~_i.  Same checks, same values as the other three kernels.  KT = ceil(k / 4) rounded up to 8 or 16.
// LPP = 4 (a quad per problem; 16 attempts per problem and round) or 16 (a DPP row per problem, 64 attempts per round: one round per draw, for the few thousand problems
// of 8 ... 16 preimages, where the launch is one chain long and a round of a quad serves sixteen problems of which the slowest decides)
template <int LPP, int S> __device__ __forceinline__ int ts_group_share(int v) {
  if constexpr (LPP == 4) return __builtin_amdgcn_update_dpp(0, v, S * 0x55, 0xf, 0xf, false);      // quad_perm: [S, S, S, S]
  else return __builtin_amdgcn_update_dpp(0, v, 0x150 + S, 0xf, 0xf, false);                        // row_share: S
}

template <int KT, int LPP = 4>
__global__ __launch_bounds__(256) void k_gadget_quad(uint64_t seed, uint64_t first_index, uint32_t n, uint32_t k, uint64_t q, uint64_t base, size_t B, size_t ld,
                                                     const uint64_t* __restrict__ V, GadgetTablesQ tb, int8_t* __restrict__ Zlo, int8_t* __restrict__ Zhi,
                                                     int* __restrict__ fail) {
  constexpr int PPW = 64 / LPP;                                    // problems per wave
  const int lane = threadIdx.x & 63, quad0 = lane & ~(LPP - 1), sub = lane & (LPP - 1);
  const size_t total = (size_t)n * B;
  const size_t pid0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW;
  if (pid0 >= total) return;                                    // wave-uniform; no barrier in this kernel
  const size_t pid = pid0 + (size_t)(lane / LPP);
  const bool active = pid < total;
  const uint32_t j = active ? (uint32_t)(pid / B) : 0;
  const size_t b = active ? pid % B : 0;
  int f = 0;
  int c[KT];
  {  // digits of v_j: row r = 4 t + sub
    const uint64_t v0 = V[(size_t)j * ld + b] % q;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int r = LPP * t + sub;
      uint64_t v = v0, d = 0;
      if (base == 2) d = (v >> r) & 1;
      else for (int u = 0; u <= r && u < (int)k; ++u) { d = v % base; v = (v - d) / base; }
      c[t] = r < (int)k ? -(int)d : 0;
    }
  }
  const uint64_t index = first_index + b;
  const uint32_t tw = tag_word(TAG_GADGET, index), idx_lo = (uint32_t)index;
  // per-step scalars and the coefficient column: held wave-wide, lane L <-> step / row L (k <= 64), read by v_readlane
  const int li = lane < (int)k ? lane : 0;
  const double my_norm2 = tb.norm2[li];
  const SampleZParams my_sz = tb.sz[li];
  const int my_glo = tb.rng[li], my_ghi = tb.rng[k + li];
  auto bcast_d = [&](double x, int src) -> double {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(__double2loint(x), src), hi = (uint32_t)__builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double((int)hi, (int)lo);
  };
  auto bcast_ll = [&](long long x, int src) -> long long {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), src);
    return (long long)(((uint64_t)hi << 32) | lo);
  };
  double gcol = tb.gso[(size_t)li * k + (k - 1)];
  int skc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) { const int r = LPP * t + sub; skc[t] = r < (int)k ? tb.Sk[(size_t)r * k + (k - 1)] : 0; }
  for (int i = (int)k - 1; i >= 0; --i) {
    const double g_now = gcol;
    int sk_now[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) sk_now[t] = skc[t];
    if (i > 0) {                                                 // in flight during this step
      gcol = tb.gso[(size_t)li * k + (i - 1)];
#pragma unroll
      for (int t = 0; t < KT; ++t) { const int r = LPP * t + sub; skc[t] = r < (int)k ? tb.Sk[(size_t)r * k + (i - 1)] : 0; }
    }
    // centre <c, b~_i> / ||b~_i||^2: ONE ascending fma chain over the support of b~_i (zeros outside: skipped, exact), per quad its own c
    double dot = 0.0;
    const int glo = __builtin_amdgcn_readlane(my_glo, i), ghi = __builtin_amdgcn_readlane(my_ghi, i);
    ts_for<0, LPP * KT>([&](auto R) {
      constexpr int r = decltype(R)::value;
      if (r >= glo && r <= ghi) dot = fma((double)ts_group_share<LPP, r % LPP>(c[r / LPP]), bcast_d(g_now, r), dot);      // (wave-uniform guard)
    });
    const double cen = dot / bcast_d(my_norm2, i);
    SampleZParams sp;
    sp.inv_s = bcast_d(my_sz.inv_s, i); sp.c6 = bcast_ll(my_sz.c6, i); sp.f6 = bcast_ll(my_sz.f6, i);
    sp.n_int = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.n_int, i); sp.thr_int = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.thr_int, i);
    sp.thr_frac = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.thr_frac, i); sp.sh = (uint32_t)__builtin_amdgcn_readlane((int)my_sz.sh, i);
    const SzRange rg = sz_range(cen, sp);
    const bool narrow = sp.sh == 16 && fabs(cen) < 0x1.0p30;
    const float c_rel = (float)((double)rg.lo - cen), inv_s_f = (float)sp.inv_s;
    const uint32_t coord = j * k + (uint32_t)i;
    long long x = 0;
    bool found = !active;
    for (uint32_t t = (uint32_t)sub; ; t += LPP) {                 // a round: the groups 4 R .. 4 R + 3 of every quad's draw; a quad that has found its draw idles
      if (!__builtin_amdgcn_ballot_w64(!found)) break;
      bool acc1 = false;
      long long xl = 0;
      if (!found) {
        if (t >= kMaxAttempts / 4) { acc1 = true; f = 1; xl = (long long)floor(cen + 0.5); }      // (every lane of the quad gets here in the same round: the lowest takes it)
        else acc1 = narrow ? sz_group4_narrow(seed, coord, idx_lo, tw, t, rg, cen, sp.inv_s, c_rel, inv_s_f, &xl)
                           : sz_group4(seed, coord, idx_lo, tw, t, rg, cen, sp.inv_s, &xl);
      }
      const uint32_t qm = (uint32_t)(__builtin_amdgcn_ballot_w64(acc1) >> quad0) & ((1u << LPP) - 1u);
      const int src = qm ? quad0 + __builtin_ctz(qm) : lane;
      const long long xs = __shfl(xl, src);
      if (!found && qm) { x = xs; found = true; }
    }
    if (active && (x > 16000 || x < -16000)) f = 1;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int nv = c[t] - (int)x * sk_now[t];
      if (active && LPP * t + sub < (int)k && (nv > 32767 || nv < -32768)) f = 1;
      c[t] = nv;
    }
  }
  int anyhi = 0;
  if (active) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int r = LPP * t + sub;
      if (r >= (int)k) continue;
      const int32_t zz = -c[t];
      const int32_t zl = (int32_t)(int8_t)(zz & 0xff);
      const int32_t zh = (zz - zl) >> 8;
      const size_t cc = (size_t)j * k + (size_t)r;
      const size_t addr = ((cc >> 4) * ld + b) * 16 + (cc & 15);
      Zlo[addr] = (int8_t)zl;
      Zhi[addr] = (int8_t)zh;
      if (zh) anyhi = 1;
    }
  }
  if (f) atomicOr(fail, 1);
  if (anyhi) atomicOr(fail + 1, 1);
}

// ---- e_top = p_top + R z for 5 ... 64 preimages (round 6) ---------------------------------------------------------------------------------------------------------
// k_recombine_mfma serves these sizes with its 128 x 128 (preimages x coordinates) tile cut along K over blockIdx.z: half or more of every tile is padding, p comes
// through an LDS transposition in every split and the K ranges meet in 64-bit atomics: 79 us at 16 and 145 us at 64 preimages of C3 (R is 237 MB: 40 us of HBM).
// MEASURED AND NOT KEPT: one wave per 64 coordinates x K range with its operands in a register ring (the form of k_trmm_stream): ~2 000 waves are needed to keep
// enough bytes in flight from registers, i.e. K ranges whose sums meet in atomics again -- and the atomics are what it then costs (4 ps each: 0.11 ms with 4 ranges,
// 0.17 with 16, at 64 preimages).
// k_recombine_wg: a workgroup owns 64 coordinates x 64 preimages over ALL of K; a ring of RW_NBUF slots of K = 128 (R tile 2 x 4 KiB in the bank-rotated layout of
// k_recombine_mfma, z planes 2 x 4 KiB) is filled by LDS-DMA three slots ahead of the MFMAs, so 48 KiB per CU are in flight without a register; Z is the A operand
// (rows = preimages), R the B operand (columns = coordinates: 16 lanes hold 16 consecutive coordinates of one preimage); e = p + sum leaves with plain stores
// (k_recombine_bottom does not have to zero anything); the hi plane of z (some |z| > 127: flags[1], decided by the gadget kernel) is a second pass over the ring.
constexpr size_t RS_SLACK_SLOTS = 4;      // ring slots (K = 128 each) the host allocates behind the digit planes of z
constexpr int RW_NBUF = 4;
constexpr size_t RW_LDS = (size_t)RW_NBUF * 16384;
// NW = 4 waves (one per 16 coordinates, every fragment of preimages) or 8 (33 ... 64 preimages: waves 4-7 take the upper half of the fragments from the same R tiles,
// so that the chain of a slot -- barrier, reads, MFMAs -- is half as long and two waves share a SIMD)
template <int NBF, int NW>
__global__ __launch_bounds__(64 * NW) void k_recombine_wg(const int8_t* __restrict__ R, size_t ldr, size_t mbar, int nk2 /* K / 128 */, const int8_t* __restrict__ Zlo,
                                                          const int8_t* __restrict__ Zhi, size_t ld, const int* __restrict__ flags, const int32_t* __restrict__ P,
                                                          size_t B, int64_t* __restrict__ E, size_t m) {
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char rw_smem[];
  constexpr int FPW = NW == 8 ? (NBF + 1) / 2 : NBF;                 // fragments per wave
  constexpr int DPW = 16 / NW;                                       // DMA pieces per wave and slot
  const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wq = wave & 3, f0 = (wave >> 2) * FPW;                   // coordinates 16 wq ..., fragments f0 ... f0 + FPW - 1 (those below NBF)
  const size_t i0 = (size_t)blockIdx.x * 64;
  {  // blockIdx.y: the column group of 64 preimages (65 ... 128 preimages: two groups, R comes from L2 / the Infinity Cache for the second)
    const size_t bg = (size_t)blockIdx.y * 64;
    Zlo += 16 * bg; Zhi += 16 * bg; P += bg; E += bg * m;
    B = B - bg < 64 ? B - bg : 64;
  }
  const bool use_hi = flags[1] != 0;
  // DMA sources per slot: piece wq of the 4 KiB tiles (R half 0, R half 1, z half 0, z half 1); with eight waves, waves 0-3 bring the first halves, 4-7 the second
  const int pp = wq * 64 + lane;                                     // 16-byte position inside a 4 KiB tile
  const int8_t* srcR = R + (i0 + (size_t)(pp >> 2)) * ldr + (size_t)(((pp & 3) - (pp >> 4)) & 3) * 16;      // LDS slot (row, position) takes k group (position - row / 4) mod 4
  const size_t zoff = ((size_t)wq * ld + (size_t)lane) * 16;         // k group wq of the half, preimage `lane`
  const int h0 = NW == 8 ? wave >> 2 : 0;                            // first half this wave brings
  v4i alo[FPW], ahi[FPW];
#pragma unroll
  for (int f = 0; f < FPW; ++f) { alo[f] = v4i{0, 0, 0, 0}; ahi[f] = v4i{0, 0, 0, 0}; }
  for (int plane = 0; plane < (use_hi ? 2 : 1); ++plane) {
    const int8_t* srcZ = (plane ? Zhi : Zlo) + zoff;
    auto fill = [&](int s) {
      unsigned char* base = rw_smem + (size_t)(s % RW_NBUF) * 16384 + wq * 1024;
#pragma unroll
      for (int d = 0; d < DPW / 2; ++d) {
        const int hh = h0 + d;
        __builtin_amdgcn_global_load_lds(srcR + (size_t)s * 128 + hh * 64, (lds_void_ptr)(base + hh * 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(srcZ + ((size_t)s * 8 + hh * 4) * ld * 16, (lds_void_ptr)(base + 8192 + hh * 4096), 16, 0, 0);
      }
    };
    // (fills past the last slot read the next columns of R / the planes' slack and are never consumed)
#pragma unroll
    for (int s = 0; s < RW_NBUF - 1; ++s) fill(s);
    for (int s = 0; s < nk2; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((RW_NBUF - 2) * DPW) : "memory");
      fill(s + RW_NBUF - 1);
      const unsigned char* sb = rw_smem + (size_t)(s % RW_NBUF) * 16384;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int row = wq * 16 + r16;
        const v4i fr = *reinterpret_cast<const v4i*>(sb + hh * 4096 + (row * 64 + i8_slot(row, g) * 16));
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
          if (f0 + f >= NBF) continue;                               // (wave-uniform: the odd fragment of three)
          const v4i fl = *reinterpret_cast<const v4i*>(sb + 8192 + hh * 4096 + ((g * 64 + (f0 + f) * 16 + r16) * 16));
          if (plane) ahi[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fl, fr, ahi[f], 0, 0, 0);
          else alo[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fl, fr, alo[f], 0, 0, 0);
        }
      }
      asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // the ring is quiet before the next plane (or the end) reuses it
  }
  // C map: column (coordinate) = lane & 15, rows (preimages) = 4 (lane >> 4) + reg
  const size_t ii = i0 + (size_t)wq * 16 + (size_t)r16;
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t bb = (size_t)(f0 + f) * 16 + (size_t)g * 4 + (size_t)r;
      if (f0 + f < NBF && ii < mbar && bb < B) E[bb * m + ii] = (int64_t)P[ii * ld + bb] + (int64_t)alo[f][r] + 256 * (int64_t)ahi[f][r];
    }
}

// ---- the whole samp_p of ONE preimage in ONE workgroup, for small parameter sets (mp_perturbation.rs:304-336) ------------------------------------
// The reference's own benchmarks call samp_p once at n = 8 (m = 121; benches/psf.rs:51-66): nine dependent launches cost more than the arithmetic.
// Here workgroup b does everything for preimage b with LDS between the stages: normals (thread = coordinate), x = sqrt(Sigma_2) d (thread = row, one
// ascending fma chain over the key's chunk stream), p <- D_{Z,r,x} (thread = coordinate), v = u - A p (all threads, partial sums mod q, then a tree),
// the gadget walk (one wave per row of v: gadget_wave_problem), e = p + [R; I] z (thread = row).  Every value is the one the multi-kernel path
// computes (same Philox streams, same chains, exact integers), so the bits are the oracle's.  Limits (host: fused_small_ok): m <= FS_MAX_M, no
// structured factor.  Failure flag as in the stage kernels.
constexpr int FS_MAX_M = 256;
constexpr int FS_THREADS = 512;
__global__ __launch_bounds__(FS_THREADS) void k_samp_p_small(uint64_t seed, uint64_t first_index, uint32_t n, uint32_t k, uint32_t mb, uint64_t q, uint64_t two64,
                                                             uint64_t base, const double* __restrict__ Lt, const uint64_t* __restrict__ A,
                                                             const int8_t* __restrict__ R, size_t ldr, SampleZParams szR, GadgetTablesQ tb,
                                                             const uint64_t* __restrict__ U, int64_t* __restrict__ E, int* __restrict__ fail) {
  __shared__ double s_d[FS_MAX_M];
  __shared__ double s_x[FS_MAX_M];
  __shared__ int32_t s_p[FS_MAX_M];
  __shared__ int32_t s_z[FS_MAX_M];
  __shared__ uint64_t s_part[FS_THREADS];
  const int tid = threadIdx.x;
  const uint32_t w = n * k, m = mb + w;
  const size_t b = blockIdx.x;
  const uint64_t index = first_index + b;
  int f = 0;
  // 1a: d <- N(0,1)^m
  for (uint32_t i = tid; i < m; i += FS_THREADS) s_d[i] = sample_normal(seed, index, i, &f);
  __syncthreads();
  // 1b: x_i = sum_{j <= i} L_ij d_j, ascending from +0 (element (i, j) of the chunk stream: psf_kernels.hpp, tr_chunk_pos)
  for (uint32_t i = tid; i < m; i += FS_THREADS) {
    const double* row = Lt + tr_rowblock_base(i / TR_BM) * TR_CHUNK;
    const int r = (int)(i % TR_BM);
    double acc = 0.0;
    for (uint32_t j = 0; j <= i; ++j) acc = fma(row[(size_t)(j / TR_BK) * TR_CHUNK + tr_chunk_pos(r, (int)(j % TR_BK))], s_d[j], acc);
    s_x[i] = acc;
  }
  __syncthreads();
  // 1c: p_i <- D_{Z, r, x_i}
  for (uint32_t i = tid; i < m; i += FS_THREADS) {
    const double c = s_x[i];
    const SzRange rg = sz_range(c, szR);
    const uint32_t tw = tag_word(TAG_PERTURB, index);
    long long x = 0;
    bool acc = false;
    for (uint32_t g = 0; g < kMaxAttempts / 4 && !acc; ++g) acc = sz_group4(seed, i, (uint32_t)index, tw, g, rg, c, szR.inv_s, &x);
    if (!acc) { f = 1; x = (long long)floor(c + 0.5); }
    if (x > kDigitRangeP || x < -kDigitRangeP) f = 1;
    s_p[i] = (int32_t)x;
  }
  __syncthreads();
  // 2: v_i = u_i - sum_j A_ij p_j mod q: thread t takes row t % n and the columns j = t / n, t / n + FS_THREADS / n, ...; partial sums mod q; tree over the row
  const uint32_t per_row = FS_THREADS / n;                     // n <= FS_MAX_M / k: at least two threads per row
  {
    const uint32_t i = tid % n, slot = tid / n;
    uint64_t s = 0;
    if (slot < per_row)
      for (uint32_t j = slot; j < m; j += per_row) {
        const int32_t pj = s_p[j];
        const uint64_t a = A[(size_t)i * m + j], ap = (uint64_t)(pj < 0 ? -(int64_t)pj : (int64_t)pj);
        uint64_t t = acc128_mod(Acc128{a * ap, (int64_t)__umul64hi(a, ap)}, q, two64);
        if (pj < 0 && t) t = q - t;
        s += t; if (s >= q) s -= q;
      }
    s_part[tid] = s;
  }
  __syncthreads();
  if (tid < (int)n) {                                           // thread i finishes row i
    uint64_t s = 0;
    for (uint32_t slot = 0; slot < per_row; ++slot) { s += s_part[slot * n + tid]; if (s >= q) s -= q; }
    const uint64_t u = U[b * n + tid] % q;
    s_part[tid] = u >= s ? u - s : u + q - s;                   // v_i (the partial sums of threads < n were read by their owners only: row i, slot 0 = thread i)
  }
  __syncthreads();
  // 3: z <- the gadget walk, one wave per row of v
  for (uint32_t j = tid >> 6; j < n; j += FS_THREADS / 64)
    f |= gadget_wave_problem(seed, index, j, k, base, s_part[j], tb, [&](int r, int zz) { s_z[j * k + r] = zz; });
  __syncthreads();
  // 4: e = p + [R; I] z
  for (uint32_t i = tid; i < m; i += FS_THREADS) {
    int64_t e = (int64_t)s_p[i];
    if (i < mb) {
      const int8_t* rr = R + (size_t)i * ldr;
      for (uint32_t c = 0; c < w; ++c) e += (int64_t)rr[c] * (int64_t)s_z[c];
    } else {
      e += (int64_t)s_z[i - mb];
    }
    E[b * m + i] = e;
  }
  if (f) atomicOr(fail, 1);
}

// e = p + [R; I] z (mp_perturbation.rs:328-335) for a handful of preimages (B <= NB <= 4): R is read ONCE, as a stream of 16-byte row pieces, by one wave per
// row (all of a row's pieces in flight at once, 15 KB at C3); z of every preimage sits in LDS as its two balanced int8 digit planes in the planes' own 16-byte
// groups, so a lane's piece of R meets the matching 16 digits with four v_dot4_i32_i8 per plane.  Integer sums: any order is exact.  The matrix-core kernel
// pads such a call to 128 columns and splits K over atomics (68 us + 15 us for the bottom part at C3, one preimage); this one is bound by reading R's 237 MB.
// LDS: 32 B x ldr / 16 x B.  Rows are dealt to the workgroups' waves round robin; the identity block (e_bot = p_bot + z) rides along.
template <int NB>
__global__ __launch_bounds__(512) void k_recombine_small(const int8_t* __restrict__ R, size_t ldr, size_t mbar, size_t w, const int8_t* __restrict__ Zlo,
                                                         const int8_t* __restrict__ Zhi, size_t ld, const int32_t* __restrict__ P, size_t B, int64_t* __restrict__ E, size_t m) {
  extern __shared__ int4 s_zd[];                                // [b][plane][group]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = (int)(ldr / 16);
  for (int e = tid; e < (int)B * 2 * ng; e += 512) {
    const int g = e % ng, pl = (e / ng) & 1, b = e / (2 * ng);
    s_zd[e] = *reinterpret_cast<const int4*>((pl ? Zhi : Zlo) + ((size_t)g * ld + (size_t)b) * 16);
  }
  __syncthreads();
  // identity block
  const int8_t* zb = reinterpret_cast<const int8_t*>(s_zd);
  for (size_t g = (size_t)blockIdx.x * 512 + tid; g < w * B; g += (size_t)gridDim.x * 512) {
    const size_t b = g / w, c = g % w;
    const size_t at = ((b * 2) * (size_t)ng + (c >> 4)) * 16 + (c & 15);
    E[b * m + mbar + c] = (int64_t)P[(mbar + c) * ld + b] + (int64_t)zb[at] + 256 * (int64_t)zb[at + (size_t)ng * 16];
  }
  // trapdoor block: one wave per row
  for (size_t i = (size_t)blockIdx.x * 8 + wave; i < mbar; i += (size_t)gridDim.x * 8) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i* row = reinterpret_cast<const v4i*>(R + i * ldr);
    int accl[NB], acch[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { accl[b] = 0; acch[b] = 0; }
    for (int g0 = 0; g0 < ng; g0 += 16 * 64) {
      v4i r[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int g = g0 + u * 64 + lane;
        r[u] = g < ng ? __builtin_nontemporal_load(row + g) : v4i{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int g = g0 + u * 64 + lane;
        if (g0 + u * 64 >= ng) break;                          // (wave-uniform)
        const int gc = g < ng ? g : 0;                         // (a lane beyond the row holds r = 0)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b >= (int)B) break;
          const int4 zl = s_zd[(b * 2) * ng + gc], zh = s_zd[(b * 2 + 1) * ng + gc];
          accl[b] = __builtin_amdgcn_sdot4(r[u].x, zl.x, accl[b], false); accl[b] = __builtin_amdgcn_sdot4(r[u].y, zl.y, accl[b], false);
          accl[b] = __builtin_amdgcn_sdot4(r[u].z, zl.z, accl[b], false); accl[b] = __builtin_amdgcn_sdot4(r[u].w, zl.w, accl[b], false);
          acch[b] = __builtin_amdgcn_sdot4(r[u].x, zh.x, acch[b], false); acch[b] = __builtin_amdgcn_sdot4(r[u].y, zh.y, acch[b], false);
          acch[b] = __builtin_amdgcn_sdot4(r[u].z, zh.z, acch[b], false); acch[b] = __builtin_amdgcn_sdot4(r[u].w, zh.w, acch[b], false);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= (int)B) break;
      long long v = (long long)accl[b] + 256 * (long long)acch[b];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) E[(size_t)b * m + i] = (int64_t)P[i * ld + (size_t)b] + (int64_t)v;
    }
  }
}

// The same product on a TWO-BIT copy of R (59 MB instead of 237 MB at C3): a trapdoor drawn from PlusMinusOneZero (trapdoor_distribution.rs:82-97) has entries
// in {-1, 0, 1}.  k_pack_R2 stores one 32-bit word per 16-byte group of the int8 matrix: bit e = "entry e is +1", bit 16 + e = "entry e is -1", and raises *bad for
// any other entry (a caller's own R through psfp_load_key / psf_gen_trapdoor_with_r: the int8 kernel above serves it).  A lane's 16-byte piece is four words = 64
// entries; a nibble of a mask (four entries) becomes four {0, 1} bytes by ONE 24-bit multiply (n * 0x204081 puts bit t at bit 8 t, no two terms overlap) and meets
// the digits of z in a v_dot4: e = p + (sum over +1) - (sum over -1).  Integer sums: any order is exact, the same e as the int8 kernel.  The upper digit plane of z
// is skipped when it is zero (every |z| <= 127: the common call).
__global__ void k_pack_R2(const int8_t* __restrict__ R, size_t ldr, size_t mbar, uint32_t* __restrict__ R2, int* __restrict__ bad) {
  const size_t ng = ldr / 16, total = mbar * ng;
  int b = 0;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / ng, gg = g % ng;
    const int8_t* src = R + i * ldr + gg * 16;
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int v = src[e];
      if (v > 1 || v < -1) b = 1;
      if (v == 1) w |= 1u << e;
      if (v == -1) w |= 1u << (16 + e);
    }
    R2[g] = w;
  }
  if (b) atomicOr(bad, 1);
}

template <int NB>
__global__ __launch_bounds__(512) void k_recombine_small2(const uint32_t* __restrict__ R2, size_t ldr, size_t mbar, size_t w, const int8_t* __restrict__ Zlo,
                                                          const int8_t* __restrict__ Zhi, size_t ld, const int32_t* __restrict__ P, size_t B, int64_t* __restrict__ E, size_t m) {
  extern __shared__ int4 s_zd[];                                // [b][plane][group]
  __shared__ int s_hi;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = (int)(ldr / 16);
  if (tid == 0) s_hi = 0;
  __syncthreads();
  int any_hi = 0;
  for (int e = tid; e < (int)B * 2 * ng; e += 512) {
    const int g = e % ng, pl = (e / ng) & 1, b = e / (2 * ng);
    const int4 v = *reinterpret_cast<const int4*>((pl ? Zhi : Zlo) + ((size_t)g * ld + (size_t)b) * 16);
    s_zd[e] = v;
    if (pl) any_hi |= v.x | v.y | v.z | v.w;
  }
  if (any_hi) s_hi = 1;
  __syncthreads();
  const bool use_hi = s_hi != 0;                                 // (workgroup-uniform)
  const int8_t* zb = reinterpret_cast<const int8_t*>(s_zd);
  for (size_t g = (size_t)blockIdx.x * 512 + tid; g < w * B; g += (size_t)gridDim.x * 512) {      // identity block
    const size_t b = g / w, c = g % w;
    const size_t at = ((b * 2) * (size_t)ng + (c >> 4)) * 16 + (c & 15);
    E[b * m + mbar + c] = (int64_t)P[(mbar + c) * ld + b] + (int64_t)zb[at] + 256 * (int64_t)zb[at + (size_t)ng * 16];
  }
  const int np = ng / 4;                                         // 16-byte pieces (four groups) per row; ldr is a multiple of 64
  for (size_t i = (size_t)blockIdx.x * 8 + wave; i < mbar; i += (size_t)gridDim.x * 8) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u* row = reinterpret_cast<const v4u*>(R2 + i * (size_t)ng);
    int accp[NB][2], accn[NB][2];                               // [preimage][digit plane]: sums over the +1 / the -1 entries
#pragma unroll
    for (int b = 0; b < NB; ++b) { accp[b][0] = accp[b][1] = accn[b][0] = accn[b][1] = 0; }
    for (int p0 = 0; p0 < np; p0 += 4 * 64) {
      v4u r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = p0 + u * 64 + lane;
        r[u] = pc < np ? __builtin_nontemporal_load(row + pc) : v4u{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p0 + u * 64 >= np) break;                            // (wave-uniform)
        const int pc = p0 + u * 64 + lane;
        const int g4 = (pc < np ? pc : 0) * 4;                   // (a lane beyond the row holds r = 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) {                            // group g4 + k = word k of the piece
          const unsigned int wv = r[u][k];
          int pos[4], neg[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pos[j] = (int)(__umul24((wv >> (4 * j)) & 0xfu, 0x204081u) & 0x01010101u);
            neg[j] = (int)(__umul24((wv >> (16 + 4 * j)) & 0xfu, 0x204081u) & 0x01010101u);
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            if (b >= (int)B) break;
            {
              const int4 z = s_zd[(b * 2) * ng + g4 + k];
              const int zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                accp[b][0] = __builtin_amdgcn_sdot4(pos[j], zz[j], accp[b][0], false);
                accn[b][0] = __builtin_amdgcn_sdot4(neg[j], zz[j], accn[b][0], false);
              }
            }
            if (use_hi) {
              const int4 z = s_zd[(b * 2 + 1) * ng + g4 + k];
              const int zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                accp[b][1] = __builtin_amdgcn_sdot4(pos[j], zz[j], accp[b][1], false);
                accn[b][1] = __builtin_amdgcn_sdot4(neg[j], zz[j], accn[b][1], false);
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= (int)B) break;
      long long v = (long long)(accp[b][0] - accn[b][0]) + 256 * (long long)(accp[b][1] - accn[b][1]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) E[(size_t)b * m + i] = (int64_t)P[i * ld + (size_t)b] + (int64_t)v;
    }
  }
}

// A p mod q (the sum of v = u - A p, mp_perturbation.rs:318) for a handful of preimages (B <= NB <= 4): A is read ONCE as 64-bit words, row by row, by one
// wave per (row, K range); the range's entries of p (int32, |p| < 2^23) sit in LDS.  A word a < 2^62 meets p as two signed 64-bit sums (low and high 32 bits of a:
// at most SYN_KLEN / 64 = 32 terms of < 2^55 per lane), joined in 128 bits, reduced over the wave and taken mod q once -- integer arithmetic, exact in any order.
// The residues go where the matrix-core product puts its per-split residues ([split][n_pad][ld]); k_zq_combine_wave finishes v as before.  The int8 matrix-core
// path (digit planes of p, LDS-staged tiles, 46 + 6 us at C3 for one preimage) is bound by its staging; this one by reading A's 126 MB.
constexpr int SYN_KLEN = 2048;
// q <= 2^32: the same from a 32-bit copy of A (k_narrow_A32: half the bytes, 63 MB at C3).  A lane takes two consecutive entries per 8-byte load; one signed
// 64-bit sum per preimage (32 terms of < 2^55 per lane), reduced over the wave, one division per (row, range).
__global__ void k_narrow_A32(const uint64_t* __restrict__ A, size_t total, uint32_t* __restrict__ A32) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) A32[g] = (uint32_t)A[g];
}
template <int NB>
__global__ __launch_bounds__(512) void k_syndrome_small32(const uint32_t* __restrict__ A32, size_t n, size_t m, const int32_t* __restrict__ P, size_t ld, size_t B, uint64_t q,
                                                          int rows_per_wg, uint64_t* __restrict__ part, size_t n_pad, size_t col0) {
  __shared__ int s_p[NB * SYN_KLEN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t j0 = (size_t)blockIdx.x * SYN_KLEN;
  const int klen = (int)(m - j0 < (size_t)SYN_KLEN ? m - j0 : (size_t)SYN_KLEN);
  for (int e = tid; e < NB * SYN_KLEN; e += 512) {
    const int b = e / SYN_KLEN, jj = e % SYN_KLEN;
    s_p[e] = (b < (int)B && jj < klen) ? P[(j0 + (size_t)jj) * ld + col0 + (size_t)b] : 0;
  }
  __syncthreads();
  const size_t i_end = ((size_t)blockIdx.y + 1) * (size_t)rows_per_wg < n ? ((size_t)blockIdx.y + 1) * (size_t)rows_per_wg : n;
  for (size_t i = (size_t)blockIdx.y * (size_t)rows_per_wg + wave; i < i_end; i += 8) {
    const uint32_t* a = A32 + i * m + j0;                        // (row starts are 4-byte aligned only: m may be odd -- two 4-byte loads per lane, adjacent addresses)
    long long acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0;
#pragma unroll 1
    for (int h0 = 0; h0 < SYN_KLEN / 128; h0 += 8) {
      if (h0 * 128 >= klen) break;
      uint32_t av[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jj = (h0 + u) * 128 + 2 * lane;
        av[u][0] = jj < klen ? __builtin_nontemporal_load(a + jj) : 0u;
        av[u][1] = jj + 1 < klen ? __builtin_nontemporal_load(a + jj + 1) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jj = (h0 + u) * 128 + 2 * lane;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          acc[b] += (long long)av[u][0] * (long long)s_p[b * SYN_KLEN + jj];
          acc[b] += (long long)av[u][1] * (long long)s_p[b * SYN_KLEN + jj + 1];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= (int)B) break;
      __int128 v = (__int128)acc[b];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long ol = __shfl_xor((unsigned long long)v, off);
        const long long oh = __shfl_xor((long long)(v >> 64), off);
        v += (((__int128)oh) << 64) | (__int128)ol;
      }
      if (lane == 0) {
        __int128 r = v % (__int128)q;
        if (r < 0) r += (__int128)q;
        part[((size_t)blockIdx.x * n_pad + i) * ld + col0 + (size_t)b] = (uint64_t)r;
      }
    }
  }
}
template <int NB>
__global__ __launch_bounds__(512) void k_syndrome_small(const uint64_t* __restrict__ A, size_t n, size_t m, const int32_t* __restrict__ P, size_t ld, size_t B, uint64_t q,
                                                        int rows_per_wg, uint64_t* __restrict__ part, size_t n_pad, size_t col0) {
  __shared__ int s_p[NB * SYN_KLEN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t j0 = (size_t)blockIdx.x * SYN_KLEN;
  const int klen = (int)(m - j0 < (size_t)SYN_KLEN ? m - j0 : (size_t)SYN_KLEN);
  for (int e = tid; e < NB * SYN_KLEN; e += 512) {
    const int b = e / SYN_KLEN, jj = e % SYN_KLEN;
    s_p[e] = (b < (int)B && jj < klen) ? P[(j0 + (size_t)jj) * ld + col0 + (size_t)b] : 0;
  }
  __syncthreads();
  const size_t i_end = ((size_t)blockIdx.y + 1) * (size_t)rows_per_wg < n ? ((size_t)blockIdx.y + 1) * (size_t)rows_per_wg : n;
  for (size_t i = (size_t)blockIdx.y * (size_t)rows_per_wg + wave; i < i_end; i += 8) {
    const uint64_t* a = A + i * m + j0;
    long long lo[NB], hi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { lo[b] = 0; hi[b] = 0; }
#pragma unroll 1
    for (int h0 = 0; h0 < SYN_KLEN / 64; h0 += 8) {
      if (h0 * 64 >= klen) break;
      uint64_t av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int jj = (h0 + u) * 64 + lane; av[u] = jj < klen ? __builtin_nontemporal_load(a + jj) : 0ull; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jj = (h0 + u) * 64 + lane;
        const long long al = (long long)(uint32_t)av[u], ah = (long long)(av[u] >> 32);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const long long pv = (long long)s_p[b * SYN_KLEN + jj];
          lo[b] += al * pv;
          hi[b] += ah * pv;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= (int)B) break;
      __int128 v = (__int128)lo[b] + (((__int128)hi[b]) << 32);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long ol = __shfl_xor((unsigned long long)v, off);
        const long long oh = __shfl_xor((long long)(v >> 64), off);
        v += (((__int128)oh) << 64) | (__int128)ol;
      }
      if (lane == 0) {
        __int128 r = v % (__int128)q;
        if (r < 0) r += (__int128)q;
        part[((size_t)blockIdx.x * n_pad + i) * ld + col0 + (size_t)b] = (uint64_t)r;
      }
    }
  }
}

}  // namespace psf
