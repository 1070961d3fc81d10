// psf_chol_kernels.hpp -- blocked Cholesky of Sigma_2 (mp_perturbation.rs:138, cholesky_decomposition_flint) on a dense row-major
// matrix whose LOWER triangle is significant.  Round 3: LEFT-looking, panel width 128, on the FP64 product of psf_gemm_kernels.hpp:
//   panel j   -=  L[j.., 0..j) L[j..j+128, 0..j)^t          one GEMM, K = j (split along K when few row tiles are left); every finished column
//                                                            of L is READ once per panel and nothing but the panel is written -- the right-looking
//                                                            form of rounds 1-2 re-wrote the whole trailing matrix per panel (0.9 TB at C3, K = 128)
//   k_chol_diag_inv : factor the 128 x 128 diagonal block in LDS (one workgroup), write L11, invert it in place (reports a non-positive pivot, :109-110)
//   rows below      =  panel L11^-t                          one GEMM, K = 128, against the inverse
// The right-looking kernels (k_chol_diag / k_chol_trsm / k_chol_syrk) stay as the PSF_CHOL=right comparison arm.
// Setup path only (once per key); the factor is compared with the oracle's unblocked one within a tolerance.
#pragma once
#include "psf_kernels.hpp"
#include "psf_gemm_kernels.hpp"

namespace psf {

constexpr int CH_NB = 128;


// ---- diagonal block ---------------------------------------------------------------------------------------------
#ifdef PSF_EXPERIMENTS   /* rounds 1-2: the right-looking Cholesky (PSF_CHOL=right), comparison arm of the experiments build */
__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ A, size_t ld, size_t off, int nb, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double ch_smem[];   // nb x (CH_NB + 1)
  constexpr int LD = CH_NB + 1;
  const int tid = threadIdx.x;
  for (int e = tid; e < nb * nb; e += 256) {
    const int r = e / nb, c = e % nb;
    ch_smem[r * LD + c] = (c <= r) ? A[(off + r) * ld + off + c] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    const double d = ch_smem[j * LD + j];
    if (!(d > 0.0)) {                       // not positive definite
      if (tid == 0) atomicCAS(info, 0, (int)(off + j + 1));
      return;
    }
    const double sd = sqrt(d);
    __syncthreads();
    for (int i = j + tid; i < nb; i += 256) ch_smem[i * LD + j] = (i == j) ? sd : ch_smem[i * LD + j] / sd;
    __syncthreads();
    // trailing update of the lower triangle: T[i][c] -= T[i][j] T[c][j], j < c <= i
    const int rem = nb - j - 1;
    for (int e = tid; e < rem * rem; e += 256) {
      const int i = j + 1 + e / rem, c = j + 1 + e % rem;
      if (c <= i) ch_smem[i * LD + c] = fma(-ch_smem[i * LD + j], ch_smem[c * LD + j], ch_smem[i * LD + c]);
    }
    __syncthreads();
  }
  for (int e = tid; e < nb * nb; e += 256) {
    const int r = e / nb, c = e % nb;
    if (c <= r) A[(off + r) * ld + off + c] = ch_smem[r * LD + c];
  }
}
#endif

// ---- diagonal block + its inverse, BLOCKED (round 6) ---------------------------------------------------------------------------------------
// The step-by-step form below (rounds 3-5) walks 128 dependent columns with three workgroup barriers each, then 128 more for the inverse: 0.19 ms alone and 0.36 ms
// in line with the stream factorisation, 241 times per C3 key = 86 of trap_gen's 289 ms (profiles/r06_keygen_timeline_c3.txt).  Here the 128 x 128 block is cut into
// four 32 x 32 leaves: a leaf is factored AND inverted by ONE wave (LDS traffic ordered inside the wave: no workgroup barrier in the 32 + 31 dependent steps), the
// rows below take  L = A X_leaf^t  (8 threads per row, the row held in registers), the trailing block its rank-32 update in 4 x 4 register tiles -- 3 barriers per
// leaf instead of 96 -- and the inverse of the whole block follows from the leaf inverses by block substitution, level by level (X_ij = -X_ii sum_t L_it X_tj).
// The inverse lives in the unused upper triangle of the LDS tile, transposed (X[i][j] at [j][i]), its diagonal in a vector.  Same algorithm, another summation order:
// the factor agrees with the step form to rounding (the tests compare against the oracle's unblocked recurrence within 1e-10 of the largest entry).
constexpr int CH_LEAF = 32;
constexpr size_t CH_DIAG_LDS = ((size_t)CH_NB * (CH_NB + 1) + CH_NB + 3 * CH_LEAF * (CH_LEAF + 1) + 2) * sizeof(double);      // tile | 1 / diagonal | three 32 x 32 temporaries | flag

__device__ __forceinline__ void ch_wave_sync() {      // LDS writes of this wave before, LDS reads of this wave after (one wave works alone: no workgroup barrier)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cholesky of a 128 x 128 block held lower-triangular in sG (stride 129) by 256 threads, and its inverse: on return sG[r][c], c <= r, is L; sG[c][r], c < r, is
// (L^-1)[r][c]; sDinv[r] = 1 / L[r][r].  Returns 0, or 1 + the index of the first non-positive pivot.
__device__ inline int chol128_blocked(double* __restrict__ sG, double* __restrict__ sDinv, double* __restrict__ sTmp, int* __restrict__ sBad, int tid) {
  constexpr int LD = CH_NB + 1, NL = CH_LEAF, TL = CH_LEAF + 1;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) *sBad = 0;
  __syncthreads();
  for (int b = 0; b < CH_NB; b += NL) {
    if (wave == 0) {
      const int i = lane & 31, h = lane >> 5;
      double* __restrict__ row = sG + (size_t)(b + i) * LD + b;
      int bad = 0;
      for (int j = 0; j < NL; ++j) {                                   // the leaf's factor: column j, then the rank-1 update of the columns behind it (two lanes per row)
        const double d = sG[(size_t)(b + j) * LD + b + j];
        if (!(d > 0.0)) { bad = b + j + 1; break; }                   // (every lane reads the same word)
        const double sd = sqrt(d), rs = 1.0 / sd;
        if (h == 0 && i >= j) {
          row[j] = (i == j) ? sd : row[j] * rs;
          if (i == j) sDinv[b + j] = rs;
        }
        ch_wave_sync();
        if (i > j) {
          const double nl = -row[j];
          for (int c = j + 1 + h; c <= i; c += 2) row[c] = fma(nl, sG[(size_t)(b + c) * LD + b + j], row[c]);
        }
        ch_wave_sync();
      }
      if (bad) { if (lane == 0) *sBad = bad; }
      else {
        // the leaf's inverse, row by row: X[i][j] = -(1 / L[i][i]) sum_{j <= t < i} L[i][t] X[t][j] for every column j < i at once (lane = column, the two halves share the sum)
        const int jj = lane & 31;
        double* __restrict__ xcol = sG + (size_t)(b + jj) * LD + b;   // X[t][jj] at xcol[t] (t > jj)
        for (int r = 1; r < NL; ++r) {
          double acc = 0.0;
          if (jj < r) {
            const double* __restrict__ lrow = sG + (size_t)(b + r) * LD + b;
            for (int t = jj + h; t < r; t += 2) acc = fma(lrow[t], t == jj ? sDinv[b + jj] : xcol[t], acc);
          }
          acc += __shfl_xor(acc, 32);
          if (h == 0 && jj < r) xcol[r] = -acc * sDinv[b + r];
          ch_wave_sync();
        }
      }
    }
    __syncthreads();
    if (*sBad) return *sBad;
    const int below = CH_NB - (b + NL);                                // rows under the leaf
    if (below > 0) {
      // L[r][b + c] = sum_{t <= c} A[r][b + t] X[c][t]: 8 threads per row (columns c = c0, c0 + 8, ...), the row in registers before anything is written
      for (int base = 0; base < below; base += 32) {
        const int r = b + NL + base + (tid >> 3), c0 = tid & 7;
        double a[NL];
        double* __restrict__ row = sG + (size_t)r * LD + b;
#pragma unroll
        for (int t = 0; t < NL; ++t) a[t] = row[t];
        ch_wave_sync();                                                // (the eight threads of a row sit in one wave: its loads are done before its stores)
        double o[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < NL; ++t) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + 8 * u;
            if (t <= c) o[u] = fma(a[t], t == c ? sDinv[b + c] : sG[(size_t)(b + t) * LD + b + c], o[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) row[c0 + 8 * u] = o[u];
      }
      __syncthreads();
      // trailing block -= panel panel^t, the lower triangle in 4 x 4 tiles
      const int nt = below / 4, ntiles = nt * (nt + 1) / 2;
      for (int idx = tid; idx < ntiles; idx += 256) {
        int tr = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
        while (tr * (tr + 1) / 2 > idx) --tr;
        while ((tr + 1) * (tr + 2) / 2 <= idx) ++tr;
        const int tc = idx - tr * (tr + 1) / 2;
        const int r0 = b + NL + 4 * tr, c0 = b + NL + 4 * tc;
        double acc[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
#pragma unroll 8
        for (int t = 0; t < NL; ++t) {
          double pr[4], pc[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) { pr[x] = sG[(size_t)(r0 + x) * LD + b + t]; pc[x] = sG[(size_t)(c0 + x) * LD + b + t]; }
#pragma unroll
          for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = fma(pr[x], pc[y], acc[x][y]);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y)
            if (c0 + y <= r0 + x) sG[(size_t)(r0 + x) * LD + c0 + y] -= acc[x][y];
      }
      __syncthreads();
    }
  }
  // the blocks of the inverse under the diagonal, by distance from it: X_ij = -X_ii (sum_{j <= t < i} L_it X_tj)
  constexpr int NBL = CH_NB / NL;
  for (int dist = 1; dist < NBL; ++dist) {
    const int nblk = NBL - dist;
    for (int o = tid; o < nblk * NL * NL; o += 256) {                  // S = sum_t L_it X_tj
      const int blk = o / (NL * NL), r = (o / NL) % NL, c = o % NL;
      const int bj = blk, bi = blk + dist;
      const double* __restrict__ lrow = sG + (size_t)(NL * bi + r) * LD;
      const double* __restrict__ xcol = sG + (size_t)(NL * bj + c) * LD;     // X[.][NL bj + c] at xcol[.]
      double acc = 0.0;
      for (int kk = c; kk < NL; ++kk) acc = fma(lrow[NL * bj + kk], kk == c ? sDinv[NL * bj + c] : xcol[NL * bj + kk], acc);      // t = j: the leaf's own (triangular) inverse
      for (int t = bj + 1; t < bi; ++t)
#pragma unroll 8
        for (int kk = 0; kk < NL; ++kk) acc = fma(lrow[NL * t + kk], xcol[NL * t + kk], acc);
      sTmp[(size_t)blk * NL * TL + r * TL + c] = acc;
    }
    __syncthreads();
    for (int o = tid; o < nblk * NL * NL; o += 256) {                  // X_ij = -X_ii S (X_ii lower triangular)
      const int blk = o / (NL * NL), r = (o / NL) % NL, c = o % NL;
      const int bj = blk, bi = blk + dist;
      const double* __restrict__ S = sTmp + (size_t)blk * NL * TL;
      double acc = 0.0;
      for (int kk = 0; kk <= r; ++kk) acc = fma(kk == r ? sDinv[NL * bi + r] : sG[(size_t)(NL * bi + kk) * LD + NL * bi + r], S[kk * TL + c], acc);
      sG[(size_t)(NL * bj + c) * LD + NL * bi + r] = -acc;
    }
    __syncthreads();
  }
  return 0;
}

__global__ __launch_bounds__(256) void k_chol_diag_inv(double* __restrict__ A, size_t ld, size_t off, int nb, double* __restrict__ Linv, int* __restrict__ info, size_t report_base) {
  extern __shared__ __attribute__((aligned(16))) double ch_smem[];
  constexpr int LD = CH_NB + 1;
  double* __restrict__ sG = ch_smem;
  double* __restrict__ sDinv = ch_smem + CH_NB * LD;
  double* __restrict__ sTmp = sDinv + CH_NB;
  int* __restrict__ sBad = reinterpret_cast<int*>(sTmp + 3 * CH_LEAF * (CH_LEAF + 1));
  const int tid = threadIdx.x;
  if (*info != 0) return;
#ifndef CHOL_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);      // a short dependent chain beside whatever shares its CU
#endif
  for (int e = tid; e < CH_NB * CH_NB; e += 256) {                     // a ragged last block is completed by the identity: its factor and inverse are the block's own, bordered by I
    const int r = e >> 7, c = e & 127;
    if (c <= r) sG[r * LD + c] = (r < nb && c < nb) ? A[(off + r) * ld + off + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  const int bad = chol128_blocked(sG, sDinv, sTmp, sBad, tid);
  if (bad) {                                                           // not positive definite (mp_perturbation.rs:109-110)
    if (tid == 0) atomicCAS(info, 0, (int)(report_base + (size_t)bad));
    return;
  }
  for (int e = tid; e < nb * CH_NB; e += 256) {
    const int r = e >> 7, c = e & 127;
    if (c <= r && c < nb) A[(off + r) * ld + off + c] = sG[r * LD + c];
    if (r < nb && c < nb) Linv[(size_t)r * CH_NB + c] = c < r ? sG[c * LD + r] : (c == r ? sDinv[r] : 0.0);
  }
}

#ifdef PSF_EXPERIMENTS   /* rounds 3-5: the step-by-step form (PSF_CHOL_DIAG=steps), comparison arm of the experiments build */
// ---- diagonal block + its inverse (left-looking form) ----------------------------------------------------------------------------------
// 256 threads; the block lives in LDS as [128][129].  Step j: pivot, column j scaled by the threads i > j, then the rank-1 update of the rows
// below -- thread (i = t % 128, h = t / 128) takes the columns j < c <= i with c = j + 1 + h (mod 2): lanes walk rows (stride 129: no bank
// conflict), the pivot column is a broadcast.  L11 goes back to A; the inverse (lower triangular, written to Linv row-major, zeros above the
// diagonal) is formed in place from the last column to the first: X[i][j] = -(sum_{j < t <= i} X[i][t] L[t][j]) / L[j][j].
__global__ __launch_bounds__(256) void k_chol_diag_inv_steps(double* __restrict__ A, size_t ld, size_t off, int nb, double* __restrict__ Linv, int* __restrict__ info, size_t report_base) {
  extern __shared__ __attribute__((aligned(16))) double ch_smem[];   // nb x (CH_NB + 1) | one column
  constexpr int LD = CH_NB + 1;
  double* __restrict__ sG = ch_smem;
  double* __restrict__ sCol = ch_smem + CH_NB * LD;
  const int tid = threadIdx.x, ti = tid & 127, th = tid >> 7;
  if (*info != 0) return;
#ifndef CHOL_NO_SETPRIO
  // This workgroup is a chain of 128 dependent steps and runs beside the product kernels of the look-ahead (their workgroups share its CU): alone it takes
  // 0.19 ms, under them 0.88 ms, on the critical path of most panels (profiles/r04_notes.md).  Instruction-issue priority over the co-resident waves.
  __builtin_amdgcn_s_setprio(3);
#endif
  for (int e = tid; e < nb * CH_NB; e += 256) {
    const int r = e >> 7, c = e & 127;
    if (c < nb) sG[r * LD + c] = (c <= r) ? A[(off + r) * ld + off + c] : 0.0;
  }
  __syncthreads();
  double* __restrict__ myrow = sG + ti * LD;
  for (int j = 0; j < nb; ++j) {
    const double d = sG[j * LD + j];
    if (!(d > 0.0)) {                       // not positive definite (uniform: every thread reads the same word)
      if (tid == 0) atomicCAS(info, 0, (int)(report_base + j + 1));
      return;
    }
    const double sd = sqrt(d);
    __syncthreads();
    if (th == 0 && ti >= j && ti < nb) {      // column j of L, kept a second time in sCol: the update below reads it from there (no aliasing with the rows it writes)
      const double v = (ti == j) ? sd : myrow[j] / sd;
      myrow[j] = v;
      sCol[ti] = v;
    }
    __syncthreads();
    if (ti > j && ti < nb) {
      const double nl = -sCol[ti];
      int c = j + 1 + th;
      // eight columns at a time: loads first, then the fmas, then the stores -- written out so that the LDS round trips overlap whatever the compiler
      // can or cannot prove about the pointers
      for (; c + 14 <= ti; c += 16) {
        double p[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { p[u] = sCol[c + 2 * u]; g[u] = myrow[c + 2 * u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = fma(nl, p[u], g[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) myrow[c + 2 * u] = g[u];
      }
      for (; c <= ti; c += 2) myrow[c] = fma(nl, sCol[c], myrow[c]);
    }
    __syncthreads();
  }
  for (int e = tid; e < nb * CH_NB; e += 256) {
    const int r = e >> 7, c = e & 127;
    if (c <= r && c < nb) A[(off + r) * ld + off + c] = sG[r * LD + c];
  }
  __syncthreads();
  tri_inverse_inplace(sG, sCol, nb, tid);                  // X = L11^-1 in place (psf_gemm_kernels.hpp)
  for (int e = tid; e < nb * CH_NB; e += 256) {
    const int r = e >> 7, c = e & 127;
    if (r < nb && c < nb) Linv[(size_t)r * CH_NB + c] = (c <= r) ? sG[r * LD + c] : 0.0;
  }
}
#endif

// ---- Cholesky directly on the key's chunk stream (round 3, the default) ---------------------------------------------------------------------
// sqrt(Sigma_2) is stored as the fragment-ordered chunk stream k_trmm_f64_big reads (psf_kernels.hpp: chunk (bi, c) = 128 rows x 16 columns).  A row block
// of that stream is at the same time the A operand AND -- its rows in the role of columns -- the B operand of an MFMA product, so the update of panel j,
//     P(bi, j)  =  Sigma_2(bi, j)  -  sum_{c < 8 j} L(bi, c) L(j, c)^t            for every row block bi >= j,
// is the register-streaming loop of k_trmm_f64_big run on two row blocks of the SAME stream (no LDS, no barrier, 256 AccVGPRs of accumulators, four
// k-steps in flight; the K loop below is that kernel's, statement for statement).  The dense m x m matrix of rounds 1-2 (7.6 GB at C3, 121 GB at C5)
// no longer exists: Sigma_2 is assembled panel by panel into a dense (m - 128 j) x 128 buffer, the product is subtracted from it, the diagonal block is
// factored and inverted (k_chol_diag_inv), the rows below are multiplied by the inverse (k_gemm_f64), and the finished panel is written into the
// stream (k_chol_pack_panel).  When a panel has fewer row blocks than the chip has CUs, K is cut over gridDim.y workgroups in units of two chunks; the
// partial tiles go to a workspace and k_chol_panel_reduce subtracts them in split order (no atomics).
// Panels are TWO column blocks (256 columns) wide: a workgroup owns one row block (128 rows) and both column blocks, wave (wr, wc) the rows 64 wr .. of it
// against column block 2 J + wc -- 4 A fragments (its own rows: read once per panel, from HBM) and 8 B fragments (the panel's row blocks: shared by every
// workgroup, from L2) per k-step for the same 32 MFMAs.  The A stream comes straight from HBM (nothing shares it), and four k-steps in flight do not
// cover that latency the way they cover L2's: 46 TFLOP/s at C5 with 128-wide panels (32 flop per unshared byte), 52 with 256 columns per pass over L.
__global__ __launch_bounds__(256, 1) void k_chol_update_big(const double* __restrict__ Lt, int J, int nbi, int rb0, int units_total,
                                                            double* __restrict__ ws, size_t ws_stride) {
  const int bl = rb0 + (int)blockIdx.x;                                // row block, counted from the panel's first one
  const int bi = 2 * J + bl;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int cb = 2 * J + wc;                                           // this wave's column block
  // K range of this split, in units of two chunks (8 k-steps of 4 coordinates): [u0, u1) of units_total = 8 J
  const int nz = (int)gridDim.y, z = (int)blockIdx.y;
  const int u0 = (int)(((long long)units_total * z) / nz), u1 = (int)(((long long)units_total * (z + 1)) / nz);
  const int nsteps = (u1 - u0) * 8;
  if (cb >= nbi) return;                                               // an odd number of column blocks: the last panel has one (no barrier in this kernel)
  d4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) acc[i][jj] = d4{0.0, 0.0, 0.0, 0.0};
  if (nsteps > 0) {
    const double* gA = Lt + (tr_rowblock_base((size_t)bi) + (size_t)u0 * 2) * TR_CHUNK + (size_t)(wr * 4) * 64;        // wave-uniform
    const double* gB = Lt + (tr_rowblock_base((size_t)cb) + (size_t)u0 * 2) * TR_CHUNK;
    const uint32_t voff = (uint32_t)lane * 8u;
    double a[TR_BIG_PD][4], b[TR_BIG_PD][8];
    auto issue = [&](double (&av)[4], double (&bv)[8], int s) {
      const double* pa = gA + (size_t)s * 512;
      const double* pb = gB + (size_t)s * 512;
      TR_LOAD8(av[0], voff, pa, 0); TR_LOAD8(av[1], voff, pa, 512); TR_LOAD8(av[2], voff, pa, 1024); TR_LOAD8(av[3], voff, pa, 1536);
      TR_LOAD8(bv[0], voff, pb, 0); TR_LOAD8(bv[1], voff, pb, 512); TR_LOAD8(bv[2], voff, pb, 1024); TR_LOAD8(bv[3], voff, pb, 1536);
      TR_LOAD8(bv[4], voff, pb, 2048); TR_LOAD8(bv[5], voff, pb, 2560); TR_LOAD8(bv[6], voff, pb, 3072); TR_LOAD8(bv[7], voff, pb, 3584);
    };
#define CH_MFMA(i, jj) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i][jj]) : "v"(av[i]), "v"(bv[jj]))
    // one k-step: every operand register is reloaded right behind its last reader (as in k_trmm_f64_big, with the roles of A and B swapped)
    auto step = [&](double (&av)[4], double (&bv)[8], int sn) {
      const double* pa = gA + (size_t)sn * 512;
      const double* pb = gB + (size_t)sn * 512;
      CH_MFMA(0, 0); CH_MFMA(1, 0); CH_MFMA(2, 0); CH_MFMA(3, 0); TR_LOAD8(bv[0], voff, pb, 0);
      CH_MFMA(0, 1); CH_MFMA(1, 1); CH_MFMA(2, 1); CH_MFMA(3, 1); TR_LOAD8(bv[1], voff, pb, 512);
      CH_MFMA(0, 2); CH_MFMA(1, 2); CH_MFMA(2, 2); CH_MFMA(3, 2); TR_LOAD8(bv[2], voff, pb, 1024);
      CH_MFMA(0, 3); CH_MFMA(1, 3); CH_MFMA(2, 3); CH_MFMA(3, 3); TR_LOAD8(bv[3], voff, pb, 1536);
      CH_MFMA(0, 4); CH_MFMA(1, 4); CH_MFMA(2, 4); CH_MFMA(3, 4); TR_LOAD8(bv[4], voff, pb, 2048);
      CH_MFMA(0, 5); CH_MFMA(1, 5); CH_MFMA(2, 5); CH_MFMA(3, 5); TR_LOAD8(bv[5], voff, pb, 2560);
      CH_MFMA(3, 6); CH_MFMA(3, 7); TR_LOAD8(av[3], voff, pa, 1536);
      CH_MFMA(2, 6); CH_MFMA(2, 7); TR_LOAD8(av[2], voff, pa, 1024);
      CH_MFMA(1, 6); CH_MFMA(1, 7); TR_LOAD8(av[1], voff, pa, 512);
      CH_MFMA(0, 6); CH_MFMA(0, 7); TR_LOAD8(av[0], voff, pa, 0);
      TR_LOAD8(bv[6], voff, pb, 3072); TR_LOAD8(bv[7], voff, pb, 3584);
    };
#define CH_WAIT12(n, A, Bv) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(Bv[0]), "+v"(Bv[1]), "+v"(Bv[2]), "+v"(Bv[3]), \
                                         "+v"(Bv[4]), "+v"(Bv[5]), "+v"(Bv[6]), "+v"(Bv[7]))
#pragma unroll
    for (int u = 0; u < TR_BIG_PD; ++u) issue(a[u], b[u], u);
    for (int s0 = 0; s0 < nsteps; s0 += TR_BIG_PD * 2) {                // nsteps is a multiple of 8
#pragma unroll
      for (int rnd = 0; rnd < 2; ++rnd)
#pragma unroll
        for (int u = 0; u < TR_BIG_PD; ++u) {
          CH_WAIT12(36, a[u], b[u]);                                    // 12 (TR_BIG_PD - 1): all but the three newest k-steps have landed
          static_assert(TR_BIG_PD == 4, "the wait count above is 12 (TR_BIG_PD - 1)");
          int sn = s0 + rnd * TR_BIG_PD + u + TR_BIG_PD;
          sn = sn < nsteps ? sn : nsteps - 1;                           // past the end: re-read the last step (never consumed)
          step(a[u], b[u], sn);
        }
    }
#undef CH_MFMA
#pragma unroll
    for (int u = 0; u < TR_BIG_PD; ++u) CH_WAIT12(0, a[u], b[u]);
#undef CH_WAIT12
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMA retires before an accumulator is read (see k_trmm_f64_big)
  }
  // C/D map of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 reg.  Plain stores into the workspace (panel layout: rows from the panel's first
  // row, 256 columns), exactly the epilogue form of k_trmm_f64_big -- a read-modify-write of the panel here cost the register allocation its shape
  // (spills of loads in flight, accumulators shuttled out of the AccVGPRs inside the loop); k_chol_panel_reduce subtracts it also when K is not cut.
  const size_t row0 = (size_t)bl * TR_BM + wr * 64, col0 = (size_t)wc * TR_BM;
  double* __restrict__ Xo = ws + (size_t)z * ws_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
        const size_t row = row0 + i * 16 + (lane >> 4) + 4 * r;
        Xo[row * (2 * TR_BM) + col0 + jj * 16 + (lane & 15)] = acc[i][jj][r];
      }
}

// Pbuf -= sum_z ws[z] (split order) over `count` doubles from `first`
__global__ void k_chol_panel_reduce(double* __restrict__ Pbuf, const double* __restrict__ ws, size_t ws_stride, int splits, size_t first, size_t count) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * ws_stride + first + e];
    Pbuf[first + e] -= s;
  }
}

// the finished panel J (dense, leading dimension 256, rows = matrix rows 256 J ..) -> chunks (bi, 16 J + kc) of the stream, kc < 8 `ncb`; above the
// diagonal and beyond m: zero
// panel J of a dense, lower-stored Sigma_2 (leading dimension lds) into the panel buffer of the stream factorisation: P[i][j] = Sigma_2[off + i][off + j] for the rows
// off ... m-1 and the panel's `cols` columns; the part of the diagonal blocks above the diagonal is taken from its mirror image
__global__ void k_chol_copy_panel(const double* __restrict__ S, size_t lds, size_t m, size_t off, size_t cols, double* __restrict__ P, size_t pw) {
  const size_t rows = m - off, total = rows * cols;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / cols, j = g % cols;
    P[i * pw + j] = j <= i ? S[(off + i) * lds + off + j] : S[(off + j) * lds + off + i];
  }
}

__global__ void k_chol_pack_panel(const double* __restrict__ Pbuf, int J, int ncb, int nbi, size_t m, double* __restrict__ Lt) {
  const size_t per_rb = (size_t)ncb * 8 * TR_CHUNK;
  const size_t total = (size_t)(nbi - 2 * J) * per_rb;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t bl = g / per_rb, kc = (g % per_rb) / TR_CHUNK;       // row block (from 2 J), chunk inside the panel
    const int pos = (int)(g % TR_CHUNK);
    const int ks = pos >> 9, tile = (pos >> 6) & 7, ln = pos & 63;
    const size_t r = tile * 16 + (ln & 15), kk = kc * 16 + ks * 4 + (ln >> 4);
    const size_t bi = (size_t)2 * J + bl;
    const size_t row = bi * TR_BM + r, col = (size_t)2 * J * TR_BM + kk;
    if ((size_t)2 * J * 8 + kc >= 8 * (bi + 1)) continue;            // the stream has no chunk there (second column block of the panel's first row block)
    double v = 0.0;
    if (row < m && col <= row) v = Pbuf[(bl * TR_BM + r) * (2 * TR_BM) + kk];
    Lt[(tr_rowblock_base(bi) + (size_t)2 * J * 8 + kc) * TR_CHUNK + pos] = v;
  }
}

// ---- panel below the diagonal block: one row per thread ------------------------------------------------------------
#ifdef PSF_EXPERIMENTS   /* rounds 1-2, as k_chol_diag */
__global__ __launch_bounds__(64) void k_chol_trsm(double* __restrict__ A, size_t ld, size_t off, int nb, size_t m, const int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double ch_smem[];   // L11 packed lower (nb(nb+1)/2) | x[nb][64]
  if (*info != 0) return;
  const int tid = threadIdx.x;
  double* sL = ch_smem;
  double* sx = ch_smem + (size_t)nb * (nb + 1) / 2;
  for (int e = tid; e < nb * (nb + 1) / 2; e += 64) {
    int r = (int)((sqrt(1.0 + 8.0 * e) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    while (r * (r + 1) / 2 > e) --r;
    const int c = e - r * (r + 1) / 2;
    sL[e] = A[(off + r) * ld + off + c];
  }
  const size_t row = off + nb + (size_t)blockIdx.x * 64 + tid;
  const bool live = row < m;
  for (int j = 0; j < nb; ++j) sx[j * 64 + tid] = live ? A[row * ld + off + j] : 0.0;
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    const double* lj = sL + (size_t)j * (j + 1) / 2;
    double s = sx[j * 64 + tid];
    for (int t = 0; t < j; ++t) s = fma(-sx[t * 64 + tid], lj[t], s);
    sx[j * 64 + tid] = s / lj[j];
  }
  if (live)
    for (int j = 0; j < nb; ++j) A[row * ld + off + j] = sx[j * 64 + tid];
}
#endif

// ---- trailing update on the matrix cores ---------------------------------------------------------------------------------
// tile (ti, tj), tj <= ti, of the trailing matrix (tile size 128): C -= P_i P_j^t with P = the freshly solved panel (K = 128).
// 4 waves (2 x 2), wave tile 64 x 64; K chunks of 16 staged by LDS-DMA as [row][16]; A and B fragments are read the same
// way because both operands are row-major in k (an "NT" product).
#ifdef PSF_EXPERIMENTS   /* rounds 1-2, as k_chol_diag */
__global__ __launch_bounds__(256, 2) void k_chol_syrk(double* __restrict__ A, size_t ld, size_t off /*panel column offset*/, size_t m,
                                                      int ntiles_side, const int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double ch_smem[];   // 2 stages x (Pi chunk 2048 | Pj chunk 2048)
  if (*info != 0) return;
  // linear tile id -> (ti, tj) in the lower triangle of an ntiles_side x ntiles_side grid
  const unsigned id = blockIdx.x;
  int ti = (int)((sqrt(1.0 + 8.0 * (double)id) - 1.0) * 0.5);
  while ((unsigned)(ti + 1) * (ti + 2) / 2 <= id) ++ti;
  while ((unsigned)ti * (ti + 1) / 2 > id) --ti;
  const int tj = (int)(id - (unsigned)ti * (ti + 1) / 2);
  if (ti >= ntiles_side) return;
  const size_t r0 = off + CH_NB + (size_t)ti * 128, c0 = off + CH_NB + (size_t)tj * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  d4 acc[4][4];
  const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;       // the product P_i P_j^t alone; C is read in the epilogue (no spills)
  // a 16 KiB chunk = 128 rows x 16 doubles; each wave-instruction moves 8 rows x 128 B
  auto stage_load = [&](int kc, int buf) {
    double* base = ch_smem + buf * 4096;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int piece = wave * 4 + p;                 // 0..15, 8 rows each
      const int row = piece * 8 + (lane >> 3), seg = lane & 7;
      size_t ri = r0 + row, rj = c0 + row;
      if (ri >= m) ri = m - 1;                        // clamp: rows past the end only feed tiles that are never stored
      if (rj >= m) rj = m - 1;
      __builtin_amdgcn_global_load_lds(A + ri * ld + off + kc * 16 + seg * 2, (lds_void_ptr)(base + piece * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(A + rj * ld + off + kc * 16 + seg * 2, (lds_void_ptr)(base + 2048 + piece * 128), 16, 0, 0);
    }
  };
  stage_load(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kc = 0; kc < 8; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < 8) stage_load(kc + 1, cur ^ 1);
    const double* sA = ch_smem + cur * 4096;
    const double* sB = sA + 2048;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = sA[(wr * 64 + i * 16 + r16) * 16 + ks * 4 + g];
        b[i] = sB[(wc * 64 + i * 16 + r16) * 16 + ks * 4 + g];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t rr = r0 + wr * 64 + i * 16 + g + 4 * r, cc = c0 + wc * 64 + j * 16 + r16;
        if (rr < m && cc < m && cc <= rr) A[rr * ld + cc] -= acc[i][j][r];
      }
}
#endif

}  // namespace psf
