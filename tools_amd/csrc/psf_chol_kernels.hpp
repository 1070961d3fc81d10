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

// ---- diagonal block + its inverse (left-looking form) ----------------------------------------------------------------------------------
// 256 threads; the block lives in LDS as [128][129].  Step j: pivot, column j scaled by the threads i > j, then the rank-1 update of the rows
// below -- thread (i = t % 128, h = t / 128) takes the columns j < c <= i with c = j + 1 + h (mod 2): lanes walk rows (stride 129: no bank
// conflict), the pivot column is a broadcast.  L11 goes back to A; the inverse (lower triangular, written to Linv row-major, zeros above the
// diagonal) is formed in place from the last column to the first: X[i][j] = -(sum_{j < t <= i} X[i][t] L[t][j]) / L[j][j].
__global__ __launch_bounds__(256) void k_chol_diag_inv(double* __restrict__ A, size_t ld, size_t off, int nb, double* __restrict__ Linv, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double ch_smem[];   // nb x (CH_NB + 1) | one column
  constexpr int LD = CH_NB + 1;
  double* __restrict__ sG = ch_smem;
  double* __restrict__ sCol = ch_smem + CH_NB * LD;
  const int tid = threadIdx.x, ti = tid & 127, th = tid >> 7;
  if (*info != 0) return;
  for (int e = tid; e < nb * CH_NB; e += 256) {
    const int r = e >> 7, c = e & 127;
    if (c < nb) sG[r * LD + c] = (c <= r) ? A[(off + r) * ld + off + c] : 0.0;
  }
  __syncthreads();
  double* __restrict__ myrow = sG + ti * LD;
  for (int j = 0; j < nb; ++j) {
    const double d = sG[j * LD + j];
    if (!(d > 0.0)) {                       // not positive definite (uniform: every thread reads the same word)
      if (tid == 0) atomicCAS(info, 0, (int)(off + j + 1));
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

// ---- panel below the diagonal block: one row per thread ------------------------------------------------------------
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

// ---- trailing update on the matrix cores ---------------------------------------------------------------------------------
// tile (ti, tj), tj <= ti, of the trailing matrix (tile size 128): C -= P_i P_j^t with P = the freshly solved panel (K = 128).
// 4 waves (2 x 2), wave tile 64 x 64; K chunks of 16 staged by LDS-DMA as [row][16]; A and B fragments are read the same
// way because both operands are row-major in k (an "NT" product).
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

}  // namespace psf
