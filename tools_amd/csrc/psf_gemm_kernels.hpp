// psf_gemm_kernels.hpp -- a plain FP64 product on the matrix cores for the key-generation paths (row-major operands, any shape):
//     C = beta C + alpha (A op(B)) diag(colscale)         A: M x K,  op(B) = B^t with B: N x K ("NT")  or  B: K x N ("NN")
// and the blocked Gram-Schmidt orthogonalisation built on it (MatQ::gso, gpv.rs:91; inside MatPolyOverZ::sample_d, gpv_ring.rs:205).
//
// Workgroup tile 128 x 128, four waves (2 x 2) with 64 x 64 wave tiles of v_mfma_f64_16x16x4_f64, K chunks of 16.  Both operand tiles are kept
// K-MAJOR in LDS ([k][row], row stride 144 doubles, rows rotated by 2 (k >> 1)): the fragment of a k-step is 16 consecutive doubles per lane
// group, conflict free, and so is the transposing write of row-major operands whose K runs along the row (the [row][k] layout of round 1's SYRK
// put sixteen rows on two bank groups).  Sixteen consecutive lanes load the 128 contiguous bytes one row contributes to a chunk.
// Global loads of chunk t+1 are in flight while chunk t is multiplied; one barrier per chunk.
// Skinny products (M = one panel) are cut along K over gridDim.z workgroups; the partial tiles go to a workspace and k_gemm_reduce adds them
// in split order -- no atomics, so every result is reproducible bit for bit (every rank regenerates the key from the seed).
#pragma once
#include "psf_kernels.hpp"

namespace psf {

constexpr int GM_T = 128;        // tile edge
constexpr int GM_BK = 16;        // K chunk
constexpr int GM_LD = 144;       // LDS row stride (doubles) of a k-major tile
constexpr size_t GM_LDS_BYTES = 2 * 2 * GM_BK * GM_LD * sizeof(double);   // two stages x (A | B) = 73 728 B

struct GemmArgs {
  const double* A; size_t lda;
  const double* B; size_t ldb;
  double* C; size_t ldc;
  size_t M, N, K;
  double alpha, beta;
  const double* colscale;        // optional, length N: the product's column j is multiplied by colscale[j]
  double* ws;                    // split-K workspace (gridDim.z x M_pad x N_pad), or nullptr when gridDim.z == 1
  size_t klen;                   // K range per split (a multiple of GM_BK)
};

__device__ inline void gemm_epilogue(const GemmArgs& g, size_t row, size_t col, double v) {
  if (g.colscale) v *= g.colscale[col];
  double out = g.alpha * v;
  if (g.beta != 0.0) out = fma(g.beta, g.C[row * g.ldc + col], out);
  g.C[row * g.ldc + col] = out;
}

template <bool NT>
__global__ __launch_bounds__(256, 2) void k_gemm_f64(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double gm_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t m0 = (size_t)blockIdx.y * GM_T, n0 = (size_t)blockIdx.x * GM_T;
  const size_t kbeg = (size_t)blockIdx.z * g.klen;
  const size_t kend = kbeg + g.klen < g.K ? kbeg + g.klen : g.K;
  const int nk = kbeg < kend ? (int)((kend - kbeg + GM_BK - 1) / GM_BK) : 0;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  // operand staging.  A (and an NT B): load i of thread t brings element (row (t >> 4) + 16 i, k = t & 15) -- sixteen consecutive lanes read the 128
  // contiguous bytes a row contributes to the chunk, so every cache line is touched by ONE instruction (first version: eight 8-byte loads per lane
  // along its own row -- each line fetched through the L1 eight times, 15 k cycles per chunk).  An NN B: (k row t >> 4, column (t & 15) + 16 i).
  // In LDS element (kk, x) of a tile sits at kk * GM_LD + ((x + 2 (kk >> 1)) & 127): with GM_LD = 16 (mod 32) both the transposing ds_write_b64 of
  // a half wave (kk = 0..15, two rows) and the fragment ds_read_b64 (sixteen rows, two kk) fall on 32 distinct 8-byte slots.
  // The loads are UNCONDITIONAL (addresses clamped into the matrix): a select on a loaded value makes the compiler wait for the load where it is
  // issued, i.e. in front of the MFMAs it should hide behind.  Rows beyond M / columns beyond N only feed outputs that are never stored, so their
  // (finite) stand-in values are harmless; the K tail of a split is zeroed on ONE operand, A, after the loads have landed (0 x finite = 0).
  const int t_hi = tid >> 4, t_lo = tid & 15;
  const double* pa[8];
  const double* pb[8];
  size_t bcol[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t ar = m0 + t_hi + 16 * i, br = n0 + t_hi + 16 * i;
    pa[i] = g.A + (ar < g.M ? ar : g.M - 1) * g.lda;
    if (NT) pb[i] = g.B + (br < g.N ? br : g.N - 1) * g.ldb;
    else { const size_t c = n0 + t_lo + 16 * i; bcol[i] = c < g.N ? c : g.N - 1; }
  }
  const size_t klast = g.K - 1;
  double ra[8], rb[8];
  auto fetch = [&](int kt) {
    const size_t k0 = kbeg + (size_t)kt * GM_BK;
    size_t ka = k0 + t_lo, kb = k0 + t_hi;             // this thread's k for A / NT-B, and for an NN B
    if (k0 + GM_BK > g.K) { ka = ka < klast ? ka : klast; kb = kb < klast ? kb : klast; }
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = pa[i][ka];
    if (NT) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rb[i] = pb[i][ka];
    } else {
      const double* prow = g.B + kb * g.ldb;
#pragma unroll
      for (int i = 0; i < 8; ++i) rb[i] = prow[bcol[i]];
    }
  };
  auto stash = [&](int kt, int buf) {
    double* sA = gm_smem + buf * (2 * GM_BK * GM_LD);
    double* sB = sA + GM_BK * GM_LD;
    const size_t k0 = kbeg + (size_t)kt * GM_BK;
    const bool a_dead = k0 + t_lo >= kend;              // K tail of this split: its A entries count as zero
    const int rot_a = 2 * (t_lo >> 1), rot_b = 2 * (t_hi >> 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) sA[t_lo * GM_LD + ((t_hi + 16 * i + rot_a) & 127)] = a_dead ? 0.0 : ra[i];
    if (NT) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sB[t_lo * GM_LD + ((t_hi + 16 * i + rot_a) & 127)] = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) sB[t_hi * GM_LD + ((t_lo + 16 * i + rot_b) & 127)] = rb[i];
    }
  };
  const int r16 = lane & 15, gq = lane >> 4;
  if (nk > 0) {
    fetch(0);
    stash(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
#if !defined(GM_PROBE) || GM_PROBE != 1     /* tools/probe_gemm.hip: GM_PROBE=1 leaves the operands in LDS */
      if (kt + 1 < nk) fetch(kt + 1);
#endif
      const double* sA = gm_smem + cur * (2 * GM_BK * GM_LD);
      const double* sB = sA + GM_BK * GM_LD;
      // one fragment set: reading k-step ks + 1 into a second set while the MFMAs of k-step ks run measured no faster at two workgroups per CU (the other
      // workgroup's MFMAs cover the LDS round trip) and spills at the 256-register budget (profiles/r03_notes.md)
#pragma unroll
      for (int ks = 0; ks < GM_BK / 4; ++ks) {
        const int kk = ks * 4 + gq, rot = 2 * (kk >> 1);
        double fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = sA[kk * GM_LD + ((wr * 64 + i * 16 + r16 + rot) & 127)];
          fb[i] = sB[kk * GM_LD + ((wc * 64 + i * 16 + r16 + rot) & 127)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
#if defined(GM_PROBE) && GM_PROBE == 2      /* no MFMAs: loads, LDS traffic, barriers */
          for (int j = 0; j < 4; ++j) acc[i][j][0] += fa[i] * fb[j];
#else
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
#endif
      }
      if (kt + 1 < nk) stash(kt + 1, cur ^ 1);
      __syncthreads();
    }
  }
  // C/D map of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 reg
  const size_t Mp = (size_t)gridDim.y * GM_T, Np = (size_t)gridDim.x * GM_T;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t row = m0 + wr * 64 + i * 16 + gq + 4 * r, col = n0 + wc * 64 + j * 16 + r16;
        if (g.ws) g.ws[((size_t)blockIdx.z * Mp + row) * Np + col] = acc[i][j][r];
        else if (row < g.M && col < g.N) gemm_epilogue(g, row, col, acc[i][j][r]);
      }
}

// adds the partial products of the K splits in split order and applies the epilogue
__global__ void k_gemm_reduce(GemmArgs g, int splits, size_t Mp, size_t Np) {
  const size_t total = g.M * g.N;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t row = e / g.N, col = e % g.N;
    double s = 0.0;
    for (int z = 0; z < splits; ++z) s += g.ws[((size_t)z * Mp + row) * Np + col];
    gemm_epilogue(g, row, col, s);
  }
}

// ---- in-panel step of the blocked Gram-Schmidt: G = L D L^t (unit lower L, no pivoting), out = L^-1 ---------------------------------------------
// One workgroup; the p x p Gram matrix of the panel (p <= 128) lives in LDS.  Applying L^-1 to the panel's rows IS Gram-Schmidt on them in
// exact arithmetic (rows of L^-1 W are W's rows minus their components along the earlier ones); the caller runs the (Gram, LDL, apply) round
// twice, which brings the rows to orthogonality at rounding level for a panel whose condition is below ~1e7 ("CholQR2").
// X = T^-1 in place for a lower triangular T held as rows of stride 129 in LDS (256 threads: thread (ti = t & 127, th = t >> 7)), columns from the
// last to the first:  X[i][j] = -( sum_{j < t <= i} X[i][t] T[t][j] ) / T[j][j],  X[j][j] = 1 / T[j][j].  Both halves of the workgroup share a row's
// dot product (even / odd t), four partial sums each.  sCol: 2 x 128 doubles of scratch.
__device__ inline void tri_inverse_inplace(double* __restrict__ sG, double* __restrict__ sCol, int nb, int tid) {
  constexpr int LD = GM_T + 1;
  const int ti = tid & 127, th = tid >> 7;
  double* __restrict__ myrow = sG + ti * LD;
  for (int j = nb - 1; j >= 0; --j) {
    if (th == 0 && ti >= j && ti < nb) sCol[ti] = myrow[j];
    __syncthreads();
    double part = 0.0;
    if (ti > j && ti < nb) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int t = j + 1 + th;
      for (; t + 6 <= ti; t += 8) {
        s0 = fma(myrow[t], sCol[t], s0);
        s1 = fma(myrow[t + 2], sCol[t + 2], s1);
        s2 = fma(myrow[t + 4], sCol[t + 4], s2);
        s3 = fma(myrow[t + 6], sCol[t + 6], s3);
      }
      for (; t <= ti; t += 2) s0 = fma(myrow[t], sCol[t], s0);      // includes t = ti: X[i][i] is final (inverted when column i was processed)
      part = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();                          // every read of column j's old values and of the rows is done
    if (th == 1 && ti > j && ti < nb) sCol[GM_T + ti] = part;        // the second half hands its partial sum over
    __syncthreads();
    if (th == 0 && ti >= j && ti < nb) {
      const double inv = 1.0 / sCol[j];
      myrow[j] = (ti == j) ? inv : -(part + sCol[GM_T + ti]) * inv;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_ldl_inverse(const double* __restrict__ Gm, size_t ldg, int p, double* __restrict__ Linv, size_t ldl, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double ldl_smem[];   // G: 128 x 129 (132 KiB) | two columns (2 KiB)
  constexpr int LD = GM_T + 1;
  double* __restrict__ sG = ldl_smem;
  double* __restrict__ sCol = ldl_smem + GM_T * LD;
  const int tid = threadIdx.x, ti = tid & 127, th = tid >> 7;
#ifndef CHOL_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);      // a chain of 128 dependent steps in one workgroup: ahead of whatever else shares its CU (k_chol_diag_inv)
#endif
  for (int e = tid; e < p * GM_T; e += 256) { const int r = e >> 7, c = e & 127; if (c < p) sG[r * LD + c] = Gm[(size_t)r * ldg + c]; }
  __syncthreads();
  double* __restrict__ myrow = sG + ti * LD;
  for (int j = 0; j < p; ++j) {
    // column j as it stands (u = L[.][j] d_j) is copied to sCol; the update G[i][c] -= (u_i / d_j) u_c reads it from there
    if (th == 0 && ti >= j && ti < p) sCol[ti] = myrow[j];
    __syncthreads();
    const double dj = sCol[j];
    if (!(dj > 0.0) && tid == 0) atomicCAS(info, 0, j + 1);              // a dependent "basis": reported, the row is left as it is
    const double inv = dj > 0.0 ? 1.0 / dj : 0.0;
    if (ti > j && ti < p) {
      const double nl = -(sCol[ti] * inv);
      int c = j + 1 + th;
      for (; c + 14 <= ti; c += 16) {
        double u[8], gg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { u[k] = sCol[c + 2 * k]; gg[k] = myrow[c + 2 * k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) gg[k] = fma(nl, u[k], gg[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) myrow[c + 2 * k] = gg[k];
      }
      for (; c <= ti; c += 2) myrow[c] = fma(nl, sCol[c], myrow[c]);
      if (th == 0) myrow[j] = sCol[ti] * inv;                            // column j of L
    }
    __syncthreads();
  }
  if (th == 0 && ti < p) myrow[ti] = 1.0;                                // unit diagonal for the inversion
  __syncthreads();
  tri_inverse_inplace(sG, sCol, p, tid);
  for (int e = tid; e < p * GM_T; e += 256) {
    const int r = e >> 7, c = e & 127;
    if (c < p) Linv[(size_t)r * ldl + c] = c <= r ? sG[r * LD + c] : 0.0;
  }
}

// 1 / ||row||^2 of `rows` rows of length d (one wave per row, fixed summation order)
__global__ __launch_bounds__(256) void k_rows_inv_norm2(const double* __restrict__ W, size_t ld, size_t d, int rows, double* __restrict__ inv) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (size_t j = lane; j < d; j += 64) acc = fma(W[(size_t)r * ld + j], W[(size_t)r * ld + j], acc);
  acc = wave_xor_sum(acc);
  if (lane == 0) inv[r] = acc > 0.0 ? 1.0 / acc : 0.0;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------
struct GemmWorkspace { double* ws = nullptr; size_t bytes = 0; };

// launches C = beta C + alpha (A op(B)) diag(colscale); cuts K when the tile grid alone would leave most of the chip idle
template <bool NT>
inline void launch_gemm(hipStream_t st, GemmArgs g, GemmWorkspace& w) {
  const unsigned tx = (unsigned)((g.N + GM_T - 1) / GM_T), ty = (unsigned)((g.M + GM_T - 1) / GM_T);
  const size_t nchunks = (g.K + GM_BK - 1) / GM_BK;
  // K splits: the chip holds 512 workgroups of this kernel at once (2 per CU); a launch of T tiles x s splits runs in ceil(T s / 512) rounds, so s is
  // chosen to fill the rounds it occupies (120 tiles: s = 4 -> 480 workgroups in one round; s = 5 would leave a second round 17 % full).  Every split
  // keeps at least 8 chunks (128 coordinates); ties go to the smaller s (less workspace traffic).
  unsigned splits = 1;
  {
    const size_t T = (size_t)tx * ty;
    size_t smax = nchunks / 8 ? nchunks / 8 : 1;
    if (smax > 64) smax = 64;
    double best_fill = 0.0;
    for (size_t sc = 1; sc <= smax; ++sc) {
      const size_t wg = T * sc, rounds = (wg + 511) / 512;
      if ((size_t)sc * ty * GM_T * tx * GM_T * sizeof(double) > w.bytes && sc > 1) break;
      const double fill = (double)wg / (double)(rounds * 512);
      if (fill > best_fill + 0.02) { best_fill = fill; splits = (unsigned)sc; }
      if (wg >= 4 * 512) break;                                            // several full rounds anyway
    }
  }
  g.klen = ((nchunks + splits - 1) / splits) * GM_BK;
  splits = (unsigned)((g.K + g.klen - 1) / g.klen);
  if (splits < 1) splits = 1;
  double* ws = splits > 1 ? w.ws : nullptr;
  GemmArgs k = g;
  k.ws = ws;
  hipLaunchKernelGGL((k_gemm_f64<NT>), dim3(tx, ty, splits), dim3(256), GM_LDS_BYTES, st, k);
  if (splits > 1) {
    const size_t total = g.M * g.N;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gemm_reduce, dim3(blocks), dim3(256), 0, st, k, (int)splits, (size_t)ty * GM_T, (size_t)tx * GM_T);
  }
}

inline hipError_t gemm_prepare() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f64<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM_LDS_BYTES);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f64<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM_LDS_BYTES);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_ldl_inverse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((GM_T * (GM_T + 1) + 2 * GM_T) * sizeof(double)));
}

// Gram-Schmidt on the rows of Gt (nrows x d, row i = basis vector i as doubles on entry, b~_i on exit), panels of 128 rows:
// twice per panel ("BCGS with re-orthogonalisation"):
//   1. the panel minus its components along every finished vector (C = W B~^t D^-1; W -= C B~);
//   2. Gram matrix of the panel, L D L^t, W <- L^-1 W -- Gram-Schmidt inside the panel.
// Every vector is a row of B minus a combination of EARLIER vectors (unit lower triangular transform), and the rows come out orthogonal at
// rounding level (tests/test_gpu_gpv_scale.py: < 1e-10 relative at d = 6208, where the vector-by-vector loop of round 2 -- projection
// coefficients from the original b_j, one pass -- had lost them to 1.6e-6).  *info != 0: a Gram-Schmidt vector vanished (dependent rows).
inline hipError_t gso_blocked(hipStream_t st, double* Gt, size_t nrows, size_t d, int* d_info) {
  hipError_t e = gemm_prepare();
  if (e != hipSuccess) return e;
  double *dInv = nullptr, *dC = nullptr, *dG = nullptr, *dLi = nullptr;
  GemmWorkspace w;
  const size_t dpad = (d + GM_T - 1) / GM_T * GM_T, rpad = (nrows + GM_T - 1) / GM_T * GM_T;
  w.bytes = (size_t)64 * GM_T * GM_T * sizeof(double);                       // a single 128 x 128 tile cut 64 ways ...
  const size_t alt = (size_t)900 * GM_T * GM_T * sizeof(double) + (dpad > rpad ? dpad : rpad) * GM_T * sizeof(double);   // ... or < 384 + 512 (tile, split) pairs
  if (alt > w.bytes) w.bytes = alt;
  auto fail = [&](hipError_t err) { hipFree(dInv); hipFree(dC); hipFree(dG); hipFree(dLi); hipFree(w.ws); return err; };
  if ((e = hipMalloc(&dInv, rpad * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc(&dC, GM_T * rpad * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc(&dG, GM_T * GM_T * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc(&dLi, GM_T * GM_T * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc(&w.ws, w.bytes)) != hipSuccess) return fail(e);
  for (size_t i0 = 0; i0 < nrows; i0 += GM_T) {
    const size_t p = nrows - i0 < (size_t)GM_T ? nrows - i0 : (size_t)GM_T;
    double* W = Gt + i0 * d;
    // (project against the finished vectors, orthogonalise inside the panel) twice, in THIS order: the in-panel step can shorten a row by orders of
    // magnitude (|b| / |b~| ~ 4000 at C2), which magnifies whatever the first projection left along the finished vectors relative to the row's final
    // length; the second projection sees the short rows and removes it.
    for (int pass = 0; pass < 2; ++pass) {
      if (i0 > 0) {
        launch_gemm<true>(st, GemmArgs{W, d, Gt, d, dC, rpad, p, i0, d, 1.0, 0.0, dInv, nullptr, 0}, w);          // C = W B~^t D^-1
        launch_gemm<false>(st, GemmArgs{dC, rpad, Gt, d, W, d, p, d, i0, -1.0, 1.0, nullptr, nullptr, 0}, w);     // W -= C B~
      }
      launch_gemm<true>(st, GemmArgs{W, d, W, d, dG, GM_T, p, p, d, 1.0, 0.0, nullptr, nullptr, 0}, w);           // G = W W^t
      hipLaunchKernelGGL(k_ldl_inverse, dim3(1), dim3(256), (GM_T * (GM_T + 1) + 2 * GM_T) * sizeof(double), st, dG, (size_t)GM_T, (int)p, dLi, (size_t)GM_T, d_info);
      // W <- L^-1 W in place: the panel is ONE row tile, a workgroup reads rows 0..p-1 of its own 128 columns (K = p) and nothing else before it writes them
      launch_gemm<false>(st, GemmArgs{dLi, GM_T, W, d, W, d, p, d, p, 1.0, 0.0, nullptr, nullptr, 0}, w);
    }
    hipLaunchKernelGGL(k_rows_inv_norm2, dim3((unsigned)((p + 3) / 4)), dim3(256), 0, st, W, d, d, (int)p, dInv + i0);
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return fail(e);
}

}  // namespace psf
