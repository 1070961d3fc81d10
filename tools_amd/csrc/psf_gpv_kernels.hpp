// psf_gpv_kernels.hpp -- HIP kernels of PSFGPV / PSFGPVRing (gpv.rs, gpv_ring.rs): short-basis assembly, Gram-Schmidt,
// solve operator application and the batched randomized nearest plane (MatZ::sample_d_precomputed_gso, GPV08 SampleD).
//
// Layout: the short basis and its Gram-Schmidt vectors are stored TRANSPOSED -- row i is basis vector i (column i of the
// reference's matrices) -- so that step i of the nearest-plane walk streams two contiguous rows.
#pragma once
#include <type_traits>
#include "psf_kernels.hpp"

namespace psf {

// ---- [0 I; S' W] transposed: row c = column c of the bottom block [S' | W] (short_basis_classical.rs:77-110) ----------
// rows 0..w-1: columns of S' = I_n (x) S_k, column order reversed iff base^k == q (:80-82)
// rows w..m-1: columns of W, W[jk+t][c] = t-th digit of -A[j][c] mod q (:105-110, tag = identity)
__global__ void k_gpv_bottom_t(const uint64_t* __restrict__ A, size_t lda, uint32_t n, uint32_t k, size_t mbar, size_t w, uint64_t q,
                               uint64_t base, const int32_t* __restrict__ Sk, int reversed, int8_t* __restrict__ BT, size_t ldw) {
  const size_t m = mbar + w, total = m * ldw;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / ldw, t = g % ldw;
    int32_t v = 0;
    if (t < w) {
      const uint32_t blk = (uint32_t)(t / k), tt = (uint32_t)(t % k);
      if (i < w) {
        const size_t src = reversed ? (w - 1 - i) : i;
        if (src / k == blk) v = Sk[tt * k + (uint32_t)(src % k)];
      } else {
        const uint64_t a = A[(size_t)blk * lda + (i - w)] % q;
        uint64_t x = a ? q - a : 0;
        for (uint32_t d = 0; d < tt; ++d) x /= base;
        v = (int32_t)(x % base);
      }
    }
    BT[g] = (int8_t)v;
  }
}

// S_A^t = [ (R S')^t ; (I + R W)^t | bottom^t ]:  St[i][r] = sum_t BT[i][t] R[r][t] + [i >= w and i - w == r]  (r < mbar),
// St[i][mbar + t] = BT[i][t].  64 x 64 tiles, four coordinates per v_dot4_i32_i8.
__global__ __launch_bounds__(256) void k_gpv_basis_t(const int8_t* __restrict__ BT, size_t ldw, const int8_t* __restrict__ R, size_t ldr,
                                                     size_t m, size_t mbar, size_t w, int32_t* __restrict__ St) {
  __shared__ uint32_t sB[64][17];
  __shared__ uint32_t sR[64][17];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const size_t i0 = (size_t)blockIdx.y * 64, r0 = (size_t)blockIdx.x * 64;
  int32_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0;
  const uint32_t* B32 = reinterpret_cast<const uint32_t*>(BT);
  const uint32_t* R32 = reinterpret_cast<const uint32_t*>(R);
  const size_t w4 = (w + 3) / 4;
  for (size_t q0 = 0; q0 < w4; q0 += 16) {
    for (int e = tid; e < 64 * 16; e += 256) {
      const int rr = e >> 4, qq = e & 15;
      uint32_t vb = 0, vr = 0;
      if (q0 + qq < w4) {
        if (i0 + rr < m) vb = B32[(i0 + rr) * (ldw / 4) + q0 + qq];
        if (r0 + rr < mbar) vr = R32[(r0 + rr) * (ldr / 4) + q0 + qq];
      }
      sB[rr][qq] = vb; sR[rr][qq] = vr;
    }
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) {
      uint32_t rv[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) rv[b] = sR[tx * 4 + b][qq];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const uint32_t bv = sB[ty * 4 + a][qq];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_sdot4((int)bv, (int)rv[b], acc[a][b], false);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const size_t i = i0 + ty * 4 + a, r = r0 + tx * 4 + b;
      if (i < m && r < mbar) St[i * m + r] = acc[a][b] + ((i >= w && i - w == r) ? 1 : 0);
    }
}
__global__ void k_gpv_basis_t_tail(const int8_t* __restrict__ BT, size_t ldw, size_t m, size_t mbar, size_t w, int32_t* __restrict__ St) {
  const size_t total = m * w;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / w, t = g % w;
    St[i * m + mbar + t] = BT[i * ldw + t];
  }
}

// ---- Gram-Schmidt (MatQ::gso, gpv.rs:91) on the rows of St, right-looking: after b~_i is final, its component is removed
// from every later row, so row t receives its subtractions in ascending i -- the oracle's order.
__global__ void k_i32_to_f64(const int32_t* __restrict__ s, double* __restrict__ d, size_t total) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) d[g] = (double)s[g];
}
__device__ inline double block_sum_256(double v, double* scratch) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void k_gs_norm(const double* __restrict__ Gt, size_t m, size_t i, double* __restrict__ norm2) {
  __shared__ double scratch[4];
  double acc = 0.0;
  for (size_t j = threadIdx.x; j < m; j += 256) acc = fma(Gt[i * m + j], Gt[i * m + j], acc);
  const double s = block_sum_256(acc, scratch);
  if (threadIdx.x == 0) norm2[i] = s;
}
// mu[t] = <b_t, b~_i> / ||b~_i||^2 for t > i; one wave per row t
__global__ __launch_bounds__(256) void k_gs_project(const int32_t* __restrict__ St, const double* __restrict__ Gt, const double* __restrict__ norm2,
                                                    size_t m, size_t i, double* __restrict__ mu) {
  const size_t t = i + 1 + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= m) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (size_t j = lane; j < m; j += 64) acc = fma((double)St[t * m + j], Gt[i * m + j], acc);
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) mu[t] = acc / norm2[i];
}
// Gt[t][j] -= mu[t] Gt[i][j] for t > i
__global__ __launch_bounds__(256) void k_gs_update(double* __restrict__ Gt, const double* __restrict__ mu, size_t m, size_t i) {
  const size_t t = i + 1 + blockIdx.y;
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m || j >= m) return;
  Gt[t * m + j] = fma(-mu[t], Gt[i * m + j], Gt[t * m + j]);
}

// ||b~_i||^2 as the contract's ascending fma chain (one thread per row; setup only)
__global__ void k_row_norm2_chain(const double* __restrict__ Gt, size_t m, double* __restrict__ norm2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double acc = 0.0;
  for (size_t j = 0; j < m; ++j) acc = fma(Gt[i * m + j], Gt[i * m + j], acc);
  norm2[i] = acc;
}

// ---- solve operator: c0[b][piv[r]] = -(T u_b)[r] mod q (gpv.rs:153-158: sol = A^{-1}(u), centre = -sol); T is passed transposed ---
__device__ inline uint64_t mulmod_dev(uint64_t a, uint64_t b, uint64_t q) {
  if (q <= 0xffffffffull) return (a * b) % q;
  uint64_t r = 0;
  while (b) {
    if (b & 1) { r += a; if (r >= q) r -= q; }
    a += a; if (a >= q) a -= q;
    b >>= 1;
  }
  return r;
}
__global__ void k_gpv_solve(const uint64_t* __restrict__ Tt, const uint32_t* __restrict__ piv, size_t n, size_t m, uint64_t q, uint64_t two64,
                            const uint64_t* __restrict__ U, size_t B, int64_t* __restrict__ C0) {
  const size_t total = n * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / n, r = g % n;
    uint64_t acc = 0;
    if (q <= 0x7fffffffull) {                       // products below 2^62: sum them in 128 bits, reduce once
      Acc128 s{0, 0};
      for (size_t t = 0; t < n; ++t) {
        uint64_t uq = U[b * n + t];
        if (uq >= q) uq %= q;
        acc128_add(s, (int64_t)(Tt[t * n + r] * uq));
      }
      acc = acc128_mod(s, q, two64);
    } else {
      for (size_t t = 0; t < n; ++t) {
        acc += mulmod_dev(Tt[t * n + r], U[b * n + t] % q, q);
        if (acc >= q) acc -= q;
      }
    }
    C0[b * m + piv[r]] = -(int64_t)acc;
  }
}

// ---- randomized nearest plane (MatZ::sample_d_precomputed_gso, gpv.rs:160) -----------------------------------------------------
// One workgroup of 256 threads (4 waves, one per SIMD) walks i = dim-1..0 for 2 preimages whose integer vectors c live in
// registers: thread t owns coordinates j = t + 256 r, r < JR.  At most 256 VGPRs, so two workgroups share a CU and, having their
// own barriers, drift apart: one samples while the other projects or updates.  Per step:
//   per-thread fma chains over r (ascending) -> xor butterfly per wave -> ((w0+w1)+(w2+w3))      [the contract's orc_dot256 order]
//   c' = dot / ||b~_i||^2 ; one wave per preimage draws z with its 64 lanes evaluating attempts t = lane,
//   lane+64, ... and taking the first accepted one (= the sequential first-accept); the Philox words of the first round do not
//   depend on c': waves 2 and 3, idle while 0 and 1 sample, compute them before the barrier and pass them through LDS ; c -= z b_i.
// The rows b~_i / b_i come from packed copies (k_np_pack): row stride padded to the template's JR with zeros (no bounds checks:
// padding coordinates keep c = 0) and permuted so that a thread's coordinates r = 2a, 2a+1 (b~) / 4a..4a+3 (b) are one 16-byte
// load, a wave's loads one contiguous KiB.  b~_{i-1} is fetched while step i samples, b_{i-1} right after step i's update.
//
// Two instantiations of the same walk:
//   FP53 = true : c is kept in doubles.  The values are integers; the projection reads them without an int64 -> f64 conversion
//                 and the update is one fma per coordinate, exact while |c| < 2^52.  That is enforced by a running bound
//                 bnd_p = c0_bound + sum_i |z_i| max_j |b_i[j]|  (the same in every thread): the first step at which it could
//                 reach 2^52 the workgroup writes redo[block] = 1 and stops without output.
//   FP53 = false: c in int64, always exact.  Launched after the FP53 pass with the redo mask (workgroups whose mask is 0
//                 return at once), or alone (redo == nullptr) when q is too large for the bound to hold.
// Both produce the same bits: conversions of exactly representable integers and exact updates.
#ifdef NP_PROFILE
__device__ long long g_np_prof[8];
#define NP_T(k) do { const long long now_ = clock64(); tacc_[k] += now_ - tprev_; tprev_ = now_; } while (0)
#else
#define NP_T(k) do { } while (0)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int np_g2(int jr) { return (jr + 1) / 2; }      // 16-byte groups per thread and row of b~
__host__ __device__ constexpr int np_s4(int jr) { return (jr + 3) / 4; }      // ... of b

// packed rows: GtP[i][a][t][e] = b~_i[t + 256 (2a + e)], StP[i][a][t][e] = b_i[t + 256 (4a + e)], zero past dim
__global__ __launch_bounds__(256) void k_np_pack(const int32_t* __restrict__ St, const double* __restrict__ Gt, size_t dim, int G2, int S4,
                                                 int32_t* __restrict__ StP, double* __restrict__ GtP) {
  const size_t i = blockIdx.x;
  const size_t ng = (size_t)G2 * 512, ns = (size_t)S4 * 1024;
  for (size_t pos = threadIdx.x; pos < ng; pos += 256) {
    const size_t a = pos / 512, rem = pos % 512, t = rem / 2, e = rem & 1, j = t + 256 * (2 * a + e);
    GtP[i * ng + pos] = j < dim ? Gt[i * dim + j] : 0.0;
  }
  for (size_t pos = threadIdx.x; pos < ns; pos += 256) {
    const size_t a = pos / 1024, rem = pos % 1024, t = rem / 4, e = rem & 3, j = t + 256 * (4 * a + e);
    StP[i * ns + pos] = j < dim ? St[i * dim + j] : 0;
  }
}

// workgroups per CU by register budget: c takes 4 JR VGPRs in the FP53 pass (2 preimages x JR doubles), the rows 3 JR
template <int JR, bool FP53>
__global__ __launch_bounds__(256, JR <= 8 ? 4 : JR <= 16 ? 3 : 2) void k_gpv_nearest_plane(const int32_t* __restrict__ StP, const double* __restrict__ GtP,
                                                              const double* __restrict__ norm2, const SampleZParams* __restrict__ sz,
                                                              const double* __restrict__ rowmax, double c0_bound,
                                                              size_t dim, uint64_t seed, uint32_t tag, uint64_t first_index, size_t B,
                                                              const int64_t* __restrict__ C0, int64_t* __restrict__ E, int* __restrict__ fail,
                                                              int* __restrict__ redo) {
  using CT = typename std::conditional<FP53, double, long long>::type;
  constexpr int G2 = np_g2(JR), S4 = np_s4(JR);
  if (!FP53 && redo && !redo[blockIdx.x]) return;
  __shared__ double s_w[2][4];
  __shared__ long long s_z[2];
  __shared__ uint2 s_rng[2][64];        // (candidate word, acceptance word) of attempts 0..63 for the two preimages
  const int tid = threadIdx.x, lane = tid & 63, t = tid;
  const int wh = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t b0 = (size_t)blockIdx.x * 2;
  CT c[2][JR];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < JR; ++r) {
      const size_t j = (size_t)t + 256 * r, b = b0 + k;
      c[k][r] = (b < B && j < dim) ? (CT)C0[b * dim + j] : (CT)0;
    }
  double bnd[2] = {c0_bound, c0_bound};
  double2 g2[G2];
  int4 s4[S4];
  // buffer loads: descriptor + uniform row offset in SGPRs, per-thread byte offset t * 16 in one VGPR -- no per-lane 64-bit
  // address arithmetic.  A packed matrix is at most 8192 rows x 64 KiB = 512 MiB, inside the 32-bit offset range.
  const uint32_t toff = (uint32_t)t * 16u;
  // raw descriptors over the whole allocations (stride 0, 2^32 - 1 records, dword 3 = 0x00020000: 32-bit untyped data on the gfx9 family)
  const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(GtP), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(StP), 0, (int)0xffffffffu, 0x00020000);
  auto load_g = [&](size_t i) {
    const uint32_t row = (uint32_t)i * (uint32_t)(G2 * 4096);
#pragma unroll
    for (int a = 0; a < G2; ++a) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g, toff, row + (uint32_t)a * 4096u, 0);
      g2[a].x = __hiloint2double(v.y, v.x);
      g2[a].y = __hiloint2double(v.w, v.z);
    }
  };
  auto load_s = [&](size_t i) {
    const uint32_t row = (uint32_t)i * (uint32_t)(S4 * 4096);
#pragma unroll
    for (int a = 0; a < S4; ++a) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, toff, row + (uint32_t)a * 4096u, 0);
      s4[a] = make_int4(v.x, v.y, v.z, v.w);
    }
  };
  load_g(dim - 1);
  load_s(dim - 1);
  // waves 0 and 1 sample for preimages 0 and 1; waves 2 and 3 prepare their random words
  const bool sampler = wh < 2;
  const int ps = wh & 1;
  const bool samp_live = b0 + ps < B;
  const uint64_t index_s = first_index + b0 + ps;
  const uint32_t tw_s = tag_word(tag, index_s);
  int f = 0;
#ifdef NP_PROFILE
  long long tacc_[6] = {0, 0, 0, 0, 0, 0};
  long long tprev_ = clock64();
#endif
  for (size_t ii = dim; ii-- > 0;) {
    const double n2 = norm2[ii];
    const double rmax = FP53 ? rowmax[ii] : 0.0;
    const SampleZParams sp = sz[ii];
    if (!sampler && samp_live) {                   // words of attempts 0..63
      uint32_t wa, wb;
      sz_attempt_words(seed, (uint32_t)ii, (uint32_t)index_s, tw_s, (uint32_t)lane, sp.sh, &wa, &wb);
      s_rng[ps][lane] = make_uint2(wa, wb);
    }
    double part[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < JR; ++r) acc = fma((double)c[k][r], (r & 1) ? g2[r >> 1].y : g2[r >> 1].x, acc);
      part[k] = acc;
    }
    NP_T(0);
    part[0] = wave_xor_sum(part[0]);
    part[1] = wave_xor_sum(part[1]);
    if (lane == 0) { s_w[0][wh] = part[0]; s_w[1][wh] = part[1]; }
    NP_T(1);
    const size_t nx = ii ? ii - 1 : 0;             // the last step re-reads row 0: unconditional loads keep the wait counts exact
    load_g(nx);                                    // b~ of the next step travels while this one samples
    __syncthreads();
    NP_T(2);
    if (sampler) {
      long long z = 0;
      if (samp_live) {                             // (sampler waves only)
        const double dot = (s_w[ps][0] + s_w[ps][1]) + (s_w[ps][2] + s_w[ps][3]);
        const double cen = dot / n2;
        const SzRange rg = sz_range(cen, sp);
        bool got = false;
        uint2 wr = s_rng[ps][lane];
        for (uint32_t t0 = 0; t0 < kMaxAttempts && !got; t0 += 64) {
          const uint32_t ta = t0 + (uint32_t)lane;
          if (t0) sz_attempt_words(seed, (uint32_t)ii, (uint32_t)index_s, tw_s, ta, sp.sh, &wr.x, &wr.y);
          long long x = 0;
          const bool acc = sz_attempt(seed, (uint32_t)ii, (uint32_t)index_s, tw_s, ta, wr.x, wr.y, rg, cen, sp.inv_s, &x);
          const uint64_t mask = __ballot(acc);
          if (mask) {
            const int first = __ffsll((long long)mask) - 1;
            const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, first);
            const uint32_t xh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)x >> 32), first);
            z = (long long)(((unsigned long long)xh << 32) | xl);
            got = true;
          }
        }
        if (!got) { f = 1; z = (long long)floor(cen + 0.5); }
      }
      if (lane == 0) s_z[ps] = z;
    }
    NP_T(3);
    __syncthreads();
    NP_T(4);
    const long long za = s_z[0], zb = s_z[1];
    if (FP53) {
      bnd[0] = fma(fabs((double)za), rmax, bnd[0]);
      bnd[1] = fma(fabs((double)zb), rmax, bnd[1]);
      const bool over = !(bnd[0] < 0x1.0p52) || !(bnd[1] < 0x1.0p52);
      if (over) {                                  // the same decision in every thread: hand the workgroup to the int64 pass
        if (tid == 0) redo[blockIdx.x] = 1;
        return;
      }
      const double nza = -(double)za, nzb = -(double)zb;
#pragma unroll
      for (int r = 0; r < JR; ++r) {
        const int4 v = s4[r >> 2];
        const double sd = (double)((r & 3) == 0 ? v.x : (r & 3) == 1 ? v.y : (r & 3) == 2 ? v.z : v.w);
        c[0][r] = (CT)fma(nza, sd, (double)c[0][r]);
        c[1][r] = (CT)fma(nzb, sd, (double)c[1][r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < JR; ++r) {
        const int4 v = s4[r >> 2];
        const long long sv = (long long)((r & 3) == 0 ? v.x : (r & 3) == 1 ? v.y : (r & 3) == 2 ? v.z : v.w);
        c[0][r] -= (CT)(za * sv);
        c[1][r] -= (CT)(zb * sv);
      }
    }
    load_s(nx);                                    // b of the next step: not needed before its update
    NP_T(5);
  }
#ifdef NP_PROFILE
  if (blockIdx.x == 0 && tid == 0) for (int k = 0; k < 6; ++k) g_np_prof[k] += tacc_[k];
  if (FP53 && tid == 0) { const double mb = bnd[0] > bnd[1] ? bnd[0] : bnd[1]; atomicMax((unsigned long long*)&g_np_prof[6], (unsigned long long)__double_as_longlong(mb)); }
#endif
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < JR; ++r) {
      const size_t j = (size_t)t + 256 * r, b = b0 + k;
      if (b < B && j < dim) E[b * dim + j] = -(int64_t)c[k][r];
    }
  if (f) atomicOr(fail, 1);
}

// max_j |b_i[j]| per basis row (one wave per row), for the FP53 bound above
__global__ __launch_bounds__(64) void k_row_absmax_i32(const int32_t* __restrict__ St, size_t dim, double* __restrict__ rowmax) {
  const size_t i = blockIdx.x;
  uint32_t mx = 0;
  for (size_t j = threadIdx.x; j < dim; j += 64) {
    const int32_t v = St[i * dim + j];
    const uint32_t a = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
    mx = a > mx ? a : mx;
  }
  for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, off); mx = o > mx ? o : mx; }
  if (threadIdx.x == 0) rowmax[i] = (double)mx;
}

// ---- R_q = Z_q[X]/(X^n + 1): negacyclic product (PolynomialRingZq multiplication under gadget_ring.rs:78 and gpv_ring.rs:245-246) ----
// One workgroup per pair; both operands in LDS; thread t owns coefficients t, t+256, ...  out[c] = sum_{i<=c} a_i b_{c-i} - sum_{i>c} a_i b_{n+c-i}.
// Exact: positive and negative parts are accumulated in 128 bits (q < 2^31) or reduced term by term (larger q).
__global__ __launch_bounds__(256) void k_polymul_negacyclic(uint64_t q, uint64_t two64, uint32_t n, const uint64_t* __restrict__ A, size_t a_stride,
                                                            const int64_t* __restrict__ Bp, size_t b_stride, uint64_t* __restrict__ out, size_t o_stride) {
  extern __shared__ __attribute__((aligned(16))) uint64_t pm_smem[];   // a[n] | b[n]
  uint64_t* sa = pm_smem;
  uint64_t* sb = pm_smem + n;
  const size_t pair = blockIdx.x;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    sa[i] = A[pair * a_stride + i] % q;
    const int64_t v = Bp[pair * b_stride + i] % (int64_t)q;
    sb[i] = (uint64_t)(v < 0 ? v + (int64_t)q : v);
  }
  __syncthreads();
  const bool small = q <= 0x7fffffffull;
  for (uint32_t c = threadIdx.x; c < n; c += 256) {
    uint64_t pos = 0, neg = 0;
    if (small) {
      Acc128 P{0, 0}, N{0, 0};
      for (uint32_t i = 0; i <= c; ++i) acc128_add(P, (int64_t)(sa[i] * sb[c - i]));
      for (uint32_t i = c + 1; i < n; ++i) acc128_add(N, (int64_t)(sa[i] * sb[n + c - i]));
      pos = acc128_mod(P, q, two64);
      neg = acc128_mod(N, q, two64);
    } else {
      for (uint32_t i = 0; i <= c; ++i) { pos += mulmod_dev(sa[i], sb[c - i], q); if (pos >= q) pos -= q; }
      for (uint32_t i = c + 1; i < n; ++i) { neg += mulmod_dev(sa[i], sb[n + c - i], q); if (neg >= q) neg -= q; }
    }
    out[pair * o_stride + c] = pos >= neg ? pos - neg : pos + q - neg;
  }
}

// ---- the same product through an (incomplete) negacyclic NTT, for NTT-friendly prime q < 2^31 -------------------------------------
// X^n + 1 splits over Z_q into 2^L factors X^d - gamma_i (L = min(v2(q-1) - 1, log2 n), d = n >> L; q = 3329, n = 256 gives Kyber's
// L = 7, d = 2).  Forward transform: L Cooley-Tukey levels in LDS with zetas in bit-reversed order; leaf products in
// Z_q[X]/(X^d - gamma_i) with gamma_i = zetas[2^(L-1) + i/2]^2 ... computed as +-zeta_leaf; inverse: Gentleman-Sande + 2^-L.
// Exact arithmetic mod q, so the result equals k_polymul_negacyclic bit for bit (tested).
__global__ __launch_bounds__(256) void k_polymul_ntt(uint32_t q32, uint32_t n, uint32_t L, uint32_t d, const uint64_t* __restrict__ zetas,
                                                     const uint64_t* __restrict__ zetas_inv, uint64_t inv_scale, const uint64_t* __restrict__ A,
                                                     const int64_t* __restrict__ Bp, uint64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint64_t nt_smem[];   // a[n] | b[n] | c[n]
  uint32_t* sa = reinterpret_cast<uint32_t*>(nt_smem);
  uint32_t* sb = sa + n;
  uint32_t* sc = sb + n;
  const uint64_t q = q32;
  const size_t pair = blockIdx.x;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    sa[i] = (uint32_t)(A[pair * n + i] % q);
    const int64_t v = Bp[pair * n + i] % (int64_t)q;
    sb[i] = (uint32_t)(v < 0 ? v + (int64_t)q : v);
  }
  __syncthreads();
  // forward: level l has 2^l blocks of length 2*len, block b uses zetas[2^l + b]
  uint32_t len = n >> 1;
  for (uint32_t l = 0; l < L; ++l, len >>= 1) {
    for (uint32_t e = threadIdx.x; e < n / 2; e += 256) {
      const uint32_t blk = e / len, j = e % len;
      const uint32_t lo = blk * 2 * len + j, hi = lo + len;
      const uint64_t z = zetas[(1u << l) + blk];
      const uint32_t ta = (uint32_t)((z * sa[hi]) % q), tb = (uint32_t)((z * sb[hi]) % q);
      const uint32_t ua = sa[lo], ub = sb[lo];
      sa[hi] = ua >= ta ? ua - ta : ua + q32 - ta;  sa[lo] = (uint32_t)(((uint64_t)ua + ta) % q);
      sb[hi] = ub >= tb ? ub - tb : ub + q32 - tb;  sb[lo] = (uint32_t)(((uint64_t)ub + tb) % q);
    }
    __syncthreads();
  }
  // leaves: 2^L rings Z_q[X]/(X^d - gamma), leaf i occupies coefficients [i d, (i+1) d); consecutive leaves are the +- pair of the last level
  for (uint32_t e = threadIdx.x; e < n; e += 256) {
    const uint32_t leaf = e / d, c = e % d;
    uint64_t gamma;
    if (L == 0) gamma = q - 1;                                     // X^n + 1 itself: X^d = -1
    else { const uint64_t z = zetas[(1u << (L - 1)) + (leaf >> 1)]; gamma = (leaf & 1) ? q - z : z; }
    const uint32_t* pa = sa + leaf * d;
    const uint32_t* pb = sb + leaf * d;
    uint64_t lo = 0, hi = 0;                                       // c-th coefficient: sum_{i+j=c} + gamma * sum_{i+j=c+d}
    for (uint32_t i = 0; i < d; ++i) {
      if (i <= c) lo = (lo + (uint64_t)pa[i] * pb[c - i]) % q;
      else hi = (hi + (uint64_t)pa[i] * pb[d + c - i]) % q;
    }
    sc[e] = (uint32_t)((lo + (gamma * hi) % q) % q);
  }
  __syncthreads();
  // inverse: Gentleman-Sande, levels in reverse order
  len = d;
  for (int l = (int)L - 1; l >= 0; --l, len <<= 1) {
    for (uint32_t e = threadIdx.x; e < n / 2; e += 256) {
      const uint32_t blk = e / len, j = e % len;
      const uint32_t lo = blk * 2 * len + j, hi = lo + len;
      const uint64_t zi = zetas_inv[(1u << l) + blk];
      const uint32_t u = sc[lo], v = sc[hi];
      sc[lo] = (uint32_t)(((uint64_t)u + v) % q);
      const uint32_t dif = u >= v ? u - v : u + q32 - v;
      sc[hi] = (uint32_t)((zi * dif) % q);
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < n; i += 256) out[pair * n + i] = (inv_scale * sc[i]) % q;
}

}  // namespace psf
