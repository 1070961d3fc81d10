// psf_gpv_kernels.hpp -- HIP kernels of PSFGPV / PSFGPVRing (gpv.rs, gpv_ring.rs): short-basis assembly, Gram-Schmidt,
// and the R_q products of the ring variant.  The batched randomized nearest plane itself is in psf_np_kernels.hpp.
//
// Layout: the short basis and its Gram-Schmidt vectors are stored TRANSPOSED -- row i is basis vector i (column i of the
// reference's matrices).
#pragma once
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

// ---- Gram-Schmidt (MatQ::gso, gpv.rs:91) on the rows of St: the blocked, re-orthogonalised form lives in psf_gemm_kernels.hpp (gso_blocked);
// here only the conversion of the basis to doubles and the contract's norms.
__global__ void k_i32_to_f64(const int32_t* __restrict__ s, double* __restrict__ d, size_t total) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) d[g] = (double)s[g];
}
// ||b~_i||^2 as the contract's ascending fma chain (one thread per row; setup only)
__global__ void k_row_norm2_chain(const double* __restrict__ Gt, size_t m, double* __restrict__ norm2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double acc = 0.0;
  for (size_t j = 0; j < m; ++j) acc = fma(Gt[i * m + j], Gt[i * m + j], acc);
  norm2[i] = acc;
}

// ---- modular product shared by the polynomial kernels below ------------------------------------------------------------------
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

// The NTT forms of this product (one transform per wavefront, Montgomery arithmetic) are in psf_ntt_kernels.hpp / psf_ntt.hip.

}  // namespace psf
