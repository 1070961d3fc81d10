// psf_np_kernels.hpp -- the randomized nearest plane of PSFGPV / PSFGPVRing (MatZ::sample_d_precomputed_gso, gpv.rs:160;
// MatPolyOverZ::sample_d after the coefficient embedding, gpv_ring.rs:204-211) in its batched, blocked form.
//
// GPV08 SampleD walks i = d-1 .. 0:  c' = <c, b~_i> / ||b~_i||^2 ;  z_i <- D_{Z, s/||b~_i||, c'} ;  c -= z_i b_i.
// With g[j][i] = <b_j, b~_i> (j > i; precomputed once per key) the projection at step i is
//     <c0, b~_i> - sum_{j > i} z_j g[j][i],
// so the integer vector c never has to be carried along: the walk needs the d x B matrix T of running projections, and the
// preimage is recovered at the end as e = -(c0 - sum_i z_i b_i).  Rows are cut into blocks of NP_NB = 64 indices.  Per call:
//     T        = B~[:, pivots] C0[pivots]               k_np_project   FP64 MFMA, K = n (c0 = -sol lives on the n pivot columns)
//     for J descending:
//        Z_J   <- rows of block J take the contribution of Z_(J+1), then the 64 steps of the block are sampled
//                                                       np_sample_body<G>     one wave per G preimages, 64/G lanes evaluate the
//                                                                          SampleZ attempts of one draw in parallel
//        T[< J-1] -= G[< J-1, J] Z_J                    np_update_tile     FP64 MFMA, K = 64 per block, every operand read once per
//                                                                          batch; extra workgroups of the NEXT block's launch
//                                                                          (k_np_step), near rows per block, far rows per panel
//     E        = Z^t B  (+ sol on the pivot columns)    k_np_combine8      int8 MFMA on balanced base-256 digit planes (exact)
// Basis and Gram-Schmidt data are therefore read once per BATCH and block, not once per pair of preimages, nothing is held in
// registers across steps except the 64 running projections of the current block, and the lattice dimension is not limited
// by the register file.  The floating-point evaluation order is part of the library's contract ("blocked
// nearest plane"); every kernel below follows it bit for bit: v_mfma_f64_16x16x4_f64 is an ascending-k fma chain
// (profiles/r01_probe_mfma_f64.log), so a block's contribution t = fma(-z_j, g[j][.], t), j ascending, is one MFMA K loop with T as the accumulator.
#pragma once
#include <type_traits>
#include "psf_kernels.hpp"

namespace psf {

constexpr int NP_NB = 64;

// chunks of the bulk panels G[< 64 J, block J] in front of block J: row-blocks of 128 rows x 4 chunks (K = 64)
__host__ __device__ inline size_t np_panel_rowblocks(size_t J) { return (J * NP_NB + 127) / 128; }
__host__ __device__ inline size_t np_panel_base(size_t J) { return 4 * ((J * J) / 4); }      // = 4 sum_{J' < J} ceil(J' / 2)

// ---- per key ---------------------------------------------------------------------------------------------------------
// Gd[j][i] = <b_j, b~_i> for j > i (0 elsewhere): per output one ascending fma chain over the coordinates, from +0.
// 64 x 64 tile per workgroup, 4 x 4 outputs per thread, coordinates staged 16 at a time.
__global__ __launch_bounds__(256) void k_np_gram(const int32_t* __restrict__ St, const double* __restrict__ Gt, size_t d, double* __restrict__ Gd) {
  __shared__ double sB[16][65];
  __shared__ double sG[16][65];
  const size_t j0 = (size_t)blockIdx.y * 64, i0 = (size_t)blockIdx.x * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  if (i0 > j0 + 63) {      // strictly upper tile: zeros
    for (int e = tid; e < 64 * 64; e += 256) {
      const size_t j = j0 + (e >> 6), i = i0 + (e & 63);
      if (j < d && i < d) Gd[j * d + i] = 0.0;
    }
    return;
  }
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (size_t t0 = 0; t0 < d; t0 += 16) {
    for (int e = tid; e < 64 * 16; e += 256) {
      const int r = e >> 4, tt = e & 15;
      const size_t t = t0 + tt;
      sB[tt][r] = (j0 + r < d && t < d) ? (double)St[(j0 + r) * d + t] : 0.0;
      sG[tt][r] = (i0 + r < d && t < d) ? Gt[(i0 + r) * d + t] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      double bv[4], gv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { bv[a] = sB[tt][ty * 4 + a]; gv[a] = sG[tt][tx * 4 + a]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(bv[a], gv[b], acc[a][b]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const size_t j = j0 + ty * 4 + a, i = i0 + tx * 4 + b;
      if (j < d && i < d) Gd[j * d + i] = j > i ? acc[a][b] : 0.0;
    }
}

// bulk panels in MFMA-fragment order (A operand of np_update_tile): chunk np_panel_base(J) + 4 rb + kc holds
// G[row i = 128 rb + r][k = 64 J + 16 kc + kk] at tr_chunk_pos(r, kk); rows i >= 64 J (the block itself and above) are zero
__global__ void k_np_pack_panels(const double* __restrict__ Gd, size_t d, size_t nblk, double* __restrict__ Gp) {
  const size_t total = np_panel_base(nblk) * TR_CHUNK;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t chunk = g / TR_CHUNK;
    const int pos = (int)(g % TR_CHUNK);
    size_t J = (size_t)(2.0 * sqrt((double)(chunk / 4)));
    while (np_panel_base(J + 1) <= chunk) ++J;
    while (np_panel_base(J) > chunk) --J;
    const size_t rel = chunk - np_panel_base(J), rb = rel / 4, kc = rel % 4;
    const int ks = pos >> 9, tile = (pos >> 6) & 7, ln = pos & 63;
    const size_t i = rb * 128 + tile * 16 + (ln & 15);
    const size_t j = J * NP_NB + kc * 16 + ks * 4 + (ln >> 4);
    Gp[g] = (i < J * NP_NB && j < d) ? Gd[j * d + i] : 0.0;
  }
}
// in-block triangles, packed: Gin[J][l (l - 1) / 2 + li] = g[64 J + l][64 J + li] for li < l  (2016 entries, padded to NP_TRI = 2048)
constexpr int NP_TRI = 2048;
__global__ void k_np_pack_inblock(const double* __restrict__ Gd, size_t d, size_t nblk, double* __restrict__ Gin) {
  const size_t total = nblk * NP_TRI;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t J = g / NP_TRI, e = g % NP_TRI;
    double v = 0.0;
    if (e < (size_t)NP_NB * (NP_NB - 1) / 2) {
      size_t l = (size_t)((1.0 + sqrt(1.0 + 8.0 * (double)e)) * 0.5);
      while (l * (l - 1) / 2 > e) --l;
      while ((l + 1) * l / 2 <= e) ++l;
      const size_t li = e - l * (l - 1) / 2;
      const size_t j = J * NP_NB + l, i = J * NP_NB + li;
      if (j < d) v = Gd[j * d + i];
    }
    Gin[g] = v;
  }
}
// panel between neighbouring blocks, applied inside the sampler: Gnx[J][k][li] = g[64 (J+1) + k][64 J + li] (contribution of block J+1 to block J)
__global__ void k_np_pack_next(const double* __restrict__ Gd, size_t d, size_t nblk, double* __restrict__ Gnx) {
  const size_t total = nblk * NP_NB * NP_NB;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t J = g / (NP_NB * NP_NB), k = (g / NP_NB) % NP_NB, li = g % NP_NB;
    const size_t j = (J + 1) * NP_NB + k, i = J * NP_NB + li;
    Gnx[g] = (j < d && i < d) ? Gd[j * d + i] : 0.0;
  }
}
// A operand of the initial projection: chunk (rb, kc) holds b~_i[piv[16 kc + kk]] for i = 128 rb + r, zero padded (piv == nullptr: every coordinate, n = d)
__global__ void k_np_pack_bpiv(const double* __restrict__ Gt, const uint32_t* __restrict__ piv, size_t d, size_t n, size_t nrb, size_t nkc,
                               double* __restrict__ Bp) {
  const size_t total = nrb * nkc * TR_CHUNK;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t chunk = g / TR_CHUNK, rb = chunk / nkc, kc = chunk % nkc;
    const int pos = (int)(g % TR_CHUNK);
    const int ks = pos >> 9, tile = (pos >> 6) & 7, ln = pos & 63;
    const size_t i = rb * 128 + tile * 16 + (ln & 15), r = kc * 16 + ks * 4 + (ln >> 4);
    Bp[g] = (i < d && r < n) ? Gt[i * d + (piv ? piv[r] : r)] : 0.0;
  }
}
// balanced base-256 digit planes of the basis, transposed for the recombination: B8[plane][j][i] (row = coordinate j, K = step i),
// rows and K padded to 128 with zeros.  *too_big is raised when an entry does not fit two digits (|v| > 32639).
__global__ void k_np_pack_basis8(const int32_t* __restrict__ St, size_t d, size_t dpad, int8_t* __restrict__ B8, int* __restrict__ info /*[0] too big, [1] hi plane used*/) {
  const size_t total = dpad * dpad;
  int big = 0, hi_used = 0;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t j = g / dpad, i = g % dpad;
    const int32_t v = (j < d && i < d) ? St[i * d + j] : 0;
    if (v > 32639 || v < -32639) big = 1;
    const int32_t lo = (int32_t)(int8_t)(v & 0xff);
    const int32_t hi = (v - lo) >> 8;
    if (hi) hi_used = 1;
    B8[g] = (int8_t)lo;
    B8[total + g] = (int8_t)hi;
  }
  if (big) atomicOr(info, 1);
  if (hi_used) atomicOr(info + 1, 1);
}

// ---- per call ----------------------------------------------------------------------------------------------------------
// sol = A^{-1}(u) on the pivot columns (gpv.rs:153-156: the factored elimination, T passed transposed), written twice:
//   Sol[r][b]           the residues, added back to the pivot columns of e at the end (gpv.rs:160: sol + sample)
//   C0p chunk (bj, kc)  (double)(-sol): the centre c0 = -sol (gpv.rs:158) as the B operand of the initial projection
__global__ void k_np_solve(const uint64_t* __restrict__ Tt, size_t n, size_t nk16, uint64_t q, uint64_t two64, const uint64_t* __restrict__ U,
                           size_t B, size_t ld, uint64_t* __restrict__ Sol, double* __restrict__ C0p, size_t cols) {
  const size_t total = nk16 * cols;    // cols: B rounded up to whole 128-column blocks at least (the padding columns of the operand are rewritten as zeros)
  const size_t nkc = nk16 / 16;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t r = g % nk16, b = g / nk16;
    uint64_t acc = 0;
    if (r < n && b < B) {
      if (q < (1ull << 24) && n <= 65536) {         // products below 2^48, the sum below 2^64: one v_mad_u64_u32 per term, reduced once (C2 / C4: q = 3329)
        const uint32_t* T32 = reinterpret_cast<const uint32_t*>(Tt);
        uint64_t s0 = 0, s1 = 0;
        size_t t = 0;
        for (; t + 2 <= n; t += 2) {
          uint64_t u0 = U[b * n + t], u1 = U[b * n + t + 1];
          if (u0 >= q) u0 %= q;
          if (u1 >= q) u1 %= q;
          s0 += (uint64_t)T32[2 * (t * n + r)] * (uint32_t)u0;
          s1 += (uint64_t)T32[2 * ((t + 1) * n + r)] * (uint32_t)u1;
        }
        if (t < n) {
          uint64_t u0 = U[b * n + t];
          if (u0 >= q) u0 %= q;
          s0 += (uint64_t)T32[2 * (t * n + r)] * (uint32_t)u0;
        }
        acc = (s0 % q + s1 % q) % q;
      } else if (q <= 0x7fffffffull) {              // products below 2^62: sum them in 128 bits, reduce once
        Acc128 s{0, 0};
        for (size_t t = 0; t < n; ++t) {
          uint64_t uq = U[b * n + t];
          if (uq >= q) uq %= q;
          acc128_add(s, (int64_t)(Tt[t * n + r] * uq));
        }
        acc = acc128_mod(s, q, two64);
      } else {
        for (size_t t = 0; t < n; ++t) {
          const uint64_t a = Tt[t * n + r], bb = U[b * n + t] % q;
          const uint64_t lo = a * bb, hi = __umul64hi(a, bb);
          acc += acc128_mod(Acc128{lo, (int64_t)hi}, q, two64);
          if (acc >= q) acc -= q;
        }
      }
      Sol[r * ld + b] = acc;
    }
    const size_t chunk = (b / TR_BN) * nkc + r / 16;
    C0p[chunk * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(r % 16))] = -(double)acc;
  }
}

// The same for q < 2^24 (C2 / C4: q = 3329) as a register-tiled integer product: a 64 (pivot rows) x 64 (preimages) tile per workgroup, 4 x 4 outputs per thread, K in
// chunks of 32 through LDS.  k_np_solve above walks n terms per OUTPUT with two dependent loads per term: 227 us for C4's 256 x 256 operator and 4096 targets
// (2.7 x 10^8 multiply-adds: the whole chip for a quarter of a millisecond); this form takes the loads off the multiply-add chain.  ACC32: n (q - 1)^2 < 2^32 -- the
// sum of a row fits 32 bits and a term is ONE full-rate v_mad_u32_u24; otherwise 64-bit sums (products below 2^48, n <= 65536 terms).  Same residues.
template <bool ACC32>
__global__ __launch_bounds__(256) void k_np_solve_tiled(const uint64_t* __restrict__ Tt, size_t n, size_t nk16, uint64_t q, const uint64_t* __restrict__ U, size_t B, size_t ld,
                                                        uint64_t* __restrict__ Sol, double* __restrict__ C0p, size_t cols) {
  __shared__ uint32_t sT[32][64 + 1];
  __shared__ uint32_t sU[32][64 + 1];
  const size_t b0 = (size_t)blockIdx.x * 64, r0 = (size_t)blockIdx.y * 64;
  const int tid = threadIdx.x, tb = tid & 15, tr = tid >> 4;
  const size_t nkc = nk16 / 16;
  typedef typename std::conditional<ACC32, uint32_t, uint64_t>::type acc_t;
  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  const uint32_t q32 = (uint32_t)q;
  for (size_t t0 = 0; t0 < n; t0 += 32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = e * 256 + tid;
      {  // operator: Tt[t][r], consecutive r
        const int tt = idx >> 6, rr = idx & 63;
        const size_t t = t0 + tt, r = r0 + rr;
        sT[tt][rr] = (t < n && r < n) ? (uint32_t)Tt[t * n + r] : 0u;
      }
      {  // targets: U[b][t], consecutive t
        const int bb = idx >> 5, tt = idx & 31;
        const size_t t = t0 + tt, b = b0 + bb;
        uint64_t u = (t < n && b < B) ? U[b * n + t] : 0;
        if (u >= q) u %= q;
        sU[tt][bb] = (uint32_t)u;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int tt = 0; tt < 32; ++tt) {
      uint32_t a[4], u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sT[tt][tr + 16 * i]; u[i] = sU[tt][tb + 16 * i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ACC32) acc[i][j] = (acc_t)__umul24(a[i], u[j]) + acc[i][j];
          else acc[i][j] += (acc_t)((uint64_t)a[i] * u[j]);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t r = r0 + tr + 16 * i, b = b0 + tb + 16 * j;
      if (r >= nk16 || b >= cols) continue;
      const uint64_t v = ACC32 ? (uint64_t)((uint32_t)acc[i][j] % q32) : (uint64_t)acc[i][j] % q;
      if (r < n && b < B) Sol[r * ld + b] = v;
      const size_t chunk = (b / TR_BN) * nkc + r / 16;
      C0p[chunk * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(r % 16))] = -(double)((r < n && b < B) ? v : 0ull);
    }
}

// Initial projection T = A B on the FP64 matrix cores: 128 x 128 tile per workgroup, wave tile 64 x 64, K chunks of 16 staged by
// LDS-DMA exactly as in k_trmm_f64 (both operands are fragment-ordered chunk streams).  The per-block updates of the walk are
// np_update_tile below.
// only_if: nullptr, or a device word -- the launch does nothing while it is 0 (the second projection of a walk that gave up, psfgpv_impl.hpp)
__global__ __launch_bounds__(256, 2) void k_np_project(const double* __restrict__ Ach, size_t a_rb_stride, const double* __restrict__ Bch, size_t b_bj_stride,
                                                       int nk, double* __restrict__ T, size_t ldt, const unsigned* __restrict__ only_if = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (only_if && __hip_atomic_load(only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const double* gA = Ach + (size_t)bi * a_rb_stride * TR_CHUNK + lane * 2;
  const double* gB = Bch + (size_t)bj * b_bj_stride * TR_CHUNK + lane * 2;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  auto stage_load = [&](int kb, int buf) {
    const double* ga = gA + (size_t)kb * TR_CHUNK;
    const double* gb = gB + (size_t)kb * TR_CHUNK;
    double* la = smem + buf * (2 * TR_CHUNK);
#pragma unroll
    for (int i = 0; i < TR_CHUNK / 512; ++i) {
      const int piece = wave * (TR_CHUNK / 512) + i;
      __builtin_amdgcn_global_load_lds(ga + piece * 128, (lds_void_ptr)(la + piece * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gb + piece * 128, (lds_void_ptr)(la + TR_CHUNK + piece * 128), 16, 0, 0);
    }
  };
  stage_load(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kb = 0; kb < nk; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nk) stage_load(kb + 1, cur ^ 1);
    const double* sA = smem + cur * (2 * TR_CHUNK);
    const double* sB = sA + TR_CHUNK;
#pragma unroll
    for (int ks = 0; ks < TR_BK / 4; ++ks) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = sA[(ks * 8 + wr * 4 + i) * 64 + lane];
        b[i] = sB[(ks * 8 + wc * 4 + i) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const size_t row0 = (size_t)bi * TR_BM + wr * 64, col0 = (size_t)bj * TR_BN + wc * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T[(row0 + i * 16 + (lane >> 4) + 4 * r) * ldt + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
      }
}

// The 64 steps of one block for G preimages per wave (LPD = 64 / G lanes per draw).  Lane (sg, lam) of a wave belongs to preimage
// sg and holds the running projections t of the rows slot * LPD + lam of the block.  Step l (descending): the owner's t is broadcast,
// c' = t / ||b~||^2 (as a product with the stored reciprocal), the LPD lanes evaluate attempts lam, lam + LPD, ... of the draw
// (coordinate i, preimage index) and the first accepted attempt in attempt order wins -- the value of the sequential sampler.
// Fast path (narrow SampleZ words, |c'| < 2^30 -- every draw of C2 / C4):
//   * the Philox blocks do not depend on c': once per four steps lane (s, g) computes block g of step l - s (a block serves the
//     attempts 4g .. 4g+3), the words go through a private LDS strip and every lane picks the word of its own attempt;
//   * per-row constants come from a 32-byte LDS record, the in-block triangle row g[l][.] from LDS;
//   * the fp32 screen (sz_screen16) classifies each attempt as rejected / certainly accepted / to be settled; only a "to be
//     settled" attempt that precedes every certain accept pays the exact f64 decision (sz_decide).
// Anything else (huge or integral centres, no accept among the fast attempts) runs the generic rounds: per-lane Philox,
// sz_maybe / sz_decide.  Either way the outcome is that of the sequential sampler on the same Philox streams.
// G == 2 (32 lanes per draw): the helper prepares TWO sets of 32 attempts per step (packed 8-byte records, same ring bytes); the sampler reads and screens
// the second set only for a draw that is still open after the first ((11/12)^32 = 6 % of the draws have no candidate there), so 64 attempts are covered by the
// fast screen as with G == 1.  Before, those draws went through the generic rounds (~3 k cycles each) and, two preimages per wave, 12 % of the wave-steps did:
// at C4 39 % of a sampler wave's time and most of the launch's slowest-wave tail (profiles/r03_notes.md): 4.88 -> 4.59 ms.
// Then the rows below in the block take t' = fma(-z, g[l][.], t').  Outputs: z as f64 in the operand layout of the update
// product, and as three balanced base-256 int8 digit planes [i/16][b][16] for the recombination.
#ifdef NP_PROFILE   /* cycle breakdown of k_np_sample (workgroup 0, wave 0): tools/np_profile.py */
__device__ long long g_np_prof[8];
__device__ unsigned long long g_np_events[4];      /* over all sampler waves: steps, steps that entered the settle loop, steps that entered the generic rounds, steps with a special centre */
#if NP_PROFILE == 4   /* event counts only (the atomics distort every timing) */
#define NP_EVENT(k) do { if (lane == 0) atomicAdd(&g_np_events[k], 1ull); } while (0)
#else
#define NP_EVENT(k) do { } while (0)
#endif
__device__ unsigned long long g_np_sum[8], g_np_max[8];      /* NP_PROFILE == 3: per bucket, summed / maximised over ALL sampler waves (per wave and launch) */
#if NP_PROFILE == 3   /* prologue | chain up to the ballots | settle | generic rounds | update | epilogue, every sampler wave */
__device__ unsigned long long g_np_single[8];    /* [0] largest single settle phase, [1] largest single generic phase, [2] settle phases > 4096 ticks, [3] generic phases > 4096 ticks, [4] largest single chain phase, [5] chain phases > 4096 */
#define NP_T(k) do { if ((k) == 0 || (k) >= 3) { const long long now_ = (long long)__builtin_readcyclecounter(); const long long d_ = now_ - tprev_; tacc_[k] += d_; tprev_ = now_; \
    if ((k) == 4) { if (d_ > smax_[0]) smax_[0] = d_; if (d_ > 4096) ++smax_[2]; } if ((k) == 7) { if (d_ > smax_[1]) smax_[1] = d_; if (d_ > 4096) ++smax_[3]; } \
    if ((k) == 3) { if (d_ > smax_[4]) smax_[4] = d_; if (d_ > 4096) ++smax_[5]; } } } while (0)
#elif NP_PROFILE == 2   /* only the prologue / steps / epilogue split: no stamps inside the step loop */
#define NP_T(k) do { if ((k) == 0 || (k) == 6) { const long long now_ = (long long)__builtin_readcyclecounter(); tacc_[k] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define NP_T(k) do { const long long now_ = (long long)__builtin_readcyclecounter(); tacc_[k] += now_ - tprev_; tprev_ = now_; } while (0)
#endif
#else
#define NP_T(k) do { } while (0)
#define NP_EVENT(k) do { } while (0)
#endif

struct NpRow { double inv_n2; float inv_sk; int32_t c6; uint32_t n_int, thr_int, thr_frac, sh; };   // 32 bytes per row; sh = 0: no fast path;
                                                                                                      // inv_sk = sqrt(pi log2 e) / s': exp(-pi a^2) = exp2(-(a inv_sk s')^2)

struct NpSampleArgs {
  const double* T; size_t ldt;            // running projections, d_pad x ld
  const double* Gin; const double* Gnx; const NpRow* rows; const SampleZParams* sz;
  double* Zf; size_t nkb;                 // chunk stream (bj, kb)
  int8_t* Z8; size_t zplane; size_t ld;   // three digit planes, zplane bytes apart
  int* flags;                             // [0] sampler failure, [1] a second digit is in use, [2] a third, [3] |z| beyond three digits
};

// fp32 screen of the attempts (the narrow form follows sz_screen16 of psf_rng.hpp).  An attempt is classified as certainly
// rejected / certainly accepted ("sure") / inside the +-0.1 % band around the threshold (settled by sz_decide).  The candidate index is
// exact in fp32 (N < 2^24), a = (idx + (lo - c)) / s carries an error below 1e-6, which moves rho by < 1e-4 relative; (float)wb is
// within 2^-24 relative.  sure means (wb + 1) 2^-sh <= 0.999 rho_f < rho, hence wb < floor(rho 2^sh): no tie, the exact rule accepts.
// The screen only decides who pays for the exact rule; the accepted attempt and its value are always those of the exact sampler.
__device__ inline size_t j0_of(size_t J) { return J * NP_NB; }

// Two waves per G preimages.  The HELPER wave runs ahead and produces, for every step and attempt, what does not depend on the
// centre: the Philox words (once per four steps lane (s, g) computes block g of step l - s; a block serves four narrow or two wide
// attempts; the words cross a private LDS strip), the candidate index for a non-integral centre, its Lemire test, idx / s' and the
// scaled acceptance word -- a 16-byte record per (step, attempt) in an LDS ring of two groups of four steps.  The SAMPLER wave
// owns the dependent chain and nothing else: broadcast t, c' = t / ||b~||^2, ceil, c_rel, one fma + exp2 per lane, two compares,
// first candidate by s_ff1, z, and the fma that updates the rows below.  The waves meet through two LDS counters (groups produced /
// consumed); all eight waves of a workgroup are resident by construction, so the spin-waits always make progress.
// COH (unused since round 4, when the one-launch walk k_np_walk was removed: it spilled and bought 3 %, profiles/r03_notes.md): the running projections read and
// the drawn z written with agent-scope accesses, for a launch in which they come from / go to workgroups on other XCDs
// WALK (k_np_walk: one launch for the whole walk): the z of the block above are the wave's own draws of the previous call of this function (zr_io, in
// registers) instead of a read-back from memory, and they are handed on the same way.
// BG (k_np_walk2): what a helper wave does instead of sleeping while its record ring is full (bg.step(): one unit of background work, false = nothing to do) and after
// its last group (bg.drain()); t_wait: a sampler wave spins on this word (then reads T) -- the block's rows are handed over by another workgroup.
struct NpNoBg {
  __device__ __forceinline__ bool step() { return false; }
  __device__ __forceinline__ void drain() {}
  __device__ __forceinline__ bool wait_word(const unsigned*) { return true; }
};
template <int G, bool COH = false, bool WALK = false, class BG = NpNoBg>
__device__ __forceinline__ void np_sample_body(unsigned char* smem_raw, unsigned wg, const NpSampleArgs& a, size_t dim, size_t J, uint64_t seed, uint32_t tag,
                                               uint64_t first_index, size_t B, long long* zr_io = nullptr, BG* bg = nullptr, const unsigned* t_wait = nullptr) {
  constexpr int LPD = 64 / G, BPS = LPD / 4;
  static_assert(G == 1 || G == 2, "one or two preimages per wave pair");
  // carved from the launch's dynamic LDS (shared with the update tiles of the same launch): 16 + 2 + 8 + 32 + 4 KiB + counters < 64 KiB
  double* s_tri = reinterpret_cast<double*>(smem_raw);                                      // in-block triangle, packed, NP_TRI
  NpRow* s_row = reinterpret_cast<NpRow*>(smem_raw + 16384);                                // NP_NB
  uint2 (*s_words)[256] = reinterpret_cast<uint2 (*)[256]>(smem_raw + 16384 + 2048);        // helper: (candidate word, acceptance word) of [sg][step][attempt]
  uint4 (*s_ring)[8 * 64] = reinterpret_cast<uint4 (*)[8 * 64]>(smem_raw + 16384 + 2048 + 8192);         // per pair: records of two groups of four steps
  double (*s_zs)[128] = reinterpret_cast<double (*)[128]>(smem_raw + 16384 + 2048 + 8192 + 32768);       // sampler: z of the block above, [sg][k]
  int (*s_cnt)[2] = reinterpret_cast<int (*)[2]>(smem_raw + 16384 + 2048 + 8192 + 32768 + 4096);         // per pair: groups produced, groups consumed
  double* s_invs = reinterpret_cast<double*>(smem_raw + 16384 + 2048 + 8192 + 32768 + 4096 + 64);        // 1 / s' of the block's rows in full precision (NP_NB)
  const SampleZParams* g_sz = a.sz + j0_of(J);                                              // full SampleZ tables, in L2: only 1 / s' is not in the LDS record
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef NP_ROLE_SPLIT_SIMD       /* every SIMD hosts one sampler and the helper of ANOTHER pair (waves w and w + 4 of a workgroup share a SIMD) */
  const bool helper = wave >= 4;
  const int pw = helper ? ((wave + 1) & 3) : wave;
#else
  // Measured in round 4 and NOT kept ("helper and sampler on different SIMDs"): samplers on the waves 0, 1, 4, 5 (two SIMDs, two dependent
  // chains each) and helpers on 2, 3, 6, 7 (the other two SIMDs, two Philox streams each).  tools/probe_np_chain.hip: the chain alone on a SIMD takes ~420 ticks
  // per step against ~1.1 k beside a helper -- but two helpers on one SIMD do not keep two samplers fed: C2 4.73 -> 5.03 ms, C4 4.84 -> 5.60 ms.  Same bits.
  const bool helper = (wave & 2) != 0;
  const int pw = (wave & 1) | ((wave >> 1) & 2);
#endif
  const size_t j0 = J * NP_NB;
  const int nrows = (int)(dim - j0 < (size_t)NP_NB ? dim - j0 : (size_t)NP_NB);
#ifdef NP_PROFILE
  long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long smax_[6] = {0, 0, 0, 0, 0, 0};
  (void)smax_;
  long long tprev_ = (long long)__builtin_readcyclecounter();
#endif
  {  // the in-block triangle goes to LDS by LDS-DMA (two 1 KiB pieces per wave), the small per-row tables through registers
    const double* src = a.Gin + J * NP_TRI + lane * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i;
      __builtin_amdgcn_global_load_lds(src + piece * 128, (lds_void_ptr)(s_tri + piece * 128), 16, 0, 0);
    }
    if (j0 + NP_NB < dim) {
      // g of the block above against this block's rows (32 KiB, the same for every wave): staged once per workgroup in the record ring, which the
      // helpers only start to fill behind the second barrier below (four 1 KiB pieces per wave)
      const double* srcn = a.Gnx + J * (NP_NB * NP_NB) + lane * 2;
      double* stage = reinterpret_cast<double*>(&s_ring[0][0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        __builtin_amdgcn_global_load_lds(srcn + piece * 128, (lds_void_ptr)(stage + piece * 128), 16, 0, 0);
      }
    }
    if (tid < NP_NB) {
      s_row[tid] = tid < nrows ? a.rows[j0 + tid] : NpRow{0.0, 0.f, 0, 1, 0, 0, 16};
      s_invs[tid] = tid < nrows ? g_sz[tid].inv_s : 1.0;
    }
    if (tid < 8) s_cnt[tid >> 1][tid & 1] = 0;
  }
  const int lam = lane & (LPD - 1);
  const int sgbase = lane & ~(LPD - 1);
  const int sg = lane / LPD;
  const size_t b = ((size_t)wg * 4 + (size_t)pw) * G + (size_t)sg;
  const bool live = b < B;
  const uint64_t index = first_index + b;
  const uint32_t tw = tag_word(tag, index);
  const uint64_t sgmask = G == 1 ? ~0ull : (((1ull << LPD) - 1) << sgbase);
  int* cnt_prod = &s_cnt[pw][0];
  int* cnt_cons = &s_cnt[pw][1];
  uint4* ring = &s_ring[pw][0];

  if (helper) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // tables and the staged g have landed
    __syncthreads();                                                // the samplers have consumed the staged g: the ring is free
    uint2* wstrip = &s_words[pw][0];
    int k = 0;
#pragma unroll
    for (int slot = G - 1; slot >= 0; --slot) {
      for (int grp = BPS - 1; grp >= 0; --grp) {                    // four steps slot * LPD + 4 grp + 3 .. + 0
        const int lbase = slot * LPD + grp * 4;
        if (lbase >= nrows) continue;                               // short top block (uniform; the sampler skips the same groups)
        while (__hip_atomic_load(cnt_cons, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < k - 1) {
          if (!std::is_same<BG, NpNoBg>::value) { if (!bg->step()) __builtin_amdgcn_s_sleep(1); }
          else __builtin_amdgcn_s_sleep(1);
        }
        {  // attempt words of the four steps, first LPD attempts each: lane (s, g) serves step lbase + 3 - s.
           // narrow rows: Philox block g holds attempts 4g .. 4g+3; wide rows: blocks g and g + BPS hold attempts 2g, 2g+1 and 2(g+BPS), 2(g+BPS)+1
          const int s = lam / BPS, g = lam % BPS;
          const int lstep = lbase + 3 - s;
          const uint32_t shs = s_row[lstep].sh;
          const uint32_t coord_s = (uint32_t)(j0 + lstep);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous group's reads of the strip are done
          uint2* dst = wstrip + (sg * 4 + s) * LPD;
          const U4 w = philox(seed, coord_s, (uint32_t)index, (uint32_t)g, tw);
          if (shs != 32) {
            uint4* d4p = reinterpret_cast<uint4*>(dst + 4 * g);
            d4p[0] = make_uint4(w.x >> 16, w.x & 0xffffu, w.y >> 16, w.y & 0xffffu);
            d4p[1] = make_uint4(w.z >> 16, w.z & 0xffffu, w.w >> 16, w.w & 0xffffu);
          } else {
            *reinterpret_cast<uint4*>(dst + 2 * g) = make_uint4(w.x, w.y, w.z, w.w);
          }
          if (__builtin_amdgcn_ballot_w64(shs == 32)) {
            const U4 w2 = philox(seed, coord_s, (uint32_t)index, (uint32_t)(g + BPS), tw);
            if (shs == 32) *reinterpret_cast<uint4*>(dst + 2 * (g + BPS)) = make_uint4(w2.x, w2.y, w2.z, w2.w);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          const int l = lbase + 3 - sp;
          const NpRow rw = s_row[l];
          const uint2 wd = wstrip[(sg * 4 + sp) * LPD + lam];
          const bool narrow = rw.sh == 16;
          const uint32_t Nf = rw.n_int - 1u;                        // candidates for a non-integral centre (the generic case)
          uint32_t low, idx;
          if (narrow) { const uint32_t prod = __umul24(wd.x, Nf); low = prod & 0xffffu; idx = prod >> 16; }
          else { low = wd.x * Nf; idx = __umulhi(wd.x, Nf); }
          const bool okidx = live && low >= rw.thr_frac;            // Lemire's rejection of the lowest fractions
          if (G == 1) {
            const float u = (float)idx * rw.inv_sk;
            const float wbf = (float)wd.y * (narrow ? 0x1.0p-16f : 0x1.0p-32f);
            ring[((k & 1) * 4 + sp) * 64 + lane] = make_uint4(__float_as_uint(u), __float_as_uint(wbf), (idx & 0x7fffffffu) | (okidx ? 0x80000000u : 0u), wd.y);
          } else {                                                  // packed: the sampler converts (two sets of attempts share the ring's bytes)
            reinterpret_cast<uint2*>(ring)[(((k & 1) * 4 + sp) * 2 + 0) * 64 + lane] = make_uint2((idx & 0x7fffffffu) | (okidx ? 0x80000000u : 0u), wd.y);
          }
        }
        if (G == 2) {
          // second set: attempts LPD .. 2 LPD - 1 of the same four steps (Philox blocks BPS .. 2 BPS - 1 of a narrow row, 2 BPS .. 4 BPS - 1 of a wide one).  With 32 lanes
          // per draw (11/12)^32 = 6 % of the draws find no candidate in the first set; the sampler screens this set in the same fast way instead of entering the generic rounds.
          const int s = lam / BPS, g = lam % BPS;
          const int lstep = lbase + 3 - s;
          const uint32_t shs = s_row[lstep].sh;
          const uint32_t coord_s = (uint32_t)(j0 + lstep);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the first set's reads of the strip are done
          uint2* dst = wstrip + (sg * 4 + s) * LPD;
          if (shs != 32) {
            const U4 w = philox(seed, coord_s, (uint32_t)index, (uint32_t)(BPS + g), tw);
            uint4* d4p = reinterpret_cast<uint4*>(dst + 4 * g);
            d4p[0] = make_uint4(w.x >> 16, w.x & 0xffffu, w.y >> 16, w.y & 0xffffu);
            d4p[1] = make_uint4(w.z >> 16, w.z & 0xffffu, w.w >> 16, w.w & 0xffffu);
          }
          if (__builtin_amdgcn_ballot_w64(shs == 32)) {
            const U4 w1 = philox(seed, coord_s, (uint32_t)index, (uint32_t)(2 * BPS + g), tw), w2 = philox(seed, coord_s, (uint32_t)index, (uint32_t)(3 * BPS + g), tw);
            if (shs == 32) {
              *reinterpret_cast<uint4*>(dst + 2 * g) = make_uint4(w1.x, w1.y, w1.z, w1.w);
              *reinterpret_cast<uint4*>(dst + 2 * (g + BPS)) = make_uint4(w2.x, w2.y, w2.z, w2.w);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) {
            const int l = lbase + 3 - sp;
            const NpRow rw = s_row[l];
            const uint2 wd = wstrip[(sg * 4 + sp) * LPD + lam];
            const bool narrow = rw.sh == 16;
            const uint32_t Nf = rw.n_int - 1u;
            uint32_t low, idx;
            if (narrow) { const uint32_t prod = __umul24(wd.x, Nf); low = prod & 0xffffu; idx = prod >> 16; }
            else { low = wd.x * Nf; idx = __umulhi(wd.x, Nf); }
            const bool okidx = live && low >= rw.thr_frac;
            reinterpret_cast<uint2*>(ring)[(((k & 1) * 4 + sp) * 2 + 1) * 64 + lane] = make_uint2((idx & 0x7fffffffu) | (okidx ? 0x80000000u : 0u), wd.y);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        ++k;
        if (lane == 0) __hip_atomic_store(cnt_prod, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (!std::is_same<BG, NpNoBg>::value) bg->drain();             // the sampler is still on its last groups: whatever background work is ready
    return;
  }

  // ---- sampler wave ------------------------------------------------------------------------------------------------------
  if (!std::is_same<BG, NpNoBg>::value && t_wait) { if (!bg->wait_word(t_wait)) { /* abort: fall through, the caller leaves after the barriers */ } }
  double t[G];
  long long zr[G];
#pragma unroll
  for (int s = 0; s < G; ++s) {
    const int row = s * LPD + lam;
    if (COH) t[s] = (live && row < nrows) ? __hip_atomic_load(a.T + (j0 + row) * a.ldt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    else t[s] = (live && row < nrows) ? a.T[(j0 + row) * a.ldt + b] : 0.0;
    zr[s] = 0;
  }
  int f = 0;
  const bool above = j0 + NP_NB < dim;
  double* zs = &s_zs[pw][0];
  if (above) {
    // the block above was sampled by the previous launch: its contribution to my rows, t = fma(-z_k, g[64 (J+1) + k][row], t) for k
    // ascending, is applied here (the update tiles stop below this block)
#pragma unroll
    for (int s = 0; s < G; ++s) {
      const int kk = s * LPD + lam;
      const size_t i = j0 + NP_NB + (size_t)kk;
      if (WALK) zs[sg * NP_NB + kk] = (live && i < dim) ? (double)zr_io[s] : 0.0;
      else zs[sg * NP_NB + kk] = (live && i < dim) ? a.Zf[((b / TR_BN) * a.nkb + i / 16) * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(i % 16))] : 0.0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (above) {
    const double* stage = reinterpret_cast<const double*>(&s_ring[0][0]) + lam;
#pragma unroll 16
    for (int kk = 0; kk < NP_NB; ++kk) {
      const double zk = zs[sg * NP_NB + kk];
#pragma unroll
      for (int s = 0; s < G; ++s) t[s] = fma(-zk, stage[kk * NP_NB + s * LPD], t[s]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  NP_T(0);
#ifndef NP_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);      // the sampler's dependent chain before the helper's Philox stream on the shared SIMD (round 4, tools/probe_np_chain.hip)
#endif
  auto bcast_d = [&](double v, int src_in_sg) -> double {
    if (G == 1) return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_in_sg), __builtin_amdgcn_readlane(__double2loint(v), src_in_sg));
    return __shfl(v, sgbase + src_in_sg);
  };
  // (an unconditional load from a clamped index + a select: the conditional form became an exec-masked block with a branch and a full LDS wait in every step)
  auto tri_at = [&](int l, int col) -> double {
    if (G != 1) return col < l ? s_tri[l * (l - 1) / 2 + col] : 0.0;      // (two draws per step: the conditional form measured faster there, C4 4.38 vs 4.44 ms)
    const double v = s_tri[col < l ? l * (l - 1) / 2 + col : 0];
    return col < l ? v : 0.0;
  };
  const uint64_t live_w = __builtin_amdgcn_ballot_w64(live);
  int k = 0;
#pragma unroll
  for (int slot = G - 1; slot >= 0; --slot) {
    for (int grp = BPS - 1; grp >= 0; --grp) {
      const int lbase = slot * LPD + grp * 4;
      if (lbase >= nrows) continue;
      while (__hip_atomic_load(cnt_prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      NP_T(1);
      const uint4* rslot = ring + (k & 1) * 4 * 64 + lane;
      const uint2* rslot2 = reinterpret_cast<const uint2*>(ring) + (k & 1) * 4 * 2 * 64 + lane;     // G == 2: [step][set][lane], 8-byte records
      // the row records of the group's four steps up front: they are wave-uniform, hipcc moves them to SGPRs (v_readfirstlane) right behind their LDS load, and
      // fetched one step ahead that was an exposed LDS round trip in front of every step's chain; once per group it is one
      // (G == 2 keeps the fetch one step ahead: with two draws per step its registers are full)
      NpRow rws[4];
#pragma unroll
      for (int sp = 0; sp < (G == 1 ? 4 : 1); ++sp) rws[sp] = s_row[lbase + 3 - sp];
      // per-lane operands of the first step of the group; those of the following steps are fetched one step ahead
      double gl[G];
#pragma unroll
      for (int s2 = 0; s2 <= slot; ++s2) gl[s2] = tri_at(lbase + 3, s2 * LPD + lam);
      uint4 rec = make_uint4(0, 0, 0, 0);
      if (G == 1) rec = rslot[0];
      else { const uint2 r2 = rslot2[0]; rec.z = r2.x; rec.w = r2.y; }
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int l = lbase + 3 - sp;
        const NpRow rw = rws[G == 1 ? sp : 0];
        NpRow rwn = rw;
        double gln[G];
        uint4 recn = rec;
#pragma unroll
        for (int s2 = 0; s2 <= slot; ++s2) gln[s2] = gl[s2];
        if (sp < 3) {
          if (G != 1) rwn = s_row[l - 1];
          // (the record first: its registers are the ones the previous step's fetch wrote, and hipcc puts a full LDS wait in front of that overwrite -- behind the
          // g load it waited for that load, a round trip in front of every step's chain; in front of it nothing is in flight)
          if (G == 1) { recn = rslot[(sp + 1) * 64]; asm volatile("" ::: "memory"); }
#pragma unroll
          for (int s2 = 0; s2 <= slot; ++s2) gln[s2] = tri_at(l - 1, s2 * LPD + lam);
          if (G != 1) { const uint2 r2 = rslot2[(sp + 1) * 2 * 64]; recn.z = r2.x; recn.w = r2.y; }
        }
        if (l < nrows) {
          const int ls = l - slot * LPD;
          const double tl = bcast_d(t[slot], ls);
          const double cen = tl * rw.inv_n2;
          const uint32_t coord = (uint32_t)(j0 + l);
          long long z = 0;
          double zd = 0.0;                                          // z as a double, converted where z is decided: from 32 bits on the common paths
          bool got = !live;
          uint32_t t0 = 0;
          NP_T(2);
          const bool narrow = rw.sh == 16;
          const uint32_t idx = rec.z & 0x7fffffffu;
          const bool okidx = (rec.z >> 31) != 0;
          const float u = G == 1 ? __uint_as_float(rec.x) : (float)idx * rw.inv_sk;                                   // the helper's expressions, evaluated here for
          const float wbf0 = G == 1 ? __uint_as_float(rec.y) : (float)rec.w * (narrow ? 0x1.0p-16f : 0x1.0p-32f);    // the packed records of G == 2
          const float wbf = okidx ? wbf0 : __builtin_inff();        // a Lemire-rejected attempt is never a candidate: folded into the word, so that each ballot below is ONE compare
          const float wbe = wbf + (narrow ? 0x1.0p-16f : 1e-7f);
          // the two thresholds moved to the words' side (off the chain: they depend on the record only), so that both compares read v_exp_f32's result directly:
          //   candidate  <=>  wbf <= 1.001 rho + 1e-9  <=>  (wbf - 1e-9) / 1.001 <= rho      (factor rounded DOWN: a superset of the old candidates)
          //   certain    <=>  wbe <= 0.999 rho         <=>  wbe / 0.999 <= rho               (factor rounded UP: a subset of the old certain accepts)
          const float wcand = (wbf - 1e-9f) * 0.998999f;
          const float wsure = wbe * 1.001002f;
          // --- the dependent chain ----------------------------------------------------------------------------------
          const double cc = ceil(cen);
          const float frf = (float)(cc - cen);                      // ceil(c) - c in [0, 1): 0 for an integral centre (and for one within 1e-38 below zero: those
                                                                    // take the generic rounds too, which are exact for every centre)
          const float c_rel = frf - (float)rw.c6;                   // lo - c with lo = ceil(c) - ceil(6 s')
          const int lo = (int)cc - rw.c6;
          const float ak = fmaf(c_rel, rw.inv_sk, u);
          const float rho = __builtin_amdgcn_exp2f(-(ak * ak));     // exp(-pi a^2)
          // Each mask is the ballot of a single compare (a v_cmp writing the SGPR pair); conditions are combined on the masks with scalar instructions.  hipcc turns
          // the ballot of a combined condition into v_cndmask 0 / 1 + v_cmp_ne: two more dependent vector instructions per ballot on the chain.
          // not covered by the lines above: integral centres (one more candidate), huge centres, candidate ranges beyond fp32
          const uint64_t mc_w = __builtin_amdgcn_ballot_w64(wcand <= rho), m1_w = __builtin_amdgcn_ballot_w64(wsure <= rho);
          const int zc = lo + (int)idx;                              // every lane's own candidate, as a double too: ready before the ballots are
          double zcd = 0.0;
          if (G == 1) { zcd = (double)zc; asm volatile("" : "+v"(zcd)); }      // (kept per lane: otherwise hipcc reads the integer across and converts behind the readlane, on the chain)
          const uint64_t plain_w = __builtin_amdgcn_ballot_w64(frf > 0.0f) & __builtin_amdgcn_ballot_w64(fabs(cen) < 0x1.0p30);      // non-integral, below 2^30
          const uint64_t bad_w = (rw.sh == 0 ? ~0ull : ~plain_w) & live_w;
          NP_T(3);
          NP_EVENT(0);
          if (bad_w) NP_EVENT(3);
          // common case: the first candidate of the draw (in attempt = lane order) is a certain accept
          bool settle = false;
          bool fast1 = false;                                       // G == 1: the draw was decided by the fast path (wave-uniform: everything rare hangs off ONE scalar branch)
          if (G == 1) {
            const int fl = mc_w ? __builtin_ctzll(mc_w) : 0;
            fast1 = bad_w == 0 && ((m1_w >> fl) & 1);
            if (fast1) {
              zd = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(zcd), fl), __builtin_amdgcn_readlane(__double2loint(zcd), fl));
              z = (long long)__builtin_amdgcn_readlane(zc, fl);
              got = true;
            }
            else settle = live;
          } else {
            // G == 2: two sets of LPD attempts with the fast screen.  Set A as always; set B (attempts LPD .. 2 LPD - 1) is read and screened only when a
            // draw is still open after set A -- no candidate in it (6 % of the draws) or all its candidates settled as rejects -- so the common step pays nothing.
            const bool usable = !(bad_w & sgmask);
            auto settle_set = [&](uint64_t m1, uint64_t m2, uint32_t idxv, uint32_t wbraw, uint32_t tbase) {
              while (true) {
                const uint64_t cnd = m1 | m2;
                const bool pending = !got && cnd != 0;
                if (!__builtin_amdgcn_ballot_w64(pending)) break;
                const int fl = pending ? (__ffsll((long long)cnd) - 1) : lane;
                const bool sure = (m1 >> fl) & 1;
                bool acc = false;
                if (pending && !sure && lane == fl) {
                  uint32_t ta = tbase + (uint32_t)lam;
                  asm volatile("" : "+v"(ta));
                  acc = sz_decide(seed, coord, (uint32_t)index, tw, ta, (long long)lo + (long long)idxv, wbraw, cen, s_invs[l], rw.sh);
                }
                const uint64_t accm = __builtin_amdgcn_ballot_w64(acc);
                const int xi = __shfl((int)idxv, fl);
                if (pending) {
                  if (sure || ((accm >> fl) & 1)) { z = (long long)(lo + xi); zd = (double)(lo + xi); got = true; }
                  else m2 &= ~(1ull << fl);
                }
              }
            };
            {
              const uint64_t cand = mc_w & sgmask;
              const int fl = cand ? (__ffsll((long long)cand) - 1) : lane;
              const int xi = __shfl((int)idx, fl);
              if (live && usable && cand && ((m1_w >> fl) & 1)) { z = (long long)(lo + xi); zd = (double)(lo + xi); got = true; }
              const bool pend = live && !got && usable && cand != 0;                 // the first candidate of set A is a "to be settled" one
              if (__builtin_expect(__builtin_amdgcn_ballot_w64(pend) != 0, 0)) { NP_EVENT(1); settle_set(pend ? (m1_w & sgmask) : 0, pend ? ((mc_w & ~m1_w) & sgmask) : 0, idx, rec.w, 0u); }
            }
            const bool need_b = live && !got && usable;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(need_b) != 0, 0)) {
              const uint2 rb = rslot2[(sp * 2 + 1) * 64];
              const uint32_t idx_b = rb.x & 0x7fffffffu;
              const bool ok_b = (rb.x >> 31) != 0;
              const float u_b = (float)idx_b * rw.inv_sk;
              const float wbf_b = (float)rb.y * (narrow ? 0x1.0p-16f : 0x1.0p-32f);
              const float wbe_b = wbf_b + (narrow ? 0x1.0p-16f : 1e-7f);
              const float ak_b = fmaf(c_rel, rw.inv_sk, u_b);
              const float rho_b = __builtin_amdgcn_exp2f(-(ak_b * ak_b));
              const bool cand_bb = ok_b && wbf_b <= fmaf(rho_b, 1.001f, 1e-9f);
              const bool sure_bb = ok_b && wbe_b <= rho_b * 0.999f;
              const uint64_t mcb_w = __builtin_amdgcn_ballot_w64(cand_bb), m1b_w = __builtin_amdgcn_ballot_w64(sure_bb);
              const uint64_t cand = mcb_w & sgmask;
              const int fl = cand ? (__ffsll((long long)cand) - 1) : lane;
              const int xi = __shfl((int)idx_b, fl);
              if (need_b && cand && ((m1b_w >> fl) & 1)) { z = (long long)(lo + xi); zd = (double)(lo + xi); got = true; }
              const bool pend = need_b && !got && cand != 0;
              if (__builtin_amdgcn_ballot_w64(pend)) settle_set(pend ? (m1b_w & sgmask) : 0, pend ? ((mcb_w & ~m1b_w) & sgmask) : 0, idx_b, rb.y, (uint32_t)LPD);
            }
            t0 = usable ? 2u * (uint32_t)LPD : 0u;                  // special centres start over with the generic rounds
          }
          if (G == 1 && __builtin_expect(!fast1, 0) && __builtin_amdgcn_ballot_w64(settle) != 0) {
            NP_EVENT(1);
            // rare: a "to be settled" attempt comes first, or there is no candidate among the first LPD attempts, or the centre is special
            const bool usable = !(bad_w & sgmask);
            uint64_t m1 = usable ? (m1_w & sgmask) : 0, m2 = usable ? ((mc_w & ~m1_w) & sgmask) : 0;
            while (true) {
              const uint64_t cand = m1 | m2;
              const bool pending = !got && cand != 0;
              if (!__builtin_amdgcn_ballot_w64(pending)) break;
              const int fl = pending ? (__ffsll((long long)cand) - 1) : lane;
              const bool sure = (m1 >> fl) & 1;
              bool acc = false;
              if (pending && !sure && lane == fl) {               // keep every bit of the exact decision (incl. its tie-break Philox block) inside the branch
                uint32_t ta = (uint32_t)lam;
                asm volatile("" : "+v"(ta));
                acc = sz_decide(seed, coord, (uint32_t)index, tw, ta, (long long)lo + (long long)idx, rec.w, cen, s_invs[l], rw.sh);
              }
              const uint64_t accm = __builtin_amdgcn_ballot_w64(acc);
              const int xi = __shfl((int)idx, fl);
              if (pending) {
                if (sure || ((accm >> fl) & 1)) { z = (long long)(lo + xi); zd = (double)(lo + xi); got = true; }
                else m2 &= ~(1ull << fl);
              }
            }
            t0 = usable ? LPD : 0;                                 // special centres start over with the generic rounds
          }
          NP_T(4);
          // A centre at or beyond 2^62 (a modulus near 2^60 over a Gram-Schmidt vector of tiny norm): ceil(c) - ceil(6 s') and the candidates no longer fit the
          // walk's 64-bit integers.  The draw ends with 0 and the call reports PSF_ERR_SAMPLER; the oracle does the same in orc_sample_z, where the conversion
          // would otherwise be undefined behaviour (found by tools/fuzz_configs.py: x86 and gfx950 saturate differently, silently).  Every such centre comes
          // through here: `bad` sends |c| >= 2^30 to the generic rounds.
          if ((G != 1 || __builtin_expect(!fast1, 0)) && __builtin_expect(__builtin_amdgcn_ballot_w64(!got) != 0, 0)) {      // generic rounds: attempts t0 + lam, t0 + LPD + lam, ...
            if (!got && !(fabs(cen) < 0x1.0p62)) { f = 1; z = 0; zd = 0.0; got = true; }      // (inside the rare branch: as a test of its own in front of it the step cost 2 % more)
            NP_EVENT(2);
            // (an "accepted for certain" class here as in the first round was measured: C2 3.5 % slower, C4 unchanged -- the rounds are rare and the
            // extra live values cost the hot path registers)
            // the row's SampleZ parameters rebuilt from LDS (an L2 round trip here cost ~2 k cycles of every generic phase; at C4, where 32 first-round
            // attempts leave 12 % of the wave-steps without a candidate, these phases are what the slowest wave of a launch is made of)
            SampleZParams sp2;
            if (rw.sh != 0) {
              sp2.inv_s = s_invs[l]; sp2.c6 = (long long)rw.c6; sp2.n_int = rw.n_int; sp2.f6 = (long long)rw.n_int - 1 - (long long)rw.c6;
              sp2.thr_int = rw.thr_int; sp2.thr_frac = rw.thr_frac; sp2.sh = rw.sh;
            } else {
              sp2 = g_sz[l];                                         // rows without a fast path (more than 2^24 candidates): the record does not carry c6
            }
            const SzRange rg = sz_range(cen, sp2);
            for (; t0 < kMaxAttempts; t0 += LPD) {
              if (!__builtin_amdgcn_ballot_w64(!got)) break;
              bool maybe = false;
              long long x = 0;
              uint32_t wa = 0, wb = 0;
              const uint32_t ta = t0 + (uint32_t)lam;
              if (!got) {
                sz_attempt_words(seed, coord, (uint32_t)index, tw, ta, rg.sh, &wa, &wb);
                maybe = sz_maybe(wa, wb, rg, cen, sp2.inv_s, &x);
              }
              uint64_t m2 = __builtin_amdgcn_ballot_w64(maybe) & sgmask;
              while (true) {
                const bool pending = !got && m2 != 0;
                if (!__builtin_amdgcn_ballot_w64(pending)) break;
                const int fl = pending ? (__ffsll((long long)m2) - 1) : lane;
                bool acc = false;
                if (pending && lane == fl) {
                  uint32_t tb = ta;
                  asm volatile("" : "+v"(tb));
                  acc = sz_decide(seed, coord, (uint32_t)index, tw, tb, x, wb, cen, sp2.inv_s, rg.sh);
                }
                const uint64_t accm = __builtin_amdgcn_ballot_w64(acc);
                const long long xs = __shfl(x, fl);
                if (pending) {
                  if ((accm >> fl) & 1) { z = xs; zd = (double)xs; got = true; }
                  else m2 &= ~(1ull << fl);
                }
              }
            }
            if (!got) { f = 1; z = (long long)floor(cen + 0.5); zd = (double)z; }
          }
          NP_T(7);
          if (lam == ls) zr[slot] = z;
          // (the common draw fits 32 bits: |c'| < 2^30 on the fast path; the 64-bit conversion is a sequence of its own on the chain)
          const double nz = -zd;
#pragma unroll
          for (int s2 = 0; s2 <= slot; ++s2) t[s2] = fma(nz, gl[s2], t[s2]);
          NP_T(5);
        }
        if (G != 1) rws[0] = rwn;
        rec = recn;
#pragma unroll
        for (int s2 = 0; s2 <= slot; ++s2) gl[s2] = gln[s2];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // my reads of this group's ring slots are complete
      ++k;
      if (lane == 0) __hip_atomic_store(cnt_cons, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  int use1 = 0, use2 = 0, big = 0;
  if (WALK) {
#pragma unroll
    for (int s = 0; s < G; ++s) zr_io[s] = zr[s];
  }
  if (live) {
    const size_t plane = a.zplane;
#pragma unroll
    for (int s = 0; s < G; ++s) {
      const int row = s * LPD + lam;
      if (row >= nrows) continue;
      const size_t i = j0 + row;
      const long long z = zr[s];
      double* zdst = a.Zf + ((b / TR_BN) * a.nkb + i / 16) * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(i % 16));
      if (COH) __hip_atomic_store(zdst, (double)z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *zdst = (double)z;
      if (z > 8355711ll || z < -8355711ll) big = 1;               // beyond three balanced digits: the 64-bit recombination takes the call
      const int z32 = (int)z;
      const int d0 = (int)(int8_t)(z32 & 0xff);
      const int z1 = (z32 - d0) >> 8;
      const int d1 = (int)(int8_t)(z1 & 0xff);
      const int d2 = (z1 - d1) >> 8;
      if (d1) use1 = 1;
      if (d2) use2 = 1;
      const size_t addr = ((i >> 4) * a.ld + b) * 16 + (i & 15);
      a.Z8[addr] = (int8_t)d0;
      a.Z8[plane + addr] = (int8_t)d1;
      a.Z8[2 * plane + addr] = (int8_t)d2;
    }
  }
  // one atomic per wave at most, and none once the flag is up (at C2 / C4 "second digit in use" is raised by almost every wave of every launch)
  if (__builtin_amdgcn_ballot_w64(f) && lane == 0) atomicOr(a.flags, 1);
  if (__builtin_amdgcn_ballot_w64(use1) && lane == 0 && __hip_atomic_load(a.flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(a.flags + 1, 1);
  if (__builtin_amdgcn_ballot_w64(use2) && lane == 0 && __hip_atomic_load(a.flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(a.flags + 2, 1);
  if (__builtin_amdgcn_ballot_w64(big) && lane == 0) atomicOr(a.flags + 3, 1);
#ifdef NP_PROFILE
  NP_T(6);
  if (wg == 0 && tid == 0) for (int kk = 0; kk < 7; ++kk) atomicAdd((unsigned long long*)&g_np_prof[kk], (unsigned long long)tacc_[kk]);
  if (lane == 0) {      // spread over the sampler waves of the launch: slowest wave, accumulated over the launches of the call
    long long tot = 0;
    for (int kk = 0; kk < 8; ++kk) tot += tacc_[kk];
    atomicMax((unsigned long long*)&g_np_prof[7], (unsigned long long)tot);
#if NP_PROFILE == 3
    for (int kk = 0; kk < 8; ++kk) { atomicAdd(&g_np_sum[kk], (unsigned long long)tacc_[kk]); atomicMax(&g_np_max[kk], (unsigned long long)tacc_[kk]); }
    atomicMax(&g_np_single[0], (unsigned long long)smax_[0]); atomicMax(&g_np_single[1], (unsigned long long)smax_[1]); atomicMax(&g_np_single[4], (unsigned long long)smax_[4]);
    atomicAdd(&g_np_single[2], (unsigned long long)smax_[2]); atomicAdd(&g_np_single[3], (unsigned long long)smax_[3]); atomicAdd(&g_np_single[5], (unsigned long long)smax_[5]);
#endif
  }
#endif
}

// One update tile of a launch: T[rows, 128 preimages] takes the blocks J_first, J_first - 1, ... (`nsub` of them, in this order), each as
// the contract's chain t = fma(-z_j, g[j][row], t) over its 64 rows j ascending -- exactly a v_mfma_f64_16x16x4_f64 K loop with the T tile
// as the accumulator and -g as the A operand.  The tile is loaded once, stays in the accumulator registers across all the blocks of
// the job and is stored once: the far rows of the walk cross HBM once per panel of NP_PANEL blocks instead of once per block.
// Rows outside [row_lo, row_hi) belong to somebody else (the sampler, another job) and are not written.
constexpr int NP_PANEL = 8;
struct NpUpdateJob { int J_first, nsub; int rb0, nrb; size_t row_lo, row_hi; };

template <bool COH = false>
__device__ __forceinline__ void np_update_tile(double* smem, unsigned tile, const NpUpdateJob& job, int nbj, const double* __restrict__ Gp, const double* __restrict__ Zf,
                                               size_t nkb, double* __restrict__ T, size_t ldt) {
  // 128 x 128 tile, eight waves as 4 x 2, wave tile 32 x 64 (64 accumulator VGPRs: the launch must keep four waves per SIMD so that
  // sampler and update workgroups share the CUs)
  const int bj = (int)(tile % (unsigned)nbj), bi = job.rb0 + (int)(tile / (unsigned)nbj);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t row0 = (size_t)bi * TR_BM + wr * 32, col0 = (size_t)bj * TR_BN + wc * 64;
  d4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = T[(row0 + i * 16 + (lane >> 4) + 4 * r) * ldt + col0 + j * 16 + (lane & 15)];
  const int nk = 4 * job.nsub;                                   // K chunks of 16 over all the blocks of the job
  auto stage_load = [&](int kc, int buf) {
    const size_t Jb = (size_t)(job.J_first - (kc >> 2));
    const double* ga = Gp + (np_panel_base(Jb) + (size_t)bi * 4 + (size_t)(kc & 3)) * TR_CHUNK + lane * 2;
    const double* gb = Zf + ((size_t)bj * nkb + Jb * (NP_NB / 16) + (size_t)(kc & 3)) * TR_CHUNK + lane * 2;
    double* la = smem + buf * (2 * TR_CHUNK);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i;                            // sixteen 1 KiB pieces per chunk
      __builtin_amdgcn_global_load_lds(ga + piece * 128, (lds_void_ptr)(la + piece * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gb + piece * 128, (lds_void_ptr)(la + TR_CHUNK + piece * 128), 16, 0, COH ? 16 : 0);   // COH: sc1 (agent scope): z comes from other XCDs
    }
  };
  stage_load(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < nk) stage_load(kc + 1, cur ^ 1);
    const double* sA = smem + cur * (2 * TR_CHUNK);
    const double* sB = sA + TR_CHUNK;
#pragma unroll
    for (int ks = 0; ks < TR_BK / 4; ++ks) {
      double av[2], bv[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = -sA[(ks * 8 + wr * 2 + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) bv[i] = sB[(ks * 8 + wc * 4 + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t row = row0 + i * 16 + (lane >> 4) + 4 * r;
        if (row >= job.row_lo && row < job.row_hi) {
          if (COH) __hip_atomic_store(T + row * ldt + col0 + j * 16 + (lane & 15), acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else T[row * ldt + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
        }
      }
}

// One launch per block J of the walk (512 threads per workgroup, 64 KiB of dynamic LDS):
//   workgroups [0, nS)   sample block J (np_sample_body): the critical path
//   then the update tiles of up to three jobs, which hide behind the sampling:
//     window  block J + 1, sampled by the previous launch, goes into the rows from the start of the panel below its own down to
//             (not including) block J, which takes it inside the sampler
//     far A   once per panel, in the launch of the first block of the panel below: all blocks of the completed panel go into the rows
//             of the panel after next (the rows the coming window updates will touch, so they must be served first)
//     far B   in the same launch, the same blocks into all rows further down (needed one panel later)
// so no second stream, no events, and the far rows of T are touched once per panel instead of once per block.
struct NpStepJobs { NpUpdateJob job[3]; unsigned ntiles[3]; };

template <int G>
__global__ __launch_bounds__(512, 4) void k_np_step(NpSampleArgs a, size_t dim, size_t J, uint64_t seed, uint32_t tag, uint64_t first_index, size_t B, unsigned nS,
                                                    NpStepJobs jobs, int nbj, const double* __restrict__ Gp, double* __restrict__ Tm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char np_smem[];
  unsigned id = blockIdx.x;
  if (id < nS) { np_sample_body<G>(np_smem, id, a, dim, J, seed, tag, first_index, B); return; }
  id -= nS;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (id < jobs.ntiles[q]) { np_update_tile(reinterpret_cast<double*>(np_smem), id, jobs.job[q], nbj, Gp, a.Zf, a.nkb, Tm, a.ldt); return; }
    id -= jobs.ntiles[q];
  }
}

// ---- the whole walk in ONE launch (k_np_walk) -----------------------------------------------------------------------------------------------------------
// k_np_step pays, per block of 64 steps, a dependent launch (~5 us) and the wait for the slowest of ALL sampler waves of the batch (rare exact decisions and generic
// rounds: a fifth of a launch at C2, profiles/r04_notes.md).  The data dependence does not ask for either: block J of a preimage needs the draws of the blocks
// above it for THAT preimage only.  Here the batch is cut into column groups of NP_GW = 64 preimages that never synchronise with each other:
//   sampler workgroups (one per CU: 4 wave pairs = 4 G preimages) walk J = nblk-1 ... 0 without leaving the kernel; after block J a workgroup publishes its
//       z (write-through stores, then one counter add) and, before block J, waits for the running projections of that block's rows;
//   updater workgroups (the second workgroup of every CU) OWN the running projections T of their group: updater u of a group keeps the 64-row blocks
//       u, u + ug, u + 2 ug, ... x 64 preimages in ACCUMULATOR REGISTERS for the whole walk (16 x 64 per wave and slot, at most three slots: 96 VGPRs), so T never
//       moves except once to the samplers.  Per block J it waits for the group's z, stages them in LDS and applies t = fma(-z_j, g[j][.], t), j ascending -- the
//       same v_mfma_f64_16x16x4_f64 K loop as np_update_tile, A operand straight from the fragment-ordered panels -- to every block it owns below J - 1 (block J - 1
//       takes Z_J inside the sampler, as before).  Block J - 2 is then complete up to its neighbour: its owner does it first, stores its 64 x 64 values
//       (write-through) and raises the block's flag, a full block time before the samplers ask for it.
// Same chains, same order (blocks descending, j ascending inside a block): the same bits as the launch-per-block walk and the oracle.
// Hand-offs follow the write-through form of the CDNA4 guide (Guideline 16, R1): every handed-off value is an agent-scope (sc1) store, every storing wave drains
// (s_waitcnt vmcnt(0)) before ONE lane signals, the consumer polls relaxed and reads the values with agent-scope loads only.  Every spin is bounded: a wait that
// runs out raises the abort word, every workgroup leaves at its next wait and the call reports PSF_ERR_SAMPLER (flags[0]); nothing is relaunched.
constexpr int NP_GW = 64;                     // preimages per column group
constexpr int NP_WALK_SLOTS = 3;              // 64-row blocks x 64 preimages of T per updater WAVE QUARTET: a workgroup owns at most 2 * NP_WALK_SLOTS blocks
struct NpWalkSync {
  unsigned* zcount;                           // [group][block]: sampler workgroups of the group that have published the block's z
  unsigned* tready;                           // [group][block]: the block's running projections are complete up to the block after next
  unsigned* abort;                            // [0] raised by a wait that ran out
  unsigned nblk_stride;
  unsigned spin_limit;                        // polls (x s_sleep(2)) before a wait gives up: 2^22 ~ seconds, a walk lasts milliseconds (env PSF_NP_WALK_SPINS: tests of the give-up path)
};

// one lane polls, the workgroup learns the outcome through LDS; false = abort
__device__ __forceinline__ bool np_walk_wait(const unsigned* word, unsigned target, unsigned* abort_word, unsigned spin_limit, int* s_flag) {
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
      if (spins >= spin_limit) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
    }
    *s_flag = ok;
  }
  __syncthreads();
  const int ok = *s_flag;
  __syncthreads();
  return ok != 0;
}

template <int G>
__device__ __forceinline__ void np_walk_sampler(unsigned char* smem, unsigned wg, const NpSampleArgs& a, size_t dim, size_t nblk, uint64_t seed, uint32_t tag,
                                                uint64_t first_index, size_t B, const NpWalkSync& sy) {
  int* s_flag = reinterpret_cast<int*>(smem + 65536 - 16);
  const unsigned grp = wg / (unsigned)(NP_GW / (4 * G));
  long long zr[G];
#pragma unroll
  for (int s = 0; s < G; ++s) zr[s] = 0;
  for (size_t J = nblk; J-- > 0;) {
    if (J + 2 < nblk) {                       // the two top blocks come from the initial projection alone
      if (!np_walk_wait(sy.tready + (size_t)grp * sy.nblk_stride + J, 1u, sy.abort, sy.spin_limit, s_flag)) return;
    }
    np_sample_body<G, true, true>(smem, wg, a, dim, J, seed, tag, first_index, B, zr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave drains its write-through stores of z ...
    __syncthreads();
    if (threadIdx.x == 0 && J >= 2) __hip_atomic_fetch_add(sy.zcount + (size_t)grp * sy.nblk_stride + J, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then ONE lane signals
  }
}

// updater `ul` of `ug` in group `grp`: blocks ul, ul + ug, ... (below nblk - 2); block number k of the workgroup lives in slot k / 2 of the waves 4 (k & 1) ... + 3,
// wave w holding its rows 16 (w & 3) ... + 15 x the group's 64 preimages as four 16 x 16 accumulator fragments
__device__ __forceinline__ void np_walk_updater(unsigned char* smem, unsigned grp, unsigned ul, unsigned ug, unsigned need, const NpSampleArgs& a, size_t nblk,
                                                const double* __restrict__ Gp, double* __restrict__ T, const NpWalkSync& sy) {
  double* s_z = reinterpret_cast<double*>(smem);                         // z of the block: [k-step 16][fragment 4][lane 64] = 32 KiB
  int* s_flag = reinterpret_cast<int*>(smem + 65536 - 16);
  int* s_arr = reinterpret_cast<int*>(smem + 65536 - 32);               // arrivals of the four waves that hand a block over
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t col0 = (size_t)grp * NP_GW;
  const size_t ldt = a.ldt;
  // every address below is a wave-uniform base (SGPRs) plus ONE 32-bit lane offset: nothing per slot, row or fragment is kept in vector registers
  // (the 96 accumulator registers leave 32; address pairs hoisted out of the block loop were the kernel's scratch spills)
  const uint32_t t_lane = (uint32_t)(((size_t)(lane >> 4) * ldt + (size_t)(lane & 15)) * sizeof(double));   // element (lane >> 4, lane & 15) of a 16 x 16 fragment of T
  const uint32_t f_lane = (uint32_t)(lane * sizeof(double));                                                 // lane of a 512-byte operand fragment
  auto t_at = [&](size_t row, size_t col) -> double* {                   // (row, col) wave-uniform
    return reinterpret_cast<double*>(reinterpret_cast<char*>(T + row * ldt + col) + t_lane);
  };
  d4 acc[NP_WALK_SLOTS][4];
  long blk[NP_WALK_SLOTS];                                              // the block of each slot, -1: none
#pragma unroll
  for (int sl = 0; sl < NP_WALK_SLOTS; ++sl) {
    const long k = 2 * sl + (wave >> 2);
    const long Jb = (long)ul + k * (long)ug;
    blk[sl] = Jb + 2 < (long)nblk ? Jb : -1;
    const size_t row0 = (size_t)(blk[sl] < 0 ? 0 : blk[sl]) * NP_NB + (size_t)(wave & 3) * 16;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[sl][nf][r] = blk[sl] < 0 ? 0.0 : *t_at(row0 + 4 * r, col0 + nf * 16);
  }
  if (tid == 0) *s_arr = 0;
  const size_t bj = col0 / TR_BN;
  const int t0 = (int)((col0 % TR_BN) / 16);                             // first of the four 16-preimage tiles of the group inside the 128-wide chunk
  for (size_t J = nblk; J-- > 2;) {
    if (!np_walk_wait(sy.zcount + (size_t)grp * sy.nblk_stride + J, need, sy.abort, sy.spin_limit, s_flag)) return;
    {  // the block's z for the group's 64 preimages: rows k = 64 J + 16 kc + 4 ks + (lane >> 4), preimage 16 nf + (lane & 15); chunk stream (bj, kb) of Zf
      const double* zsrc = a.Zf + (bj * a.nkb + J * (NP_NB / 16)) * TR_CHUNK;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = i * 8 + wave;                                      // 64 (k-step, fragment) pairs of 64 lanes
        const int kstep = e >> 2, nf = e & 3, kc = kstep >> 2, ks = kstep & 3;
        const double* src = reinterpret_cast<const double*>(reinterpret_cast<const char*>(zsrc + (size_t)kc * TR_CHUNK + (size_t)((ks * 8 + t0 + nf) * 64)) + f_lane);
        s_z[e * 64 + lane] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    // the block after next first: its owner hands it to the samplers
    const long Ju = (long)J - 2;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int sl = 0; sl < NP_WALK_SLOTS; ++sl) {
        const bool urgent = blk[sl] == Ju;
        if (blk[sl] < 0 || blk[sl] > Ju || urgent != (pass == 0)) continue;
        const size_t i0 = (size_t)blk[sl] * NP_NB + (size_t)(wave & 3) * 16;
        const double* ga = Gp + (np_panel_base(J) + (i0 / 128) * 4) * TR_CHUNK + ((i0 % 128) / 16) * 64;     // wave-uniform
#pragma unroll
        for (int kstep = 0; kstep < 16; ++kstep) {
          const double av = -*reinterpret_cast<const double*>(reinterpret_cast<const char*>(ga + (size_t)(kstep >> 2) * TR_CHUNK + (kstep & 3) * 512) + f_lane);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[sl][nf] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, s_z[(kstep * 4 + nf) * 64 + lane], acc[sl][nf], 0, 0, 0);
        }
        if (urgent) {
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r)      // (rows from the loop's own J: from blk[sl] the sixteen addresses are loop invariants that hipcc keeps in registers and spills)
              __hip_atomic_store(t_at((size_t)Ju * NP_NB + (size_t)(wave & 3) * 16 + 4 * r, col0 + nf * 16), acc[sl][nf][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            const int prev = __hip_atomic_fetch_add(s_arr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((prev & 3) == 3) __hip_atomic_store(sy.tready + (size_t)grp * sy.nblk_stride + (size_t)Ju, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    __syncthreads();                                                     // the staged z are free
  }
}

// grid: nS sampler workgroups, then the updater workgroups (ug per group); two workgroups per CU, all resident at once
template <int G>
__global__ __launch_bounds__(512, 4) void k_np_walk(NpSampleArgs a, size_t dim, size_t nblk, uint64_t seed, uint32_t tag, uint64_t first_index, size_t B, unsigned nS,
                                                    unsigned ngroups, unsigned ug, const double* __restrict__ Gp, double* __restrict__ Tm, NpWalkSync sy) {
  extern __shared__ __attribute__((aligned(16))) unsigned char np_smem[];
  unsigned id = blockIdx.x;
  if (id < nS) {
    np_walk_sampler<G>(np_smem, id, a, dim, nblk, seed, tag, first_index, B, sy);
  } else {
    id -= nS;
    const unsigned grp = id / ug, ul = id % ug;
    if (grp >= ngroups) return;
    const unsigned per = (unsigned)(NP_GW / (4 * G));                     // sampler workgroups of a full group
    const unsigned first = grp * per, need = nS - first < per ? nS - first : per;
    np_walk_updater(np_smem, grp, ul, ug, need, a, nblk, Gp, Tm, sy);
  }
  // (a workgroup that gave up gets here with the abort word raised: the call is then re-run by k_np_walk_solo, enqueued right behind this launch)
}

// ---- the walk without any wait between workgroups (k_np_walk_solo): what runs when k_np_walk gave up -----------------------------------------------------------
// k_np_walk needs every one of its workgroups resident at the same time; the host decides that from the device's occupancy figures, which hold when the launch has
// the chip to itself.  A second tenant (another process, a long kernel of another stream, a CU mask) can keep part of the grid from being dispatched while the part
// that is resident spins: after `spin_limit` polls the abort word is raised and everybody leaves.  The reference's sampler never fails on a valid key
// (gpv.rs:152-161), so a wait that gave up must cost time, not the call: this kernel is enqueued behind every k_np_walk launch, returns at once while the abort
// word is 0, and otherwise walks the whole batch again in a form in which no workgroup ever waits for another: a sampler workgroup (4 G preimages) applies the
// updates of its OWN columns itself, left-looking -- before block J, the rows of block J take the blocks nblk-1 ... J+2 (descending; j ascending inside a block:
// the contract's chain, as plain fma instead of the MFMA K loop, same bits), block J+1 inside the sampler as always.  T comes from a fresh initial projection
// (k_np_project with the same predicate).  Slow (every workgroup reads all of g: ~10 ms at C2) and rare by construction.
template <int G>
__global__ __launch_bounds__(512, 4) void k_np_walk_solo(NpSampleArgs a, size_t dim, size_t nblk, uint64_t seed, uint32_t tag, uint64_t first_index, size_t B, const double* __restrict__ Gp,
                                                         double* __restrict__ Tm, const unsigned* __restrict__ only_if, unsigned long long* __restrict__ reruns) {
  extern __shared__ __attribute__((aligned(16))) unsigned char np_smem[];
  if (__hip_atomic_load(only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  const unsigned wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wg == 0 && tid == 0) atomicAdd(reruns, 1ull);
  long long zr[G];
#pragma unroll
  for (int s = 0; s < G; ++s) zr[s] = 0;
  // chain owner: wave w < 4 G holds preimage (4 wg) G + w, lane = row of the block
  const size_t b = (size_t)wg * 4 * G + (size_t)wave;
  const bool owner = wave < 4 * G && b < B;
  for (size_t J = nblk; J-- > 0;) {
    if (J + 2 < nblk) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // the z of the blocks above were stored write-through by this workgroup's samplers
      if (owner) {
        const size_t i = J * NP_NB + (size_t)lane;
        double t = Tm[i * a.ldt + b];                                     // the initial projection: untouched so far (rows of block J are written below, once)
        const double* zb = a.Zf + (b / TR_BN) * a.nkb * TR_CHUNK;
        const int bc = (int)(b % TR_BN);
        for (size_t Jb = nblk; Jb-- > J + 2;) {
          const double* gp = Gp + (np_panel_base(Jb) + (i / 128) * 4) * TR_CHUNK;
          const double* zp = zb + Jb * (NP_NB / 16) * TR_CHUNK;
#pragma unroll 4
          for (int kc = 0; kc < NP_NB / 16; ++kc)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
              t = fma(-zp[(size_t)kc * TR_CHUNK + tr_chunk_pos(bc, kk)], gp[(size_t)kc * TR_CHUNK + tr_chunk_pos((int)(i % 128), kk)], t);
        }
        __hip_atomic_store(Tm + i * a.ldt + b, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    np_sample_body<G, true, true>(np_smem, wg, a, dim, J, seed, tag, first_index, B, zr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// ---- the same walk where the sampler workgroups fill the chip (two per CU: C4, 4096 preimages) and T does not fit registers: k_np_walk2 -----------------------------
// No workgroup is left for updating, so the HELPER waves do it: a helper produces its sampler's Philox records ahead of time and used to sleep while the record ring
// was full -- now it applies published z to the running projections instead.  Each helper OWNS 16-row x 64-preimage tiles of its group's T (tile f of the group
// belongs to helper f mod HW, HW = 4 x the group's workgroups; the four tiles of a block to the four helpers of one workgroup), reads and writes them in memory with
// write-through accesses (nobody else touches them until the hand-over), and walks the published blocks in order: for block J every owned tile below block J - 1
// takes t = fma(-z_j, g[j][.], t), j ascending (one 16-step MFMA loop, operands straight from the fragment-ordered streams), the tiles of block J - 2 first -- they
// are complete then and their owner raises the block's flag.  Before its sampler starts block J a helper finishes everything up to the z of block J + 2 (a sampler of
// the group may be waiting for exactly those rows), so the chain of waits always ends at a workgroup that is still sampling.  Same chains, same bits.
#ifdef PSF_EXPERIMENTS   /* measured slower than one launch per block (6.97 against 4.49 ms at C4, profiles/r05_notes.md): comparison arm of the experiments build */
template <int G>
struct NpWalk2Jobs {
  const double* Gp; double* T; const double* Zf;
  const unsigned* zcount_g; unsigned* tready_g; unsigned* abort_w; int* s_arr;
  size_t ldt, nkb, col0, bj;
  unsigned need, spin_limit;
  int nblk, t0, hw, HW, lane;
  uint32_t t_lane, f_lane;
  int Jn, fi, zmin;
  bool aborted;

  __device__ __forceinline__ bool spin(const unsigned* word, unsigned target) {          // bounded; false = gave up (abort raised or seen)
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (aborted) return false;
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0 && __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { aborted = true; return false; }
      if (spins >= spin_limit) { __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; return false; }
    }
    return true;
  }
  __device__ __forceinline__ bool wait_word(const unsigned* w) { return spin(w, 1u); }
  __device__ __forceinline__ int imax(int J) const {                   // owned tiles hw + HW i that lie in blocks <= J - 2
    const int fmax = 4 * (J - 2) + 3;
    return fmax >= hw ? (fmax - hw) / HW : -1;
  }
  __device__ __forceinline__ double* t_at(size_t row, size_t col) const {
    return reinterpret_cast<double*>(reinterpret_cast<char*>(T + row * ldt + col) + t_lane);
  }
  // tile f (rows 16 f ... + 15, the group's 64 preimages) takes the z of block J.  Loads are PLAIN loads behind the one agent-scope acquire that follows the poll
  // of the block's counter (advance): hipcc keeps a dozen plain loads in flight, while it puts a full s_waitcnt vmcnt(0) behind every agent-scope atomic load
  // (measured: 50 us per tile).  The z of the block were stored write-through by their samplers before the counter moved; the tile itself is private to this wave
  // (its own earlier stores) until the hand-over, which stores it write-through.
  __device__ __forceinline__ void job(int J, int f) {
    const size_t i0 = (size_t)f * 16;
    const bool urgent = f / 4 == J - 2;
    d4 acc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nf][r] = *t_at(i0 + 4 * r, col0 + nf * 16);
    const double* ga = Gp + (np_panel_base((size_t)J) + (i0 / 128) * 4) * TR_CHUNK + ((i0 % 128) / 16) * 64;
    const double* zs = Zf + (bj * nkb + (size_t)J * (NP_NB / 16)) * TR_CHUNK + (size_t)t0 * 64;
#pragma unroll
    for (int kstep = 0; kstep < 16; ++kstep) {
      const size_t ko = (size_t)(kstep >> 2) * TR_CHUNK + (size_t)(kstep & 3) * 512;
      const double av = -*reinterpret_cast<const double*>(reinterpret_cast<const char*>(ga + ko) + f_lane);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const double bv = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(zs + ko + nf * 64) + f_lane);
        acc[nf] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[nf], 0, 0, 0);
      }
    }
    if (!urgent) {
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) *t_at(i0 + 4 * r, col0 + nf * 16) = acc[nf][r];
    } else {                                                           // the block after next is complete: hand it to the samplers
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) __hip_atomic_store(t_at(i0 + 4 * r, col0 + nf * 16), acc[nf][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const int prev = __hip_atomic_fetch_add(s_arr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((prev & 3) == 3) __hip_atomic_store(tready_g + (J - 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // one unit of work if there is one; `blocking`: wait for the block's z
  __device__ __forceinline__ bool advance(bool blocking) {
    if (Jn < 2) return false;
    if (fi == -2) {                                                    // block Jn not started: are its z published?
      if (zmin > Jn) {
        if (blocking) { if (!spin(zcount_g + Jn, need)) { /* aborted: go on without waiting */ } }
        else if (__hip_atomic_load(zcount_g + Jn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) return false;
        zmin = Jn;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // ONE acquire per block: this CU's L1 holds nothing older than the publication
      }
      fi = imax(Jn);
    }
    if (fi >= 0) { job(Jn, hw + HW * fi); --fi; }
    if (fi < 0) { --Jn; fi = -2; }
    return true;
  }
  __device__ __forceinline__ bool step() { return advance(false); }
  __device__ __forceinline__ void drain() { while (advance(false)) {} }
  __device__ __forceinline__ void finish_through(int Jt) { while (Jn >= Jt && Jn >= 2) advance(true); }
};

// grid: the nS sampler workgroups, two per CU, all resident at once
template <int G>
__global__ __launch_bounds__(512, 4) void k_np_walk2(NpSampleArgs a, size_t dim, size_t nblk, uint64_t seed, uint32_t tag, uint64_t first_index, size_t B, unsigned nS,
                                                     const double* __restrict__ Gp, double* __restrict__ Tm, NpWalkSync sy) {
  extern __shared__ __attribute__((aligned(16))) unsigned char np_smem[];
  const unsigned wg = blockIdx.x;
  const unsigned per = (unsigned)(NP_GW / (4 * G)), grp = wg / per, wl = wg % per, first = grp * per, need = nS - first < per ? nS - first : per;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool helper = wave >= 4;
  int* s_arr = reinterpret_cast<int*>(np_smem + 65536 - 32);
  if (tid == 0) *s_arr = 0;
  __syncthreads();
  NpWalk2Jobs<G> jobs;
  jobs.Gp = Gp; jobs.T = Tm; jobs.Zf = a.Zf;
  jobs.zcount_g = sy.zcount + (size_t)grp * sy.nblk_stride; jobs.tready_g = sy.tready + (size_t)grp * sy.nblk_stride; jobs.abort_w = sy.abort; jobs.s_arr = s_arr;
  jobs.ldt = a.ldt; jobs.nkb = a.nkb; jobs.col0 = (size_t)grp * NP_GW; jobs.bj = jobs.col0 / TR_BN; jobs.t0 = (int)((jobs.col0 % TR_BN) / 16);
  jobs.need = need; jobs.spin_limit = sy.spin_limit; jobs.nblk = (int)nblk; jobs.hw = (int)(4 * wl) + (wave & 3); jobs.HW = (int)(4 * need); jobs.lane = lane;
  jobs.t_lane = (uint32_t)(((size_t)(lane >> 4) * a.ldt + (size_t)(lane & 15)) * sizeof(double)); jobs.f_lane = (uint32_t)(lane * sizeof(double));
  jobs.Jn = (int)nblk - 1; jobs.fi = -2; jobs.zmin = (int)nblk; jobs.aborted = false;
  long long zr[G];
#pragma unroll
  for (int s = 0; s < G; ++s) zr[s] = 0;
  for (size_t J = nblk; J-- > 0;) {
    if (helper) jobs.finish_through((int)J + 2);
    const unsigned* tw = J + 2 < nblk ? jobs.tready_g + J : nullptr;
    np_sample_body<G, true, true, NpWalk2Jobs<G>>(np_smem, wg, a, dim, J, seed, tag, first_index, B, zr, &jobs, tw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && J >= 2) __hip_atomic_fetch_add(sy.zcount + (size_t)grp * sy.nblk_stride + J, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane == 0 && __hip_atomic_load(sy.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) atomicOr(a.flags, 1);
}
#endif

// E[b][j] (+)= scale * sum_i Z8[i][b] B8[j][i] on the int8 matrix cores (v_mfma_i32_16x16x64_i8): Z as the A operand (rows = preimages),
// the basis plane as the B operand (columns = coordinates j), so 16 lanes hold 16 consecutive j of one preimage and the int64 stores
// are 128-byte runs.  Workgroup tile 128 (b) x 128 (j), K = 128 per stage, two LDS stages filled by LDS-DMA.  int32 accumulation is
// exact (at most 2^17 terms of |z b| <= 2^14).  `gate`: the pass is skipped unless *gate != 0 (hi plane of z, decided on device).
#ifdef PSF_EXPERIMENTS   /* one launch per digit pair: replaced by k_np_combine8_fused (round 5), comparison arm of the experiments build */
template <bool ACCUM>
__global__ __launch_bounds__(256, 2) void k_np_combine8(const int8_t* __restrict__ B8, size_t ldb, size_t d, int nk128, const int8_t* __restrict__ Z8, size_t ld,
                                                        size_t B, long long scale, const int* __restrict__ gate, int64_t* __restrict__ E, size_t lde) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
  if (gate && !*gate) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t b0 = (size_t)blockIdx.x * 128, i0 = (size_t)blockIdx.y * 128;
  v4i acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = v4i{0, 0, 0, 0};
  auto stage_load = [&](int ks2, int buf) {
    unsigned char* base = rc_smem + buf * 32768;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = (wave * 4 + j) * 64 + lane;                  // 16-byte piece, 0..1023
      const int kk = p >> 9, row = (p >> 2) & 127, col = ((p & 3) - (row >> 2)) & 3;      // k groups rotated by row / 4 in LDS (i8_slot, psf_kernels.hpp): no bank conflicts
      __builtin_amdgcn_global_load_lds(B8 + (i0 + (size_t)row) * ldb + (size_t)ks2 * 128 + kk * 64 + col * 16,
                                       (lds_void_ptr)(base + (wave * 4 + j) * 1024), 16, 0, 0);
      const int kg = p >> 7, bb = p & 127;
      __builtin_amdgcn_global_load_lds(Z8 + (((size_t)ks2 * 8 + kg) * ld + b0 + (size_t)bb) * 16,
                                       (lds_void_ptr)(base + 16384 + (wave * 4 + j) * 1024), 16, 0, 0);
    }
  };
  const int r16 = lane & 15, g = lane >> 4;
  stage_load(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks2 = 0; ks2 < nk128; ++ks2) {
    const int cb = ks2 & 1;
    if (ks2 + 1 < nk128) stage_load(ks2 + 1, cb ^ 1);
    const unsigned char* sR = rc_smem + cb * 32768;
    const unsigned char* sL = sR + 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      v4i fr[4], fl[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fr[t] = *reinterpret_cast<const v4i*>(sR + kk * 8192 + ((wc * 64 + t * 16 + r16) * 64 + i8_slot(wc * 64 + t * 16 + r16, g) * 16));
        fl[t] = *reinterpret_cast<const v4i*>(sL + (((kk * 4 + g) * 128 + wr * 64 + t * 16 + r16) * 16));
      }
#pragma unroll
      for (int bt = 0; bt < 4; ++bt)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[bt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fl[bt], fr[it], acc[bt][it], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // C/D map: column (coordinate j) = lane & 15, row (preimage) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int bt = 0; bt < 4; ++bt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t bb = b0 + wr * 64 + bt * 16 + 4 * g + r, jj = i0 + wc * 64 + it * 16 + r16;
        if (bb < B && jj < d) {
          int64_t* p = E + bb * lde + jj;
          const int64_t v = scale * (long long)acc[bt][it][r];
          *p = ACCUM ? *p + v : v;
        }
      }
}
#endif

// ---- the recombination in ONE launch (round 5) ---------------------------------------------------------------------------------------------
// Six digit-pair products (three z digits x two basis digits) used to be six launches, each a read-modify-write of the whole of E; measured at C2 five of
// them ran at full length although the second basis digit holds a handful of entries (|b| > 127 is a 5-sigma event of R W) and the third z digit a few rows.
// k_np_occ_* record which 128 x 128 tiles of the digit planes hold anything at all (the basis once per key, z digits 1 and 2 once per call; digit 0 is
// taken as dense); k_np_combine8_fused walks, per output tile, the list of (pair, K block) items whose two tiles are both occupied, ordered by the
// pair's scale 2^(8 (zi + bi)), folds the int32 accumulator into 64 bits whenever the scale changes and writes E once.  Same integers as the six
// launches: every skipped item is a product with an all-zero tile.
__global__ __launch_bounds__(256) void k_np_occ_z(const int8_t* __restrict__ Z8, size_t zplane, size_t ld, int nk128, unsigned char* __restrict__ occ /*[2][ld / 128][nk128]; the grid covers the column blocks in use*/) {
  const int ks2 = (int)(blockIdx.x % (unsigned)nk128), bblk = (int)(blockIdx.x / (unsigned)nk128);
  const int8_t* src = Z8 + (size_t)(blockIdx.y + 1) * zplane;
  int any = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + (int)threadIdx.x, kg = p >> 7, bb = p & 127;
    const v4i v = *reinterpret_cast<const v4i*>(src + (((size_t)ks2 * 8 + kg) * ld + (size_t)bblk * 128 + bb) * 16);
    any |= v[0] | v[1] | v[2] | v[3];
  }
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) occ[(size_t)blockIdx.y * (ld / 128) * nk128 + blockIdx.x] = any != 0;
}
__global__ __launch_bounds__(256) void k_np_occ_basis(const int8_t* __restrict__ B8, size_t dpad, int nk128, unsigned char* __restrict__ occ /*[2][dpad / 128][nk128]*/) {
  const int ks2 = (int)(blockIdx.x % (unsigned)nk128), iblk = (int)(blockIdx.x / (unsigned)nk128);
  const int8_t* src = B8 + (size_t)blockIdx.y * dpad * dpad;
  int any = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + (int)threadIdx.x, row = p >> 3, seg = p & 7;
    const v4i v = *reinterpret_cast<const v4i*>(src + ((size_t)iblk * 128 + row) * dpad + (size_t)ks2 * 128 + seg * 16);
    any |= v[0] | v[1] | v[2] | v[3];
  }
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) occ[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = any != 0;
}

__global__ __launch_bounds__(256, 2) void k_np_combine8_fused(const int8_t* __restrict__ B8, size_t ldb, size_t d, int nk128, int nbdig, const unsigned char* __restrict__ bocc,
                                                              const int8_t* __restrict__ Z8, size_t zplane, size_t ld, size_t B, const unsigned char* __restrict__ zocc,
                                                              int64_t* __restrict__ E, size_t lde) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];      // two stages of 32 KiB, then the item list of one chunk of 64 K blocks
  unsigned short* const list = reinterpret_cast<unsigned short*>(rc_smem + 65536);
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t b0 = (size_t)blockIdx.x * 128, i0 = (size_t)blockIdx.y * 128;
  const size_t nbb = ld / 128, nrb = ldb / 128;
  v4i acc[4][4];
  long long tot[4][4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      acc[x][y] = v4i{0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[x][y][r] = 0;
    }
  // pair p: z digit / basis digit, ordered by scale zi + bi
  auto zi_of = [](int p) { return (0x212010 >> (4 * p)) & 3; };      // 0 1 0 2 1 2
  auto bi_of = [](int p) { return (0x110100 >> (4 * p)) & 1; };      // 0 0 1 0 1 1
  const int r16 = lane & 15, g = lane >> 4;
  for (int c0 = 0; c0 < nk128; c0 += 64) {
    if (wave == 0) {
      const int ks = c0 + lane;
      const bool in = ks < nk128;
      unsigned long long zm[3], bm[2];
      zm[0] = __builtin_amdgcn_ballot_w64(in);
      zm[1] = __builtin_amdgcn_ballot_w64(in && zocc[((size_t)0 * nbb + blockIdx.x) * nk128 + ks] != 0);
      zm[2] = __builtin_amdgcn_ballot_w64(in && zocc[((size_t)1 * nbb + blockIdx.x) * nk128 + ks] != 0);
      bm[0] = __builtin_amdgcn_ballot_w64(in && bocc[((size_t)0 * nrb + blockIdx.y) * nk128 + ks] != 0);
      bm[1] = nbdig > 1 ? __builtin_amdgcn_ballot_w64(in && bocc[((size_t)1 * nrb + blockIdx.y) * nk128 + ks] != 0) : 0ull;
      int base = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const unsigned long long m = zm[zi_of(p)] & bm[bi_of(p)];
        if ((m >> lane) & 1ull) list[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned short)((p << 8) | lane);
        base += __builtin_popcountll(m);
      }
      if (lane == 0) s_n = base;
    }
    __syncthreads();
    const int n = s_n;
    auto stage_load = [&](int item, int buf) {
      const int p = item >> 8, ks2 = c0 + (item & 255);
      const int8_t* Bp = B8 + (size_t)bi_of(p) * ldb * ldb;
      const int8_t* Zp = Z8 + (size_t)zi_of(p) * zplane;
      unsigned char* base = rc_smem + buf * 32768;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = (wave * 4 + j) * 64 + lane;                  // 16-byte piece, 0..1023
        const int kk = q >> 9, row = (q >> 2) & 127, col = ((q & 3) - (row >> 2)) & 3;      // as k_np_combine8
        __builtin_amdgcn_global_load_lds(Bp + (i0 + (size_t)row) * ldb + (size_t)ks2 * 128 + kk * 64 + col * 16, (lds_void_ptr)(base + (wave * 4 + j) * 1024), 16, 0, 0);
        const int kg = q >> 7, bb = q & 127;
        __builtin_amdgcn_global_load_lds(Zp + (((size_t)ks2 * 8 + kg) * ld + b0 + (size_t)bb) * 16, (lds_void_ptr)(base + 16384 + (wave * 4 + j) * 1024), 16, 0, 0);
      }
    };
    if (n > 0) {
      int item = __builtin_amdgcn_readfirstlane((int)list[0]);
      stage_load(item, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int t = 0; t < n; ++t) {
        const int cb = t & 1;
        const int next = t + 1 < n ? __builtin_amdgcn_readfirstlane((int)list[t + 1]) : -1;
        if (next >= 0) stage_load(next, cb ^ 1);
        const unsigned char* sR = rc_smem + cb * 32768;
        const unsigned char* sL = sR + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          v4i fr[4], fl[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            fr[u] = *reinterpret_cast<const v4i*>(sR + kk * 8192 + ((wc * 64 + u * 16 + r16) * 64 + i8_slot(wc * 64 + u * 16 + r16, g) * 16));
            fl[u] = *reinterpret_cast<const v4i*>(sL + (((kk * 4 + g) * 128 + wr * 64 + u * 16 + r16) * 16));
          }
#pragma unroll
          for (int bt = 0; bt < 4; ++bt)
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[bt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fl[bt], fr[it], acc[bt][it], 0, 0, 0);
        }
        const int sc = zi_of(item >> 8) + bi_of(item >> 8);
        const int scn = next >= 0 ? zi_of(next >> 8) + bi_of(next >> 8) : -1;
        if (sc != scn) {      // at most 64 K blocks x 2 pairs x 128 terms of |z b| <= 2^14 since the last fold: 2^28, exact in int32
          const long long scale = 1ll << (8 * sc);
#pragma unroll
          for (int bt = 0; bt < 4; ++bt)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
#pragma unroll
              for (int r = 0; r < 4; ++r) tot[bt][it][r] += scale * (long long)acc[bt][it][r];
              acc[bt][it] = v4i{0, 0, 0, 0};
            }
        }
        item = next;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    __syncthreads();      // the list is rebuilt by the next chunk
  }
  // C/D map: column (coordinate j) = lane & 15, row (preimage) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int bt = 0; bt < 4; ++bt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t bb = b0 + wr * 64 + bt * 16 + 4 * g + r, jj = i0 + wc * 64 + it * 16 + r16;
        if (bb < B && jj < d) E[bb * lde + jj] = tot[bt][it][r];
      }
}

// holds its stream for `ticks` of the 100 MHz wall clock (one wave): the second half of a two-halves call starts half a block time behind the first, so that the
// update tiles at the end of one half's launches fall into the middle of the other half's sampling instead of colliding with its own phase twin
__global__ void k_np_delay(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// e[b][piv[r]] += sol[r][b]  (gpv.rs:160: sol + sample; e = -(c0 - sum z b) with c0 = -sol)
__global__ void k_np_add_sol(const uint64_t* __restrict__ Sol, const uint32_t* __restrict__ piv, size_t n, size_t B, size_t ld, int64_t* __restrict__ E, size_t lde) {
  const size_t total = n * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t r = g / B, b = g % B;
    E[b * lde + piv[r]] += (int64_t)Sol[r * ld + b];
  }
}

// ---- second pass of the two-pass walk (large moduli, psfgpv_impl.hpp): centre c1 = -e1 over ALL d coordinates --------------------------------
// C1 chunk (bj, kc), kc < nkd: (double)(-e1[b][16 kc + kk]) -- the B operand of the initial projection T = B~ C1 (K = d).  |e1| < 2^53 or flags[0].
__global__ void k_np_center_from_e(const int64_t* __restrict__ E1, size_t d, size_t B, size_t ld, size_t nkd, double* __restrict__ C1, int* __restrict__ flags) {
  const size_t K = nkd * 16, total = K * ld;
  int bad = 0;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t j = g % K, b = g / K;
    const int64_t v = (j < d && b < B) ? E1[b * d + j] : 0;
    if (v >= (1ll << 53) || v <= -(1ll << 53)) bad = 1;
    const size_t chunk = (b / TR_BN) * nkd + j / 16;
    C1[chunk * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(j % 16))] = -(double)v;
  }
  if (bad) atomicOr(flags, 1);
}
// e = e1 + sum z'_i b_i
__global__ void k_np_add_e1(const int64_t* __restrict__ E1, size_t total, int64_t* __restrict__ E) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) E[g] += E1[g];
}

// The same recombination in 64-bit integers, one output per thread: for a basis or a z that does not fit two int8 digits.
// Runs after everything else and rewrites E completely; `gate` as above (nullptr = always).
__global__ void k_np_combine_generic(const int32_t* __restrict__ St, size_t d, const double* __restrict__ Zf, size_t nkb, const uint64_t* __restrict__ Sol,
                                     const uint32_t* __restrict__ piv, size_t n, size_t B, size_t ld, const int* __restrict__ gate,
                                     int64_t* __restrict__ E, size_t lde) {
  if (gate && !*gate) return;
  const size_t total = d * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / d, j = g % d;
    int64_t acc = 0;
    const double* zc = Zf + (b / TR_BN) * nkb * TR_CHUNK;
    for (size_t i = 0; i < d; ++i) {
      const double z = zc[(i / 16) * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(i % 16))];
      acc += (int64_t)z * (int64_t)St[i * d + j];
    }
    E[b * lde + j] = acc;
  }
  __threadfence();
  // the pivot columns receive sol afterwards (same thread set: every (b, j) pair is owned by exactly one thread)
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / d, j = g % d;
    for (size_t r = 0; r < n; ++r)
      if (piv[r] == j) E[b * lde + j] += (int64_t)Sol[r * ld + b];
  }
}

}  // namespace psf
