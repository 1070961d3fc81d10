// psf_kernels.hpp -- HIP kernels of the PSFPerturbation::samp_p hot path (gfx950 / CDNA4 only).
//
// Data layout in HBM (DESIGN.md "Data layout"):  every per-batch matrix is COORDINATE-major with the batch
// index contiguous ("[coord][b]", row stride Bs = batch padded to 128): lanes of a wave walk over preimages,
// so key material (A, R, sqrt(Sigma_2), gadget tables) is wave-uniform and batch data is coalesced.
// The two operands of the dominant kernel (x = sqrt(Sigma_2) d) are stored in MFMA-fragment order so that a
// 16 KiB chunk is one contiguous, fully coalesced global read and every LDS fragment read is a conflict-free
// ds_read_b64.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "psf_rng.hpp"

namespace psf {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// ---- geometry of the triangular product -------------------------------------------------------------
constexpr int TR_BM = 128;      // rows of sqrt(Sigma_2) per workgroup
constexpr int TR_BN = 128;      // preimages per workgroup
#ifndef PSF_TR_BK
#define PSF_TR_BK 16
#endif
constexpr int TR_BK = PSF_TR_BK;  // coordinates per staged chunk (TR_BK / 4 MFMA k-steps)
constexpr int TR_CHUNK = TR_BM * TR_BK;            // doubles per chunk (2048 = 16 KiB)
constexpr int TR_KB_PER_BLOCK = TR_BM / TR_BK;     // 8

// first chunk of row-block bi in the packed lower-triangular tile stream: sum_{t<bi} 8 (t+1)
__host__ __device__ inline size_t tr_rowblock_base(size_t bi) { return (size_t)TR_KB_PER_BLOCK * bi * (bi + 1) / 2; }
__host__ __device__ inline size_t tr_total_chunks(size_t nbi) { return (size_t)TR_KB_PER_BLOCK * nbi * (nbi + 1) / 2; }

// position inside a chunk of element (r in [0,128), kk in [0,16)):  [ks = kk/4][tile = r/16][lane = (kk%4)*16 + r%16]
__host__ __device__ inline int tr_chunk_pos(int r, int kk) {
  return ((kk >> 2) * 8 + (r >> 4)) * 64 + ((kk & 3) << 4) + (r & 15);
}

// ---- L repack: row-major lower triangle -> fragment-ordered chunk stream ------------------------------
// PACKED: source row i starts at i(i+1)/2 (host key format); otherwise dense with leading dimension ld.
template <bool PACKED>
__global__ void k_repack_L(const double* __restrict__ src, size_t ld, size_t m, double* __restrict__ dst, size_t nbi) {
  const size_t total = tr_total_chunks(nbi) * TR_CHUNK;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t chunk = g / TR_CHUNK;
    const int pos = (int)(g % TR_CHUNK);
    // invert chunk -> (bi, bk): bi = largest with 4 bi (bi+1) <= chunk
    size_t bi = (size_t)((sqrt(1.0 + 8.0 * (double)chunk / TR_KB_PER_BLOCK) - 1.0) * 0.5);
    while (tr_rowblock_base(bi + 1) <= chunk) ++bi;
    while (tr_rowblock_base(bi) > chunk) --bi;
    const size_t bk = chunk - tr_rowblock_base(bi);
    const int ks = pos >> 9, tile = (pos >> 6) & 7, lane = pos & 63;
    const size_t row = bi * TR_BM + tile * 16 + (lane & 15);
    const size_t col = bk * TR_BK + ks * 4 + (lane >> 4);
    double v = 0.0;
    if (row < m && col <= row) v = PACKED ? src[row * (row + 1) / 2 + col] : src[row * ld + col];
    dst[g] = v;
  }
}

// inverse direction (export): chunk stream -> packed rows
// entries [first, first + total) of the packed triangle (a whole key: first = 0, total = m(m+1)/2; a row block otherwise)
__global__ void k_unpack_L(const double* __restrict__ chunks, size_t first, size_t total, double* __restrict__ packed) {
  for (size_t g0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g0 < total; g0 += (size_t)gridDim.x * blockDim.x) {
    const size_t g = first + g0;
    size_t row = (size_t)((sqrt(1.0 + 8.0 * (double)g) - 1.0) * 0.5);
    while ((row + 1) * (row + 2) / 2 <= g) ++row;
    while (row * (row + 1) / 2 > g) --row;
    const size_t col = g - row * (row + 1) / 2;
    const size_t bi = row / TR_BM, bk = col / TR_BK;
    const size_t chunk = tr_rowblock_base(bi) + bk;
    packed[g0] = chunks[chunk * TR_CHUNK + tr_chunk_pos((int)(row % TR_BM), (int)(col % TR_BK))];
  }
}

// ---- standard normals, written straight into the B-operand chunk stream ------------------------------
// chunk (bj, bk) at (bj * nkb + bk); element (kk, c) -> coordinate bk*16+kk of preimage bj*128+c
// Wave-compacted: a wave owns NR_SEG consecutive positions of the chunk stream; every iteration each lane evaluates
// one ratio-of-uniforms attempt of its current position, accepted lanes store and take the next unassigned position
// (ballot + mbcnt).  The value of a position is the first accepted attempt of its own (coordinate, preimage) Philox stream.
constexpr int NR_SEG = 4096;   // two chunks
// positions per wave: NR_SEG for full batches; with few positions in all (a single call) the launch lasts as long as its longest wave
__host__ inline uint32_t nr_segment(size_t total) {
  size_t seg = (total / 4096 + 63) / 64 * 64;
  if (seg < 128) seg = 128;
  return (uint32_t)(seg > (size_t)NR_SEG ? (size_t)NR_SEG : seg);
}
__device__ inline int lane_rank(uint64_t mask);
// o / d and o % d for o d < 2^32 by one high multiply: M = ceil(2^32 / d) (d >= 2; M = 0 stands for d = 1).  o M / 2^32 = o / d + o e / (d 2^32) with e < d, so the
// floor is that of o / d as long as o e < 2^32.  (The kernels below locate a sample (coordinate, preimage) from its offset in a segment: o < 2^15, d < 2^13; a 32-bit
// division is ~40 vector instructions, twice per sample.)
__host__ __device__ inline uint32_t udiv_magic_of(uint32_t d) { return d < 2 ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1) / d); }
__device__ inline void udivmod_magic(uint32_t o, uint32_t d, uint32_t M, uint32_t* q, uint32_t* r) {
  const uint32_t qq = M ? __umulhi(o, M) : o;
  *q = qq;
  *r = o - qq * d;
}

// Structured sqrt(Sigma_2) (opt-in, struct NormalsFixed): the coordinates from `split` on (the gadget half, d_2) are taken in FIXED POINT,
// d = q 2^-32 with q = floor(n 2^32 + 1/2) for the drawn normal n -- so that R d_2 is an exact integer sum on the int8 matrix cores.
// For those coordinates the kernel also writes the five balanced base-256 digits of q as int8 planes [c/16][b][16] (c = coordinate - split)
// and the finished centre x = h d into X (the bottom block of the factor is a multiple of the identity).
struct NormalsFixed { size_t split; int8_t* planes; size_t plane_bytes; size_t ld; double* X; double h; };
constexpr int kFixPlanes = 5;      // |q| < 2^39: normals beyond +-64 do not occur

// ncf > 0: the COMPACT stream of small batches (k_trmm_stream, psf_stream_kernels.hpp): [k-step][column fragment of 16 preimages, ncf of them][lane],
// position g -> coordinate 4 (g / (64 ncf)) + (g % 64) / 16, preimage 16 ((g / 64) % ncf) + g % 16 -- only the fragments in use exist, so a single
// call draws m normals among m_pad x 16 positions instead of m_pad x 128.  ncf = 0x100 + bc: the DENSE stream of a call with at most bc <= 16 preimages,
// [k-step][preimage < bc][k % 4] (k_trmm_stream CD == 2): m_pad x bc positions, no padding columns at all.
__global__ __launch_bounds__(256) void k_normals_wave(uint64_t seed, uint64_t first_index, size_t m, size_t B, size_t nkb, size_t nbj,
                                                      double* __restrict__ Dt, int* __restrict__ fail, NormalsFixed fx, uint32_t ncf, uint32_t seg) {
  const int lane = threadIdx.x & 63;
  const size_t total = ncf >= 0x100u ? nkb * 4 * 4 * (size_t)(ncf - 0x100u) : ncf ? nkb * 4 * (size_t)ncf * 64 : nbj * nkb * TR_CHUNK;
  const size_t seg0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * seg;     // seg <= NR_SEG positions per wave (nr_segment: short for small batches)
  if (seg0 >= total) return;
  const size_t seg1 = seg0 + seg < total ? seg0 + seg : total;
  const size_t chunk0 = seg0 / TR_CHUNK;
  constexpr int NCH = NR_SEG / TR_CHUNK + 1;
  size_t cbj[NCH], cbk[NCH];                       // (column block, K block) of the chunks this wave touches
#pragma unroll
  for (int i = 0; i < NCH; ++i) { cbj[i] = (chunk0 + i) / nkb; cbk[i] = (chunk0 + i) % nkb; }
  // (the small-batch layouts: positions below 2^32, bc a power of two, ncf <= 128 -- shifts and one high multiply instead of 64-bit divisions, which were two per position)
  const uint32_t lbc = ncf >= 0x100u ? (uint32_t)__builtin_ctz(ncf - 0x100u) : 0u;
  const uint32_t ncf_magic = ncf && ncf < 0x100u ? udiv_magic_of(ncf) : 0u;
  auto locate = [&](size_t g, size_t* coord, size_t* b) {
    if (ncf >= 0x100u) {                                       // dense stream of <= 16 preimages: [k-step][preimage < bc][k % 4], bc = ncf - 0x100 = 2^lbc
      const uint32_t g32 = (uint32_t)g;
      *coord = (size_t)((g32 >> (2 + lbc)) * 4 + (g32 & 3));
      *b = (size_t)((g32 >> 2) & ((1u << lbc) - 1u));
      return;
    }
    if (ncf) {
      const uint32_t ln = (uint32_t)(g & 63);
      const uint32_t fr = (uint32_t)(g >> 6);                  // fragment index = k-step * ncf + column fragment (< 2^21)
      uint32_t ks, cf;
      udivmod_magic(fr, ncf, ncf_magic, &ks, &cf);
      *coord = (size_t)(ks * 4 + (ln >> 4));
      *b = (size_t)(cf * 16 + (ln & 15));
      return;
    }
    const int ci = (int)(g / TR_CHUNK - chunk0);
    const int pos = (int)(g % TR_CHUNK);
    size_t bj = cbj[0], bk = cbk[0];
#pragma unroll
    for (int i = 1; i < NCH; ++i) if (ci == i) { bj = cbj[i]; bk = cbk[i]; }
    const int ks = pos >> 9, tile = (pos >> 6) & 7, ln = pos & 63;
    *coord = bk * TR_BK + ks * 4 + (ln >> 4);
    *b = bj * TR_BN + tile * 16 + (ln & 15);
  };
  size_t my = seg0 + lane, next_free = seg0 + 64;
  bool active = my < seg1;
  size_t coord = 0, b = 0;
  uint32_t t = 0;
  int f = 0;
  if (active) locate(my, &coord, &b);
  while (__ballot(active)) {
    bool accept = false;
    double v = 0.0;
    if (active) {
      if (coord >= m || b >= B) accept = true;                      // padding of the operand: exact zero, nothing to draw
      else {
        const uint64_t index = first_index + b;
        const U4 w = philox(seed, (uint32_t)coord, (uint32_t)index, t, tag_word(TAG_NORMAL, index));
        const double u = (double)(((((uint64_t)w.y << 32) | w.x) >> 11) + 1) * 0x1.0p-53;
        const uint64_t vv = (((uint64_t)w.w << 32) | w.z) >> 12;
        const double x = (((double)(2 * vv + 1) * 0x1.0p-52 - 1.0) * 0.8577638849607068) / u;
        {
          // fp32 screen of u <= exp(-x^2 / 4) (round 5): v_exp_f32 of the fp32-rounded argument is within 1e-5 relative of det_exp for x^2 / 4 <= 88, u converts with
          // 6e-8; beyond, both sides of the comparison are decided by the flush to zero (u >= 2^-53).  Only the 2e-4 band around the threshold pays the f64 exponential.
          const float xf = (float)x, uf = (float)u;
          const float e32 = __builtin_amdgcn_exp2f(-0.36067376f * (xf * xf));
          const bool sure = uf <= e32 * 0.9998f, never = uf > e32 * 1.0002f;
          accept = sure;
          if (!sure && !never) accept = u <= det_exp(-0.25 * (x * x));
        }
        v = x;
        if (!accept && ++t >= kMaxAttempts) { accept = true; f = 1; v = 0.0; }
        if (accept && fx.planes && coord >= fx.split) {              // fixed-point coordinate of the structured factor
          const double sc = floor(v * 0x1.0p32 + 0.5);
          if (!(fabs(sc) < 0x1.0p38)) f = 1;                         // five balanced digits cover |q| < 2^39
          long long qv = (long long)sc;
          v = sc * 0x1.0p-32;
          const size_t c = coord - fx.split;
          const size_t addr = ((c >> 4) * fx.ld + b) * 16 + (c & 15);
#pragma unroll
          for (int pl = 0; pl < kFixPlanes; ++pl) {
            const long long dg = (long long)(int8_t)(qv & 0xff);
            fx.planes[(size_t)pl * fx.plane_bytes + addr] = (int8_t)dg;
            qv = (qv - dg) >> 8;
          }
          fx.X[coord * fx.ld + b] = fx.h * v;
        }
      }
      if (accept) Dt[my] = v;
    }
    const uint64_t mask = __ballot(accept);
    if (mask) {
      const size_t nid = next_free + (size_t)lane_rank(mask);
      if (accept) {
        my = nid;
        active = nid < seg1;
        t = 0;
        if (active) locate(nid, &coord, &b);
      }
      next_free += (size_t)__popcll(mask);
    }
  }
  if (f) atomicOr(fail, 1);
}

// int64 -> int32 rows for the inter-GPU gather; two values per thread-iteration, 16-byte loads
__global__ __launch_bounds__(256) void k_narrow_rows(const int64_t* __restrict__ src, int32_t* __restrict__ dst, size_t count, int* __restrict__ overflow) {
  const size_t stride = (size_t)gridDim.x * 256;
  int bad = 0;
  const size_t pairs = count / 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += stride) {
    const longlong2 v = reinterpret_cast<const longlong2*>(src)[i];
    bad |= (v.x != (int64_t)(int32_t)v.x) | (v.y != (int64_t)(int32_t)v.y);
    reinterpret_cast<int2*>(dst)[i] = make_int2((int32_t)v.x, (int32_t)v.y);
  }
  if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t v = src[count - 1];
    bad |= v != (int64_t)(int32_t)v;
    dst[count - 1] = (int32_t)v;
  }
  if (bad) atomicOr(overflow, 1);
}

// targets: pinned host memory -> device by a kernel on the compute stream.  An SDMA / second-stream upload of the next call's targets is queued by
// the runtime behind the previous call's download (measured with rocprofv3: the upload ran 25 ms late and the compute stream idled), a kernel in
// stream order is not.  16-byte loads over PCIe, 16 MB at C3.
__global__ __launch_bounds__(256) void k_copy_words(const uint64_t* __restrict__ host_src, uint64_t* __restrict__ dst, size_t count) {
  const size_t stride = (size_t)gridDim.x * 256, pairs = count / 2;
  typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += stride)
    reinterpret_cast<u2*>(dst)[i] = __builtin_nontemporal_load(reinterpret_cast<const u2*>(host_src) + i);
  if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[count - 1] = host_src[count - 1];
}

// read one normal back out of the chunk stream (stage export)
__global__ void k_export_normals(const double* __restrict__ Dt, size_t m, size_t B, size_t nkb, double* __restrict__ out /*B x m*/, uint32_t ncf) {
  const size_t total = m * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / m, coord = g % m;
    if (ncf >= 0x100u) { out[g] = Dt[((coord / 4) * (ncf - 0x100u) + b) * 4 + coord % 4]; continue; }      // dense stream of <= 16 preimages
    if (ncf) { out[g] = Dt[((coord / 4) * ncf + b / 16) * 64 + (coord % 4) * 16 + b % 16]; continue; }      // compact stream of small batches
    const size_t chunk = (b / TR_BN) * nkb + coord / TR_BK;
    out[g] = Dt[chunk * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(coord % TR_BK))];
  }
}

// ---- X = L * D with v_mfma_f64_16x16x4_f64 -----------------------------------------------------------
// One workgroup (4 waves, 2x2) owns a 128 x 128 tile of X; wave tile 64 x 64 = 4x4 MFMA tiles.
// Per element the accumulation is a single fma chain in ascending coordinate order starting from +0
// (measured on gfx950: the instruction itself is an ascending-k fma chain, profiles/r01_probe_mfma_f64.log),
// which is the summation order of the contract.  Zero entries above the diagonal only add exact zeros.
//
// Workgroup order: heavy row-blocks first; blocks that share an XCD (blockIdx % 8) walk an 8 x 8 super-tile of
// (row-block, column-block) pairs so that each staged chunk is reused from that XCD's L2.
__device__ inline void tr_map_block(unsigned id, int nbi, int nbj, int GR, int GC, int* bi, int* bj) {
  // super-tiles of GR x GC blocks (GR * GC = 64), enumerated with row groups descending
  const int ncg = (nbj + GC - 1) / GC, nrg = (nbi + GR - 1) / GR;
  const unsigned xcd = id & 7u, slot = id >> 3;
  const unsigned group = (slot / (GR * GC)) * 8u + xcd;
  const unsigned t = slot % (GR * GC);
  const int rg = nrg - 1 - (int)(group / ncg), cg = (int)(group % ncg);
  *bi = rg * GR + (GR - 1 - (int)(t / GC));
  *bj = cg * GC + (int)(t % GC);
  if (rg < 0) *bi = -1;
}
__host__ inline unsigned tr_grid_size(int nbi, int nbj, int GR = 8, int GC = 8) {
  const int ncg = (nbj + GC - 1) / GC, nrg = (nbi + GR - 1) / GR;
  const unsigned groups = (unsigned)(ncg * nrg);
  const unsigned rounds = (groups + 7) / 8;
  return rounds * 8u * (unsigned)(GR * GC);
}

#ifdef TRMM_CLOCK_PROBE   /* measurement builds only (tools/trmm_clock_probe.py, tools/probe_trmm.hip): shader-clock ticks and 100 MHz real-time ticks per workgroup */
__device__ unsigned long long g_trmm_clk[4];
__device__ unsigned long long* g_trmm_log;      /* optional: per workgroup {start, end} in 100 MHz ticks */
#define TR_CLK_BEGIN const unsigned long long clk0 = clock64(), rt0 = wall_clock64();
#define TR_CLK_END if (threadIdx.x == 0) { const unsigned long long rt1 = wall_clock64(); atomicAdd(&g_trmm_clk[0], clock64() - clk0); atomicAdd(&g_trmm_clk[1], rt1 - rt0); atomicAdd(&g_trmm_clk[2], 1ull); \
    if (g_trmm_log) { g_trmm_log[2 * (size_t)blockIdx.x] = rt0; g_trmm_log[2 * (size_t)blockIdx.x + 1] = rt1; } }
#else
#define TR_CLK_BEGIN
#define TR_CLK_END
#endif
#ifdef PSF_EXPERIMENTS   /* round 1's LDS-staged product: lost to k_trmm_f64_big (-5 %), comparison arm of the experiments build */
__global__ __launch_bounds__(256, PSF_TR_BK == 16 ? 2 : 3) void k_trmm_f64(const double* __restrict__ Lt, const double* __restrict__ Dt,
                                                     double* __restrict__ X, int nbi, int nbj, size_t nkb, size_t ldx, int GR, int GC, size_t row_hi) {
  // LDS: 2 stages x (A chunk 2048 doubles | B chunk 2048 doubles); filled by LDS-DMA (global_load_lds_dwordx4), no
  // staging registers: the accumulators (128 VGPRs) leave no room to hold a chunk in flight (hipcc serialised
  // register-staged prefetch loads behind vmcnt(0) waits).
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int bi, bj;
  tr_map_block(blockIdx.x, nbi, nbj, GR, GC, &bi, &bj);
  if (bi < 0 || bi >= nbi || bj >= nbj) return;
  TR_CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = TR_KB_PER_BLOCK * (bi + 1);
  const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + lane * 2;
  const double* gB = Dt + (size_t)bj * nkb * TR_CHUNK + lane * 2;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  // each wave moves pieces 4*wave .. 4*wave+3 (1 KiB each) of both chunks: LDS address = wave-uniform base + lane*16
  auto stage_load = [&](int kb, int buf) {
#ifdef TRMM_FAKE_L2   /* timing experiment only: every load hits the same few chunks (results are garbage) */
    const double* ga = Lt + lane * 2 + (size_t)(kb & 7) * TR_CHUNK;
    const double* gb = Dt + lane * 2 + (size_t)(kb & 7) * TR_CHUNK;
#else
    const double* ga = gA + (size_t)kb * TR_CHUNK;
    const double* gb = gB + (size_t)kb * TR_CHUNK;
#endif
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
  // C/D map of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
  const size_t row0 = (size_t)bi * TR_BM + wr * 64, col0 = (size_t)bj * TR_BN + wc * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
        const size_t row = row0 + i * 16 + (lane >> 4) + 4 * r;      // rows from row_hi on are padding of the factor (structured mode: they belong to x_bot)
        if (row < row_hi) X[row * ldx + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
      }
  TR_CLK_END
}
#endif

// Same product, operands streamed from global memory straight into MFMA operand registers: the chunk streams are already in fragment order, so
// k-step s of a wave is eight 512-byte loads (4 A fragments, 4 B fragments) at stream offset 512 s.  No LDS, no barrier: the four waves of a
// workgroup only share cache lines.  TR_PD k-steps are in flight per wave.  Bit-identical to k_trmm_f64 (same ascending chains); PSF_TRMM_VARIANT=1
// (the default is k_trmm_f64_big below); measured on
// MI355X at C3: 54.0 vs 55.0 ms inside the library, 0.967 of the FP64 MFMA peak AT THE CLOCK THE KERNEL RUNS AT (2.25-2.35 GHz under this load;
// a loop with the same MFMAs and no loads at all reaches 0.963, profiles/r02_notes.md).
constexpr int TR_PD = 6;
// hipcc hoists plain loads of an unrolled prefetch ring to the top of the loop body and drains them with vmcnt(0) at its end, so the loads and
// their waits are written out: loads return in order, hence "all but the newest 8 (TR_PD - 1) have landed".
#define TR_LOAD8(dst, voff, base, imm) asm volatile("global_load_dwordx2 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(voff), "s"(base) : "memory")
#define TR_WAIT(n, A, Bv) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(Bv[0]), "+v"(Bv[1]), "+v"(Bv[2]), "+v"(Bv[3]))
#ifdef PSF_EXPERIMENTS   /* round 2's register-streamed product: lost to k_trmm_f64_big, comparison arm of the experiments build */
__global__ __launch_bounds__(256, 2) void k_trmm_f64_reg(const double* __restrict__ Lt, const double* __restrict__ Dt,
                                                         double* __restrict__ X, int nbi, int nbj, size_t nkb, size_t ldx, int GR, int GC, size_t row_hi) {
  int bi, bj;
  tr_map_block(blockIdx.x, nbi, nbj, GR, GC, &bi, &bj);
  if (bi < 0 || bi >= nbi || bj >= nbj) return;
  TR_CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nsteps = TR_KB_PER_BLOCK * (bi + 1) * (TR_BK / 4);
  const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK + (size_t)(wr * 4) * 64;      // wave-uniform
  const double* gB = Dt + (size_t)bj * nkb * TR_CHUNK + (size_t)(wc * 4) * 64;
  const uint32_t voff = (uint32_t)lane * 8u;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
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
  for (int u = 0; u < TR_PD; ++u) issue(a[u], b[u], u);                    // at least 32 steps per row-block
  for (int s0 = 0; s0 < nsteps; s0 += TR_PD) {
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
  // the re-reads behind the last step are never consumed: their destination registers must stay allocated until they have landed (a load that
  // lands in a register the compiler has reused since corrupts it -- seen as a memory fault through a clobbered store address)
#pragma unroll
  for (int u = 0; u < TR_PD; ++u) TR_WAIT(0, a[u], b[u]);
  const size_t row0 = (size_t)bi * TR_BM + wr * 64, col0 = (size_t)bj * TR_BN + wc * 64;
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
#endif

// The default since the end of round 2: ONE workgroup per CU, four waves, each alone on its SIMD with a 128 x 64 tile of X -- 256 AccVGPRs of
// accumulators (the MFMAs are asm statements: hipcc keeps builtin accumulators in architectural VGPRs) and the operand ring in the 256
// architectural ones.  Workgroup tile 256 x 128 = two row-blocks x one column block; wave (wr, wc) streams all 8 A fragments of row-block
// 2 bt + wr and the 4 B fragments of its half of the column block: 12 loads for 32 MFMAs per k-step, a quarter fewer operand bytes per flop than
// two 128 x 128 workgroups per CU, TR_BIG_PD k-steps in flight.  No co-resident workgroup: nothing competes for the matrix cores, so the 32
// workgroups of an XCD (a super-tile of 8 row tiles x 4 column blocks) run at one pace and share their fetches in L2.  Same ascending chains,
// same bits.  Measured at C3 (tools/probe_trmm.hip, profiles/r02_notes.md): 52.2 ms against 53.1-54.3 (k_trmm_f64_reg) and 54.9 (k_trmm_f64) in the
// same runs, 0.946 of the FP64 peak at 2.4 GHz, FETCH_SIZE 63 GB against 138 / 121.
constexpr int TR_BIG_PD = 4;
#ifndef TR_BIG_UNROLL
#define TR_BIG_UNROLL 2      /* rounds of the operand ring per loop iteration (must divide 8); measured 1 -> 2: +0.4 %, 4: no more */
#endif
#define TR_WAIT12(n, A, Bv) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), \
                                         "+v"(Bv[0]), "+v"(Bv[1]), "+v"(Bv[2]), "+v"(Bv[3]))
__global__ __launch_bounds__(256, 1) void k_trmm_f64_big(const double* __restrict__ Lt, const double* __restrict__ Dt,
                                                         double* __restrict__ X, int nbi, int nbj, size_t nkb, size_t ldx, int GR, int GC, size_t row_hi) {
  int bt, bj;
  const int nbt = (nbi + 1) / 2;                                     // row tiles of 256
  tr_map_block(blockIdx.x, nbt, nbj, GR, GC, &bt, &bj);
  if (bt < 0 || bt >= nbt || bj >= nbj) return;
  TR_CLK_BEGIN
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int bi = 2 * bt + wr;
  if (bi < nbi) {                                                    // an odd number of row-blocks leaves the last tile's lower half empty (no barrier in here)
    const int nsteps = TR_KB_PER_BLOCK * (bi + 1) * (TR_BK / 4);     // a multiple of 32
    const double* gA = Lt + tr_rowblock_base((size_t)bi) * TR_CHUNK;                    // wave-uniform
    const double* gB = Dt + (size_t)bj * nkb * TR_CHUNK + (size_t)(wc * 4) * 64;
    const uint32_t voff = (uint32_t)lane * 8u;
    d4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    double a[TR_BIG_PD][8], b[TR_BIG_PD][4];
    auto issue = [&](double (&av)[8], double (&bv)[4], int s) {
      const double* pa = gA + (size_t)s * 512;
      const double* pb = gB + (size_t)s * 512;
      TR_LOAD8(av[0], voff, pa, 0); TR_LOAD8(av[1], voff, pa, 512); TR_LOAD8(av[2], voff, pa, 1024); TR_LOAD8(av[3], voff, pa, 1536);
      TR_LOAD8(av[4], voff, pa, 2048); TR_LOAD8(av[5], voff, pa, 2560); TR_LOAD8(av[6], voff, pa, 3072); TR_LOAD8(av[7], voff, pa, 3584);
      TR_LOAD8(bv[0], voff, pb, 0); TR_LOAD8(bv[1], voff, pb, 512); TR_LOAD8(bv[2], voff, pb, 1024); TR_LOAD8(bv[3], voff, pb, 1536);
    };
    // One k-step: 32 MFMAs and the 12 loads that refill the SAME ring slot with step `sn`.  Every statement is asm volatile, so the emitted order is the
    // written one (volatile asms keep their relative order; the compiler may still move plain scalar address arithmetic between them): each operand register
    // is reloaded right after its last reader has issued -- an A fragment after its row of four MFMAs; for the B fragments the last two rows are walked
    // column-wise so that b[3] .. b[0] are released one by one.  Each accumulator still sees its k-steps in ascending order: same bits as before.
#define TR_MFMA(i, j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(av[i]), "v"(bv[j]))
    auto step = [&](double (&av)[8], double (&bv)[4], int sn) {
      const double* pa = gA + (size_t)sn * 512;
      const double* pb = gB + (size_t)sn * 512;
      TR_MFMA(0, 0); TR_MFMA(0, 1); TR_MFMA(0, 2); TR_MFMA(0, 3); TR_LOAD8(av[0], voff, pa, 0);
      TR_MFMA(1, 0); TR_MFMA(1, 1); TR_MFMA(1, 2); TR_MFMA(1, 3); TR_LOAD8(av[1], voff, pa, 512);
      TR_MFMA(2, 0); TR_MFMA(2, 1); TR_MFMA(2, 2); TR_MFMA(2, 3); TR_LOAD8(av[2], voff, pa, 1024);
      TR_MFMA(3, 0); TR_MFMA(3, 1); TR_MFMA(3, 2); TR_MFMA(3, 3); TR_LOAD8(av[3], voff, pa, 1536);
      TR_MFMA(4, 0); TR_MFMA(4, 1); TR_MFMA(4, 2); TR_MFMA(4, 3); TR_LOAD8(av[4], voff, pa, 2048);
      TR_MFMA(5, 0); TR_MFMA(5, 1); TR_MFMA(5, 2); TR_MFMA(5, 3); TR_LOAD8(av[5], voff, pa, 2560);
      TR_MFMA(6, 3); TR_MFMA(7, 3); TR_LOAD8(bv[3], voff, pb, 1536);
      TR_MFMA(6, 2); TR_MFMA(7, 2); TR_LOAD8(bv[2], voff, pb, 1024);
      TR_MFMA(6, 1); TR_MFMA(7, 1); TR_LOAD8(bv[1], voff, pb, 512);
      TR_MFMA(6, 0); TR_MFMA(7, 0); TR_LOAD8(bv[0], voff, pb, 0);
      TR_LOAD8(av[6], voff, pa, 3072); TR_LOAD8(av[7], voff, pa, 3584);
    };
#pragma unroll
    for (int u = 0; u < TR_BIG_PD; ++u) issue(a[u], b[u], u);
    for (int s0 = 0; s0 < nsteps; s0 += TR_BIG_PD * TR_BIG_UNROLL) {      // nsteps is a multiple of 32
#pragma unroll
      for (int rnd = 0; rnd < TR_BIG_UNROLL; ++rnd)
#pragma unroll
      for (int u = 0; u < TR_BIG_PD; ++u) {
        TR_WAIT12(36, a[u], b[u]);                                   // 12 (TR_BIG_PD - 1): all but the three newest k-steps have landed
        static_assert(TR_BIG_PD == 4, "the wait count above is 12 (TR_BIG_PD - 1)");
        int sn = s0 + rnd * TR_BIG_PD + u + TR_BIG_PD;
        sn = sn < nsteps ? sn : nsteps - 1;                          // past the end: re-read the last step (never consumed)
        step(a[u], b[u], sn);
      }
    }
#undef TR_MFMA
    // the unconsumed re-reads keep their registers until they have landed (see k_trmm_f64_reg)
#pragma unroll
    for (int u = 0; u < TR_BIG_PD; ++u) TR_WAIT12(0, a[u], b[u]);
    // The hazard recogniser does not see MFMAs inside asm statements: nothing may read an accumulator until the last MFMA has retired (16 passes = 64
    // cycles).  The MFMAs and this statement are all asm volatile, so none of them can sink below it; the stores cannot move above it (memory clobber), and
    // the accumulator reads (v_accvgpr_read) feed only those stores.  tests/test_kernel_isa.py checks the emitted order on every build.
    // (Naming the accumulators as in/out operands here made hipcc keep half of them in architectural VGPRs and copy them around every MFMA.)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const size_t row0 = (size_t)bi * TR_BM, col0 = (size_t)bj * TR_BN + wc * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          const size_t row = row0 + i * 16 + (lane >> 4) + 4 * r;
          if (row < row_hi) X[row * ldx + col0 + j * 16 + (lane & 15)] = acc[i][j][r];
        }
  }
  TR_CLK_END
}

// ---- p_i <- D_{Z, r, x_i} ------------------------------------------------------------------------------
// Wave-compacted rounding: a wave owns SEG consecutive samples of the flattened valid index space
// g = coord * B + b.  Every iteration all 64 lanes evaluate one attempt of THEIR current sample; the lanes that
// accepted store the result and take the next unassigned sample ids (ballot + mbcnt prefix), whose centres are
// fetched from a register window of the next 192 centres by cross-lane reads.  Which lane evaluates which
// (sample, attempt) changes nothing: the value is the first accepted attempt of the sample's own Philox stream.
constexpr int PR_SEG = 4096;
constexpr long long kDigitRangeP = (1ll << 23) - 1;

// acc += shfl_xor(acc, off) for off = 32, 16, 8, 4, 2, 1 -- the xor butterfly of the dot256 contract -- without LDS-crossbar
// permutes: v_permlane32_swap / v_permlane16_swap (gfx950) for the two widest levels, DPP moves for the rest.  Bit-identical to
// the __shfl_xor form (a + b = b + a); checked on hardware by tools/probe_dpp_butterfly.hip.
__device__ inline double wave_xor_sum(double x) {
  typedef unsigned int u2v __attribute__((ext_vector_type(2)));
  auto mk = [](uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); };
  uint32_t lo = (uint32_t)__double2loint(x), hi = (uint32_t)__double2hiint(x);
  { const u2v a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = mk(a.x, b.x) + mk(a.y, b.y); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { const u2v a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = mk(a.x, b.x) + mk(a.y, b.y); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { const int pl = __builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xf, 0xf, false), ph = __builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, false);   // row_ror:8
    x = x + mk((uint32_t)pl, (uint32_t)ph); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { int pl = __builtin_amdgcn_update_dpp(0, (int)lo, 0x104, 0xf, 0x5, false); pl = __builtin_amdgcn_update_dpp(pl, (int)lo, 0x114, 0xf, 0xa, false);       // row_shl:4 | row_shr:4
    int ph = __builtin_amdgcn_update_dpp(0, (int)hi, 0x104, 0xf, 0x5, false); ph = __builtin_amdgcn_update_dpp(ph, (int)hi, 0x114, 0xf, 0xa, false);
    x = x + mk((uint32_t)pl, (uint32_t)ph); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { const int pl = __builtin_amdgcn_update_dpp(0, (int)lo, 0x4e, 0xf, 0xf, false), ph = __builtin_amdgcn_update_dpp(0, (int)hi, 0x4e, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
    x = x + mk((uint32_t)pl, (uint32_t)ph); lo = (uint32_t)__double2loint(x); hi = (uint32_t)__double2hiint(x); }
  { const int pl = __builtin_amdgcn_update_dpp(0, (int)lo, 0xb1, 0xf, 0xf, false), ph = __builtin_amdgcn_update_dpp(0, (int)hi, 0xb1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
    x = x + mk((uint32_t)pl, (uint32_t)ph); }
  return x;
}

__device__ inline int lane_rank(uint64_t mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__global__ __launch_bounds__(256) void k_perturb_round_wave(uint64_t seed, uint64_t first_index, size_t m, size_t B, size_t ld,
                                                            const double* __restrict__ X, SampleZParams sp, int32_t* __restrict__ P,
                                                            int* __restrict__ fail) {
  const int lane = threadIdx.x & 63;
  const size_t wave_id = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t total = m * B;
  const size_t seg0 = wave_id * PR_SEG;
  if (seg0 >= total) return;
  const size_t seg1 = seg0 + PR_SEG < total ? seg0 + PR_SEG : total;
  // sample id g = seg0 + off  ->  (coordinate, preimage) by 32-bit arithmetic relative to the segment start
  const size_t coord0 = seg0 / B;
  const uint32_t b00 = (uint32_t)(seg0 % B), B32 = (uint32_t)B;
  const bool one_wrap = B32 >= (uint32_t)PR_SEG;          // then o < b00 + PR_SEG < 2 B: the division is a compare (uniform choice)
  auto locate = [&](size_t g, size_t* coord, uint32_t* bb) {
    const uint32_t o = b00 + (uint32_t)(g - seg0);
    if (one_wrap) {
      const bool wrap = o >= B32;
      *coord = coord0 + (wrap ? 1 : 0);
      *bb = wrap ? o - B32 : o;
    } else {
      *coord = coord0 + o / B32;
      *bb = o % B32;
    }
  };
  auto centre_of = [&](size_t g) -> double {
    if (g >= seg1) return 0.0;
    size_t cc; uint32_t bb;
    locate(g, &cc, &bb);
    return X[cc * ld + bb];
  };
  // window of prefetched centres: w0 = [base, base+64), w1 = [base+64, base+128), w2 = [base+128, base+192)
  size_t base = seg0 + 64;
  double w0 = centre_of(base + lane), w1 = centre_of(base + 64 + lane), w2 = centre_of(base + 128 + lane);
  size_t my = seg0 + lane, next_free = seg0 + 64;
  bool active = my < seg1;
  double c = centre_of(my);
  size_t coord = 0; uint32_t b = 0;
  locate(active ? my : seg0, &coord, &b);
  uint32_t t = 0;
  int f = 0;
  while (__ballot(active)) {
    bool accept = false;
    long long x = 0;
    if (active) {
      // four attempts of this lane's sample (one Philox block when the range is narrow), screened in fp32: see sz_group4
      const uint64_t index = first_index + b;
      const uint32_t tw = tag_word(TAG_PERTURB, index);
      const SzRange rg = sz_range(c, sp);
      accept = sz_group4(seed, (uint32_t)coord, (uint32_t)index, tw, t, rg, c, sp.inv_s, &x);
      if (!accept && ++t >= kMaxAttempts / 4) { accept = true; f = 1; x = (long long)floor(c + 0.5); }
      if (accept) {
        if (x > kDigitRangeP || x < -kDigitRangeP) f = 1;    // the syndrome product needs |p| < 2^23 (k_split_P)
        P[coord * ld + b] = (int32_t)x;
      }
    }
    const uint64_t mask = __ballot(accept);
    if (mask) {
      const size_t nid = next_free + (size_t)lane_rank(mask);
      const int off = (int)(nid - base);                     // in [0, 128) for accepting lanes
      const double v0 = __shfl(w0, off & 63), v1 = __shfl(w1, off & 63);
      if (accept) {
        my = nid;
        active = nid < seg1;
        c = off < 64 ? v0 : v1;
        t = 0;
        if (active) locate(nid, &coord, &b);
      }
      next_free += (size_t)__popcll(mask);
      if (next_free >= base + 64) {                           // slide the window
        base += 64;
        w0 = w1; w1 = w2;
        w2 = centre_of(base + 128 + lane);
      }
    }
  }
  if (f) atomicOr(fail, 1);
}

// The same rounding with a lean refill (round 3; narrow SampleZ words, i.e. ceil(6 r) + floor(6 r) + 1 <= 4096 -- every PSFPerturbation of BASELINE).
// What the wave-compacted kernel above pays per iteration beside its Philox block and four screens -- sz_range in f64 for every lane, two 64-bit
// cross-lane shuffles of the window registers, 64-bit sample ids -- is moved to where it is needed: the constants of a sample (range start, fp32
// offset, Lemire threshold) are computed once when a lane TAKES the sample, the window of upcoming centres lives in a wave-private LDS ring of 256
// doubles (one ds_read per refill; the ring is topped up 64 centres at a time from a register loaded one slide earlier), ids are 32-bit offsets into
// the wave's segment.  Values are those of k_perturb_round_wave bit for bit: a sample's value is the first accepted attempt of its own stream.
constexpr int PRL_SEG = 8192;
constexpr int PRL_WIN = 256;
// `seg` = samples per wave (a multiple of 64, at most PRL_SEG): PRL_SEG for full batches; a single call (one preimage, psf.rs:48-80) has only m
// samples in all, and the launch lasts as long as its longest wave, so the host cuts them into short segments (prl_segment).
__host__ inline uint32_t prl_segment(size_t total) {
  // ~2048 waves (two per SIMD) for a handful of preimages, up to ~8192 as the batch grows: a wave's last samples run with most lanes idle (~12 iterations), so few
  // long waves waste least, but below ~500 samples per wave-slot more waves hide each other's latencies better (round 6, tools/segment_sweep.py at C3: 64 preimages
  // 0.080 -> 0.068 ms, 256 preimages 0.237 -> 0.205)
  size_t waves = total / 512;
  if (waves < 2048) waves = 2048;
  if (waves > 8192) waves = 8192;
  size_t seg = (total / waves + 63) / 64 * 64;
  if (seg < 64) seg = 64;
  return (uint32_t)(seg > (size_t)PRL_SEG ? (size_t)PRL_SEG : seg);
}
__global__ __launch_bounds__(256) void k_perturb_round_lean(uint64_t seed, uint64_t first_index, size_t m, size_t B, size_t ld,
                                                            const double* __restrict__ X, SampleZParams sp, int32_t* __restrict__ P,
                                                            int* __restrict__ fail, uint32_t seg) {
  __shared__ double s_win[4][PRL_WIN];
  const int lane = threadIdx.x & 63;
  double* win = s_win[threadIdx.x >> 6];
  const size_t total = m * B;
  const size_t seg0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * seg;
  if (seg0 >= total) return;                                   // (no workgroup barrier below: a wave may leave alone)
  const uint32_t nseg = (uint32_t)(total - seg0 < (size_t)seg ? total - seg0 : (size_t)seg);
  const uint32_t coord0 = (uint32_t)(seg0 / B), b00 = (uint32_t)(seg0 % B), B32 = (uint32_t)B;
  const bool few_wraps = B32 >= seg;                           // then an offset wraps at most once
  const uint32_t Bmagic = udiv_magic_of(B32);                  // (o < seg + B <= 2^14, B < seg <= 2^13 where it is used)
  auto locate = [&](uint32_t off, uint32_t* coord, uint32_t* bb) {
    const uint32_t o = b00 + off;
    if (few_wraps) { const bool wrap = o >= B32; *coord = coord0 + (wrap ? 1u : 0u); *bb = wrap ? o - B32 : o; }
    else { uint32_t qd; udivmod_magic(o, B32, Bmagic, &qd, bb); *coord = coord0 + qd; }
  };
  auto gload = [&](uint32_t off) -> double {
    if (off >= nseg) return 0.0;
    uint32_t cc, bb;
    locate(off, &cc, &bb);
    return X[(size_t)cc * ld + bb];
  };
  // ring: offsets [loaded - 256, loaded) sit at index (offset & 255); pf holds the block [loaded, loaded + 64)
  win[lane] = gload(lane); win[64 + lane] = gload(64 + lane); win[128 + lane] = gload(128 + lane);
  uint32_t loaded = 192;
  double pf = gload(192 + lane);
  uint32_t my = (uint32_t)lane, next_free = 64;
  bool active = my < nseg;
  const float inv_s_f = (float)sp.inv_s;
  double c = 0.0; float c_rel = 0.f; uint32_t coord = 0, b = 0, idx_lo = 0, tw = 0, t = 0;
  SzRange rg{0, 1, 0, 16};
  bool generic = false;
  auto take = [&](uint32_t off) {                               // a lane adopts sample `off`: everything that depends on the sample only
    c = win[off & (PRL_WIN - 1)];
    locate(off, &coord, &b);
    const uint64_t index = first_index + b;
    idx_lo = (uint32_t)index;
    tw = tag_word(TAG_PERTURB, index);
    rg = sz_range(c, sp);
    c_rel = (float)((double)rg.lo - c);
    generic = !(fabs(c) < 0x1.0p40);
    t = 0;
  };
  if (active) take(my);
  int f = 0;
  while (__ballot(active)) {
    bool accept = false;
    long long x = 0;
    if (active) {
      accept = generic ? sz_group4(seed, coord, idx_lo, tw, t, rg, c, sp.inv_s, &x)
                       : sz_group4_narrow(seed, coord, idx_lo, tw, t, rg, c, sp.inv_s, c_rel, inv_s_f, &x);
      if (!accept && ++t >= kMaxAttempts / 4) { accept = true; f = 1; x = (long long)floor(c + 0.5); }
      if (accept) {
        if (x > kDigitRangeP || x < -kDigitRangeP) f = 1;       // the syndrome product needs |p| < 2^23 (k_split_P)
        P[(size_t)coord * ld + b] = (int32_t)x;
      }
    }
    const uint64_t mask = __ballot(accept);
    if (mask) {
      const uint32_t nid = next_free + (uint32_t)lane_rank(mask);
      next_free += (uint32_t)__popcll(mask);
      while (loaded < next_free + 64 && loaded < nseg) {        // uniform: keep every offset below next_free + 64 in the ring
        win[(loaded + lane) & (PRL_WIN - 1)] = pf;
        loaded += 64;
        pf = gload(loaded + lane);
      }
      if (accept) {
        my = nid;
        active = nid < nseg;
        if (active) take(nid);
      }
    }
  }
  if (f) atomicOr(fail, 1);
}

// k_perturb_round_lean with the table screen of psf_rng.hpp (sz_screen16_tab) in place of the fp32 screen: same attempts, same exact decisions, same values.
// ROW: a segment is exactly one row of the [coordinate][preimage] matrices (seg == B, the host's choice from 1024 preimages on): the coordinate is the wave's, the preimage
// the offset, and the loads and stores go through the row's base address
template <bool ROW>
__global__ __launch_bounds__(256) void k_perturb_round_tab(uint64_t seed, uint64_t first_index, size_t m, size_t B, size_t ld,
                                                            const double* __restrict__ X, SampleZParams sp, int32_t* __restrict__ P,
                                                            int* __restrict__ fail, uint32_t seg, SzTable tb) {
  __shared__ double s_win[4][PRL_WIN];
  extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];      // the screen table of this s: rows x F words
  const int lane = threadIdx.x & 63;
  double* win = s_win[threadIdx.x >> 6];
  const size_t total = m * B;
  const size_t seg0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * seg;
  for (uint32_t e = threadIdx.x; e < tb.rows * tb.F; e += 256) s_tab[e] = tb.t[e];
  __syncthreads();
  if (seg0 >= total) return;                                   // (no workgroup barrier below: a wave may leave alone)
  const uint32_t nseg = (uint32_t)(total - seg0 < (size_t)seg ? total - seg0 : (size_t)seg);
  const uint32_t coord0 = (uint32_t)(seg0 / B), b00 = (uint32_t)(seg0 % B), B32 = (uint32_t)B;
  const bool few_wraps = B32 >= seg;                           // then an offset wraps at most once
  const uint32_t Bmagic = udiv_magic_of(B32);                  // (o < seg + B <= 2^14, B < seg <= 2^13 where it is used)
  const double* __restrict__ Xrow = X + (size_t)coord0 * ld;
  int32_t* __restrict__ Prow = P + (size_t)coord0 * ld;
  auto locate = [&](uint32_t off, uint32_t* coord, uint32_t* bb) {
    if constexpr (ROW) { *coord = coord0; *bb = off; return; }
    const uint32_t o = b00 + off;
    if (few_wraps) { const bool wrap = o >= B32; *coord = coord0 + (wrap ? 1u : 0u); *bb = wrap ? o - B32 : o; }
    else { uint32_t qd; udivmod_magic(o, B32, Bmagic, &qd, bb); *coord = coord0 + qd; }
  };
  auto gload = [&](uint32_t off) -> double {
    if (off >= nseg) return 0.0;
    if constexpr (ROW) return Xrow[off];
    uint32_t cc, bb;
    locate(off, &cc, &bb);
    return X[(size_t)cc * ld + bb];
  };
  // ring: offsets [loaded - 256, loaded) sit at index (offset & 255); pf holds the block [loaded, loaded + 64)
  win[lane] = gload(lane); win[64 + lane] = gload(64 + lane); win[128 + lane] = gload(128 + lane);
  uint32_t loaded = 192;
  double pf = gload(192 + lane);
  uint32_t my = (uint32_t)lane, next_free = 64;
  bool active = my < nseg;
  const double Fd = (double)tb.F;
  const int c6_32 = (int)sp.c6;
  double c = 0.0; uint32_t coord = 0, b = 0, idx_lo = 0, tw = 0, t = 0, bin = 0;
  SzRange rg{0, 1, 0, 16};
  bool generic = false;
  auto take = [&](uint32_t off) {                               // a lane adopts sample `off`: everything that depends on the sample only
    c = win[off & (PRL_WIN - 1)];
    locate(off, &coord, &b);
    const uint64_t index = first_index + b;
    idx_lo = (uint32_t)index;
    tw = tag_word(TAG_PERTURB, index);
    generic = !(fabs(c) < 0x1.0p30);
    if (generic) rg = sz_range(c, sp);
    else {                                                      // sz_range with the range start formed in 32 bits (|c| < 2^30, ceil(6 s) <= 2048)
      const double cc = ceil(c);
      const bool integral = cc == c;
      rg.lo = (long long)((int)cc - c6_32);
      rg.N = integral ? sp.n_int : sp.n_int - 1;
      rg.thr = integral ? sp.thr_int : sp.thr_frac;
      rg.sh = 16;
      const uint32_t bq = (uint32_t)((cc - c) * Fd);           // delta in [0, 1): its bin (the table's bins overlap by 1e-9, psfp.hip)
      bin = bq < tb.F ? bq : tb.F - 1;
    }
    t = 0;
  };
  if (active) take(my);
  int f = 0;
  while (__ballot(active)) {
    bool accept = false;
    long long x = 0;
    if (active) {
      accept = generic ? sz_group4(seed, coord, idx_lo, tw, t, rg, c, sp.inv_s, &x)
                       : sz_group4_tab(seed, coord, idx_lo, tw, t, rg, c, sp.inv_s, s_tab + bin, tb.F, &x);
      if (!accept && ++t >= kMaxAttempts / 4) { accept = true; f = 1; x = (long long)floor(c + 0.5); }
      if (accept) {
        if (x > kDigitRangeP || x < -kDigitRangeP) f = 1;       // the syndrome product needs |p| < 2^23 (k_split_P)
        if constexpr (ROW) Prow[b] = (int32_t)x;
        else P[(size_t)coord * ld + b] = (int32_t)x;
      }
    }
    const uint64_t mask = __ballot(accept);
    if (mask) {
      const uint32_t nid = next_free + (uint32_t)lane_rank(mask);
      next_free += (uint32_t)__popcll(mask);
      while (loaded < next_free + 64 && loaded < nseg) {        // uniform: keep every offset below next_free + 64 in the ring
        win[(loaded + lane) & (PRL_WIN - 1)] = pf;
        loaded += 64;
        pf = gload(loaded + lane);
      }
      if (accept) {
        my = nid;
        active = nid < nseg;
        if (active) take(nid);
      }
    }
  }
  if (f) atomicOr(fail, 1);
}

// ---- integer products over Z_q ---------------------------------------------------------------------------
// S[i][c] = sum_t a[i][t] * p[t][c]  (a in [0,q) as u64, p small signed), reduced mod q, then an epilogue:
//   ZQ_SYNDROME : out[i][c] = (u[c][i] - S) mod q      out n x ld   (mp_perturbation.rs:318)
//   ZQ_FA       : out[c][i] = S mod q                   out B x n    (mp_perturbation.rs:368)
//   ZQ_TRAPDOOR : A[i][off + c] = ((H G)[i][c] - S) mod q   (gadget_classical.rs:66; tag H = identity when `tagm` is null)
// a is split into 31-bit limbs; each 32-term tile is summed in int64 (|a_limb * p| < 2^31 * 2^25) and folded
// into a 128-bit running total, so no intermediate ever wraps for any q < 2^62.
enum ZqMode { ZQ_SYNDROME = 0, ZQ_FA = 1, ZQ_TRAPDOOR = 2 };

struct Acc128 { uint64_t lo; int64_t hi; };
__device__ inline void acc128_add(Acc128& t, int64_t v) {
  const uint64_t nl = t.lo + (uint64_t)v;
  t.hi += (v >> 63) + (nl < t.lo ? 1 : 0);
  t.lo = nl;
}
// (hi * 2^64 + lo) mod q for |hi| small; two64 = 2^64 mod q
__device__ inline uint64_t acc128_mod(Acc128 t, uint64_t q, uint64_t two64) {
  uint64_t r = t.lo % q;
  int64_t h = t.hi;
  const bool neg = h < 0;
  uint64_t hm = (uint64_t)(neg ? -h : h);
  // hm * two64 mod q by double-and-add (hm < 2^16 in every use)
  uint64_t term = 0, base = two64;
  while (hm) {
    if (hm & 1) { term += base; if (term >= q) term -= q; }
    base += base; if (base >= q) base -= q;
    hm >>= 1;
  }
  if (neg) { r = r >= term ? r - term : r + q - term; }
  else { r += term; if (r >= q) r -= q; }
  return r;
}

template <typename PT, bool WIDE>
__global__ __launch_bounds__(256) void k_zq_matmul(int mode, const uint64_t* __restrict__ Amat, size_t lda, size_t a_off,
                                                   size_t nrows, size_t K, const PT* __restrict__ Pm, size_t ldp, size_t ncols,
                                                   uint64_t q, uint64_t two64, uint64_t two31,
                                                   const uint64_t* __restrict__ U, uint64_t* __restrict__ out, size_t ldo,
                                                   size_t out_off, const uint64_t* __restrict__ gvec, uint64_t gk,
                                                   const uint64_t* __restrict__ tagm) {
  // tile 64 (rows i) x 64 (cols c), K tile 32; thread -> 4 x 4 outputs
  __shared__ uint32_t sAlo[64][33];
  __shared__ uint32_t sAhi[WIDE ? 64 : 1][33];
  __shared__ int32_t sP[32][64];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const size_t i0 = (size_t)blockIdx.y * 64, c0 = (size_t)blockIdx.x * 64;
  Acc128 tlo[4][4], thi[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) { tlo[r][c] = Acc128{0, 0}; thi[r][c] = Acc128{0, 0}; }

  for (size_t k0 = 0; k0 < K; k0 += 32) {
    // stage A tile (64 x 32) and P tile (32 x 64)
    for (int e = tid; e < 64 * 32; e += 256) {
      const int r = e >> 5, kk = e & 31;
      uint64_t a = 0;
      if (i0 + r < nrows && k0 + kk < K) a = Amat[(i0 + r) * lda + a_off + k0 + kk];
      sAlo[r][kk] = (uint32_t)(a & 0x7fffffffu);
      if (WIDE) sAhi[r][kk] = (uint32_t)(a >> 31);
    }
    for (int e = tid; e < 32 * 64; e += 256) {
      const int kk = e >> 6, c = e & 63;
      int32_t p = 0;
      if (k0 + kk < K && c0 + c < ncols) p = (int32_t)Pm[(k0 + kk) * ldp + c0 + c];
      sP[kk][c] = p;
    }
    __syncthreads();
    int64_t alo[4][4], ahi[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) { alo[r][c] = 0; ahi[r][c] = 0; }
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      int32_t p[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) p[c] = sP[kk][tx * 4 + c];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int32_t al = (int32_t)sAlo[ty * 4 + r][kk];
#pragma unroll
        for (int c = 0; c < 4; ++c) alo[r][c] += (int64_t)al * p[c];
        if (WIDE) {
          const int32_t ah = (int32_t)sAhi[ty * 4 + r][kk];
#pragma unroll
          for (int c = 0; c < 4; ++c) ahi[r][c] += (int64_t)ah * p[c];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc128_add(tlo[r][c], alo[r][c]);
        if (WIDE) acc128_add(thi[r][c], ahi[r][c]);
      }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t i = i0 + ty * 4 + r, cc = c0 + tx * 4 + c;
      if (i >= nrows || cc >= ncols) continue;
      uint64_t s = acc128_mod(tlo[r][c], q, two64);
      if (WIDE) {
        const uint64_t h = acc128_mod(thi[r][c], q, two64);
        // s += h * 2^31 mod q  (h, two31 < q < 2^62): double-and-add on two31's bits is overkill; use 128-bit split
        const uint64_t plo = h * two31, phi = __umul64hi(h, two31);
        Acc128 t{plo, (int64_t)phi};
        // phi < 2^60: fold with acc128_mod's double-and-add on hi (hm up to 2^60 -> 60 iterations, once per output)
        const uint64_t hs = acc128_mod(t, q, two64);
        s += hs; if (s >= q) s -= q;
      }
      if (mode == ZQ_SYNDROME) {
        const uint64_t u = U[cc * nrows + i] % q;
        out[i * ldo + cc] = u >= s ? u - s : u + q - s;
      } else if (mode == ZQ_FA) {
        out[cc * ldo + i] = s;
      } else {
        // G[j][cc] = base^(cc % gk) if cc / gk == j (gadget_classical.rs:91-107), so (H G)[i][cc] = H[i][cc / gk] base^(cc % gk)
        uint64_t g;                                             // gvec[t] = base^t mod q
        if (!tagm) g = (cc / gk == i) ? gvec[cc % gk] : 0;
        else {
          const uint64_t hv = tagm[i * nrows + cc / gk] % q, gv = gvec[cc % gk];
          g = acc128_mod(Acc128{hv * gv, (int64_t)__umul64hi(hv, gv)}, q, two64);
        }
        out[i * ldo + out_off + cc] = g >= s ? g - s : g + q - s;
      }
    }
}

// ---- v = u - A p and u = A e on the int8 matrix cores ----------------------------------------------------
// a in [0,q) is written in balanced base-256 digits d_0..d_{NA-1} (d_i in [-128,127]), p (|p| < 2^23) in three.
// sum_k a_k p_k = sum_c 256^c T_c,  T_c = sum_{i+j=c} sum_k d_i[k] e_j[k]: each (i,j) pair is one int8 MFMA product
// accumulated exactly in int32 (<= 16384 terms of |d e| <= 2^14 per pair, <= 3 pairs per class before the fold), and
// every 16384 coordinates the classes are folded into a running residue mod q.  Integer arithmetic: exact.
// Operand tiles of the int8 MFMA kernels that are staged as [row][64 bytes] (one row = the 64 coordinates of a K step = four 16-byte k groups): the lanes that
// read k group g of 16 consecutive rows would all start in the same 16 of the 64 LDS banks (row stride 64 B: a 4-way conflict on every ds_read_b128).  The
// groups of a row are therefore stored rotated by row / 4: group g of row r sits in slot (g + r / 4) mod 4, and the 16 lanes cover all 64 banks once.
__host__ __device__ inline int i8_slot(int row, int group) { return (group + (row >> 2)) & 3; }

__global__ void k_split_A(const uint64_t* __restrict__ A, size_t lda, size_t n, size_t K, size_t n_pad, size_t K_pad, int NA,
                          int8_t* __restrict__ A8) {
  const size_t total = n_pad * K_pad;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / K_pad, kk = g % K_pad;
    int64_t a = (i < n && kk < K) ? (int64_t)A[i * lda + kk] : 0;
    for (int d = 0; d < NA; ++d) {
      int64_t dig = (d + 1 < NA) ? (int64_t)(int8_t)(a & 0xff) : a;
      // tile-packed: [digit][row tile of 64][k step of 64][row in tile][64 bytes, k groups rotated: i8_slot] -- one 4 KiB tile is one contiguous read
      const size_t off = (((size_t)d * (n_pad / 64) + i / 64) * (K_pad / 64) + kk / 64) * 4096 + (i % 64) * 64 + i8_slot((int)(i % 64), (int)((kk % 64) >> 4)) * 16 + (kk % 16);
      A8[off] = (int8_t)dig;
      a = (a - dig) >> 8;
    }
  }
}

// P (K x ld int32) -> three digit planes [K_pad/16][ld][16]; thread = (16-coordinate group, preimage).  Only the columns [col0, col0 + cw) are
// converted (the two halves of a batch run their stages on two streams: a half must not touch the other half's columns).
__global__ void k_split_P(const int32_t* __restrict__ P, size_t K, size_t ld, size_t ngroups, int8_t* __restrict__ P8, int* __restrict__ fail, size_t col0, size_t cw) {
  const size_t total = ngroups * cw;
  const size_t plane = ngroups * ld * 16;
  int f = 0, hi2 = 0;
  for (size_t g0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g0 < total; g0 += (size_t)gridDim.x * blockDim.x) {
    const size_t kg = g0 / cw, b = col0 + g0 % cw;
    const size_t g = kg * ld + b;
    v4i o0, o1, o2;
    int32_t w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const size_t kk = kg * 16 + j;
      const int32_t p = kk < K ? P[kk * ld + b] : 0;
      const int32_t e0 = (int32_t)(int8_t)(p & 0xff);
      const int32_t p1 = (p - e0) >> 8;
      const int32_t e1 = (int32_t)(int8_t)(p1 & 0xff);
      const int32_t e2 = (p1 - e1) >> 8;
      if (e2 > 127 || e2 < -128) f = 1;
      if (e2) hi2 = 1;
      w0[j >> 2] |= (e0 & 0xff) << (8 * (j & 3));
      w1[j >> 2] |= (e1 & 0xff) << (8 * (j & 3));
      w2[j >> 2] |= (e2 & 0xff) << (8 * (j & 3));
    }
    o0 = v4i{w0[0], w0[1], w0[2], w0[3]}; o1 = v4i{w1[0], w1[1], w1[2], w1[3]}; o2 = v4i{w2[0], w2[1], w2[2], w2[3]};
    *reinterpret_cast<v4i*>(P8 + g * 16) = o0;
    *reinterpret_cast<v4i*>(P8 + plane + g * 16) = o1;
    *reinterpret_cast<v4i*>(P8 + 2 * plane + g * 16) = o2;
  }
  if (f) atomicOr(fail, 1);
  // fail[2]: the third digit plane holds something (|p| >= 2^15 somewhere): k_zq_mfma multiplies it only then.  Sticky within a call (cleared with the failure words).
  if (__syncthreads_or(hi2) && threadIdx.x == 0 && __hip_atomic_load(fail + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(fail + 2, 1);
}

struct ZqConsts { uint64_t q, two64; uint64_t pw[12]; double inv_q; double pwd[12]; };   // pw[c] = 256^c mod q; inv_q = 1 / q and pwd[c] = (double)pw[c] for the narrow fold (q < 2^31)

// (T mod q) * pw mod q for |T| < 2^31
// q < 2^31 (round 5): both reductions by a double-precision quotient estimate and at most two corrections instead of two 64-bit divisions -- |T| < 2^31 and
// t pw < 2^62, so floor(T / q) and floor(t pw / q) < 2^31 are estimated within one unit by the 53-bit products below; the residues are exact.
__device__ inline uint64_t zq_term_narrow(int32_t T, uint32_t pw, double pwd, uint32_t q, double inv_q) {
  const int32_t q1 = (int32_t)floor((double)T * inv_q);
  int64_t t = (int64_t)T - (int64_t)q1 * (int64_t)q;                // in (-q, 2 q)
  if (t < 0) t += (int64_t)q;
  if (t >= (int64_t)q) t -= (int64_t)q;
  const uint32_t q2 = (uint32_t)((double)(uint32_t)t * pwd * inv_q);
  int64_t r = (int64_t)((uint64_t)(uint32_t)t * pw - (uint64_t)q2 * q);
  if (r < 0) r += (int64_t)q;
  if (r < 0) r += (int64_t)q;
  if (r >= (int64_t)q) r -= (int64_t)q;
  if (r >= (int64_t)q) r -= (int64_t)q;
  return (uint64_t)r;
}
__device__ inline uint64_t zq_term(int32_t T, uint64_t pw, uint64_t q, uint64_t two64, bool wide) {
  if (!wide) {
    int64_t t = (int64_t)T % (int64_t)q;
    if (t < 0) t += (int64_t)q;
    return ((uint64_t)t * pw) % q;
  }
  const uint64_t a = (uint64_t)(T < 0 ? -(int64_t)T : (int64_t)T);
  Acc128 t{a * pw, (int64_t)__umul64hi(a, pw)};
  const uint64_t r = acc128_mod(t, q, two64);
  return (T < 0 && r) ? q - r : r;
}

// grid = (column tiles, row tiles, K splits).  A split covers at most 256 K-steps (16384 coordinates), so its int32 class
// accumulators never overflow and are folded into a residue exactly once, after the loop; the per-split residues go to
// `part[split][i][c]` and k_zq_combine adds them.  (Folding inside the K loop made hipcc spill accumulators to scratch.)
// POW2 (round 5): q is a power of two that the NA digits of A cover (8 NA >= log2 q), so 256^c = 0 mod q for every class c >= NA and the digit pairs
// that only feed those classes are not multiplied at all (q = 2^30: 9 of the 12 pairs) -- the same residues.
template <int NA, bool FOLD128, bool POW2 = false>
__global__ __launch_bounds__(256) void k_zq_mfma(const int8_t* __restrict__ A8, size_t n_pad, size_t K_pad, const int8_t* __restrict__ P8, size_t ld,
                                                 int ks_per_split, ZqConsts zc, int wide, uint64_t* __restrict__ part, size_t col0, const int* __restrict__ flags) {
  constexpr int STAGE = (NA + 3) * 4096;
  constexpr int NC = POW2 ? NA : NA + 2;                 // live classes
  extern __shared__ __attribute__((aligned(16))) unsigned char zq_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t b0 = col0 + (size_t)blockIdx.x * 64, i0 = (size_t)blockIdx.y * 64;
  const size_t planeA = n_pad * K_pad, planeP = (K_pad / 16) * ld * 16;
  const int nks_all = (int)(K_pad / 64);
  const int ks0 = (int)blockIdx.z * ks_per_split;
  const int ks1 = ks0 + ks_per_split < nks_all ? ks0 + ks_per_split : nks_all;

  v4i acc[NC][2][2];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[c][x][y] = v4i{0, 0, 0, 0};
  const int pA = wave * 64 + lane;   // 16-byte piece of a 4 KiB tile
  const int8_t* srcA = A8 + (size_t)blockIdx.y * (K_pad / 64) * 4096 + (size_t)pA * 16;   // tile-packed planes, see k_split_A
  const int8_t* srcP = P8 + ((size_t)(pA >> 6) * ld + b0 + (size_t)(pA & 63)) * 16;
  // the K loop, for NP = 3 digit planes of p or -- when k_split_P found the third one empty (flags[2] == 0: |p| < 2^15 everywhere, the usual case) -- for two
  auto k_loop = [&](auto np_tag) {
    constexpr int NP = decltype(np_tag)::value;
    auto stage_load = [&](int ks, int buf) {
      unsigned char* base = zq_smem + buf * STAGE + wave * 1024;
#pragma unroll
      for (int d = 0; d < NA; ++d)
        __builtin_amdgcn_global_load_lds(srcA + (size_t)d * planeA + (size_t)ks * 4096, (lds_void_ptr)(base + d * 4096), 16, 0, 0);
#pragma unroll
      for (int e = 0; e < NP; ++e)
        __builtin_amdgcn_global_load_lds(srcP + (size_t)e * planeP + (size_t)ks * 4 * ld * 16, (lds_void_ptr)(base + (NA + e) * 4096), 16, 0, 0);
    };
    if (ks0 < ks1) {
      stage_load(ks0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int r16 = lane & 15, g = lane >> 4;
    for (int ks = ks0; ks < ks1; ++ks) {
      const int cur = (ks - ks0) & 1;
      if (ks + 1 < ks1) stage_load(ks + 1, cur ^ 1);
      const unsigned char* sb = zq_smem + cur * STAGE;
      v4i fa[NA][2], fp[NP][2];
#pragma unroll
      for (int d = 0; d < NA; ++d)
#pragma unroll
        for (int x = 0; x < 2; ++x) fa[d][x] = *reinterpret_cast<const v4i*>(sb + d * 4096 + ((wr * 32 + x * 16 + r16) * 64 + i8_slot(wr * 32 + x * 16 + r16, g) * 16));
#pragma unroll
      for (int e = 0; e < NP; ++e)
#pragma unroll
        for (int y = 0; y < 2; ++y) fp[e][y] = *reinterpret_cast<const v4i*>(sb + (NA + e) * 4096 + ((g * 64 + wc * 32 + y * 16 + r16) * 16));
#pragma unroll
      for (int d = 0; d < NA; ++d)
#pragma unroll
        for (int e = 0; e < NP; ++e)
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
              if (d + e < NC) acc[d + e < NC ? d + e : 0][x][y] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[d][x], fp[e][y], acc[d + e < NC ? d + e : 0][x][y], 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  if (flags[2] != 0) k_loop(std::integral_constant<int, 3>{});
  else k_loop(std::integral_constant<int, 2>{});
  const int r16 = lane & 15, g = lane >> 4;
  // fold the classes: residue = sum_c (T_c mod q) 256^c mod q.  C/D map: column (preimage) = lane & 15, row (i) = 4 * (lane >> 4) + reg
  uint64_t* dst = part + (size_t)blockIdx.z * n_pad * ld;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        uint64_t t = 0;
        if constexpr (POW2) {
          // q = 2^e divides 2^64: the sum of the classes in wrap-around 64-bit arithmetic, then a mask -- no division (the general fold below spends two 64-bit
          // divisions per class and output: half of this kernel's vector instructions at C3, SQ counters in profiles/r05_notes.md)
          uint64_t S = 0;
#pragma unroll
          for (int c = 0; c < NC; ++c) S += (uint64_t)(int64_t)acc[c][x][y][r] << (8 * c);
          t = S & (zc.q - 1);
        } else if constexpr (FOLD128) {
          // S = sum_c T_c 256^c as ONE signed 128-bit integer (|S| < 2^31 2^(8 (NC - 1))), then a single reduction mod q: a short split of a single call
          // spends more time in the per-class form below (two 64-bit divisions per class) than in its K loop
          Acc128 S{0, 0};
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int64_t T = (int64_t)acc[c][x][y][r];
            const int sh = 8 * c;
            uint64_t lo; int64_t hi;
            if (sh == 0) { lo = (uint64_t)T; hi = T >> 63; }
            else if (sh < 64) { lo = (uint64_t)T << sh; hi = T >> (64 - sh); }
            else { lo = 0; hi = (int64_t)((uint64_t)T << (sh - 64)); }
            const uint64_t nl = S.lo + lo;
            S.hi += hi + (nl < S.lo ? 1 : 0);
            S.lo = nl;
          }
          t = acc128_mod(S, zc.q, zc.two64);
        } else {
          for (int c = 0; c < NC; ++c) {
            int32_t T;
            switch (c) {   // static register indices
              case 0: T = acc[0][x][y][r]; break;
              case 1: T = acc[1][x][y][r]; break;
              case 2: T = acc[2 < NC ? 2 : 0][x][y][r]; break;
              case 3: T = acc[3 < NC ? 3 : 0][x][y][r]; break;
              case 4: T = acc[4 < NC ? 4 : 0][x][y][r]; break;
              case 5: T = acc[5 < NC ? 5 : 0][x][y][r]; break;
              case 6: T = acc[6 < NC ? 6 : 0][x][y][r]; break;
              case 7: T = acc[7 < NC ? 7 : 0][x][y][r]; break;
              case 8: T = acc[8 < NC ? 8 : 0][x][y][r]; break;
              default: T = acc[9 < NC ? 9 : 0][x][y][r]; break;
            }
            t += wide ? zq_term(T, zc.pw[c], zc.q, zc.two64, true) : zq_term_narrow(T, (uint32_t)zc.pw[c], zc.pwd[c], (uint32_t)zc.q, zc.inv_q);
            if (t >= zc.q) t -= zc.q;
          }
        }
        const size_t i = i0 + wr * 32 + x * 16 + 4 * g + r, cc = b0 + wc * 32 + y * 16 + r16;
        dst[i * ld + cc] = t;
      }
}

// out = (u - sum_z part[z]) mod q  (syndrome, mp_perturbation.rs:318)  or  sum_z part[z] mod q written row-per-preimage (f_a, :368)
__global__ void k_zq_combine(int mode, const uint64_t* __restrict__ part, int splits, size_t n, size_t n_pad, size_t ld, size_t ncols, uint64_t q,
                             const uint64_t* __restrict__ U, uint64_t* __restrict__ out, size_t ldo, size_t col0) {
  const size_t total = n * ncols;                  // columns [col0, col0 + ncols)
  for (size_t g0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g0 < total; g0 += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g0 / ncols, cc = col0 + g0 % ncols;
    const size_t g = i * ld + cc;
    uint64_t s = 0;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {                 // eight independent loads in flight (a single call has up to 64 splits and few outputs)
      uint64_t v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = part[(size_t)(z + t) * n_pad * ld + g];
#pragma unroll
      for (int t = 0; t < 8; ++t) { s += v[t]; if (s >= q) s -= q; }
    }
    for (; z < splits; ++z) {
      s += part[(size_t)z * n_pad * ld + g];
      if (s >= q) s -= q;
    }
    if (mode == ZQ_SYNDROME) {
      const uint64_t u = U[cc * n + i] % q;
      out[i * ldo + cc] = u >= s ? u - s : u + q - s;
    } else {
      out[cc * ldo + i] = s;
    }
  }
}

// the same for a single call (few outputs, up to 64 splits): one WAVE per output, lane z holds split z's residue, a butterfly of modular additions
// (exact in any order) instead of a thread walking 64 dependent loads -- 18 -> 4 us at one preimage
// SPLIT_MINOR: part[(row * ld + column) * splits + split] -- the residues of one output side by side (the fused tail of k_trmm_stream_fused leaves ~1000 of them per output)
template <bool SPLIT_MINOR = false>
__global__ __launch_bounds__(256) void k_zq_combine_wave(int mode, const uint64_t* __restrict__ part, int splits, size_t n, size_t n_pad, size_t ld, size_t ncols, uint64_t q,
                                                         const uint64_t* __restrict__ U, uint64_t* __restrict__ out, size_t ldo, size_t col0) {
  const size_t g0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g0 >= n * ncols) return;                                        // wave-uniform
  const size_t i = g0 / ncols, cc = col0 + g0 % ncols;
  const size_t g = i * ld + cc;
  uint64_t s = 0;
  for (int z = lane; z < splits; z += 64) { s += SPLIT_MINOR ? part[g * (size_t)splits + (size_t)z] : part[(size_t)z * n_pad * ld + g]; if (s >= q) s -= q; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s += (uint64_t)__shfl_xor((unsigned long long)s, off);
    if (s >= q) s -= q;
  }
  if (lane == 0) {
    if (mode == ZQ_SYNDROME) {
      const uint64_t u = U[cc * n + i] % q;
      out[i * ldo + cc] = u >= s ? u - s : u + q - s;
    } else {
      out[cc * ldo + i] = s;
    }
  }
}

// ---- structured sqrt(Sigma_2): x_top -= g (R d_2) on the int8 matrix cores ---------------------------------------------------------
// R (ternary) as the A operand, tile-packed like the digit planes of A (k_pack_R8); the five digit planes of q = d_2 2^32 as the B operand
// ([c/16][b][16], written by k_normals_wave).  Per plane an exact int32 sum over the w columns (|.| <= w 128 < 2^31), the planes are
// combined in int64 (|sum_c R q| < w 2^39 < 2^62), converted to double ONCE and folded into the centre:
//   X[i][b] = fma(-g, (double)(sum_c R[i][c] q[c][b]) 2^-32, X[i][b])      (X holds L_1 d_1 from k_trmm_f64 on entry)
// 64 x 64 tile per workgroup (4 waves, 2 x 2 of 32 x 32), K steps of 64, LDS-DMA double buffering as in k_zq_mfma.
__global__ void k_pack_R8(const int8_t* __restrict__ R, size_t ldr, size_t mbar, size_t w, size_t rows_pad, size_t K_pad, int8_t* __restrict__ R8) {
  const size_t total = rows_pad * K_pad;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / K_pad, kk = g % K_pad;
    const int8_t v = (i < mbar && kk < w) ? R[i * ldr + kk] : (int8_t)0;
    R8[((i / 64) * (K_pad / 64) + kk / 64) * 4096 + (i % 64) * 64 + i8_slot((int)(i % 64), (int)((kk % 64) >> 4)) * 16 + (kk % 16)] = v;
  }
}

__global__ __launch_bounds__(256, 2) void k_rd2_mfma(const int8_t* __restrict__ R8, size_t K_pad, const int8_t* __restrict__ D8, size_t plane_bytes, size_t ld,
                                                     size_t mbar, double g, double* __restrict__ X) {
  constexpr int NP = kFixPlanes, STAGE = (1 + NP) * 4096, NS = 3;      // three 24 KiB stages in flight: a K step is only 20 short MFMAs per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char rd_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t b0 = (size_t)blockIdx.x * 64, i0 = (size_t)blockIdx.y * 64;
  const int nks = (int)(K_pad / 64);
  v4i acc[NP][2][2];
#pragma unroll
  for (int c = 0; c < NP; ++c)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[c][x][y] = v4i{0, 0, 0, 0};
  const int pA = wave * 64 + lane;   // 16-byte piece of a 4 KiB tile
  const int8_t* srcA = R8 + (size_t)blockIdx.y * (K_pad / 64) * 4096 + (size_t)pA * 16;
  const int8_t* srcP = D8 + ((size_t)(pA >> 6) * ld + b0 + (size_t)(pA & 63)) * 16;
  auto stage_load = [&](int ks, int buf) {
    unsigned char* base = rd_smem + buf * STAGE + wave * 1024;
    __builtin_amdgcn_global_load_lds(srcA + (size_t)ks * 4096, (lds_void_ptr)base, 16, 0, 0);
#pragma unroll
    for (int e = 0; e < NP; ++e)
      __builtin_amdgcn_global_load_lds(srcP + (size_t)e * plane_bytes + (size_t)ks * 4 * ld * 16, (lds_void_ptr)(base + (1 + e) * 4096), 16, 0, 0);
  };
  for (int s0 = 0; s0 < NS - 1 && s0 < nks; ++s0) stage_load(s0, s0);
  const int r16 = lane & 15, gq = lane >> 4;
  int cur = 0;
  for (int ks = 0; ks < nks; ++ks) {
    // stage ks has landed when at most the NS - 2 younger stages (1 + NP DMA instructions each) are outstanding
    if (nks - 1 - ks >= NS - 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(1 + NP == 6, "the wait count above is the number of DMA instructions per stage");
    __syncthreads();                                  // everybody's part of stage ks is in LDS; everybody is done with stage ks - 1
    const int nxt = ks + NS - 1;
    if (nxt < nks) stage_load(nxt, cur == 0 ? NS - 1 : cur - 1);
    const unsigned char* sb = rd_smem + cur * STAGE;
    v4i fa[2], fp[NP][2];
#pragma unroll
    for (int x = 0; x < 2; ++x) fa[x] = *reinterpret_cast<const v4i*>(sb + ((wr * 32 + x * 16 + r16) * 64 + i8_slot(wr * 32 + x * 16 + r16, gq) * 16));
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int y = 0; y < 2; ++y) fp[e][y] = *reinterpret_cast<const v4i*>(sb + (1 + e) * 4096 + ((gq * 64 + wc * 32 + y * 16 + r16) * 16));
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[e][x][y] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[x], fp[e][y], acc[e][x][y], 0, 0, 0);
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  // C/D map: column (preimage) = lane & 15, row (i) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        long long tot = 0;
#pragma unroll
        for (int e = NP - 1; e >= 0; --e) tot = tot * 256 + (long long)acc[e][x][y][r];
        const size_t i = i0 + wr * 32 + x * 16 + 4 * gq + r, cc = b0 + wc * 32 + y * 16 + r16;
        if (i < mbar) {
          const double rd = (double)tot * 0x1.0p-32;
          X[i * ld + cc] = fma(-g, rd, X[i * ld + cc]);
        }
      }
}

// ---- gadget: digit decomposition + randomized nearest plane on S_k ------------------------------------
struct GadgetTables {   // device pointers, all of length k or k*k
  const int32_t* Sk;        // k x k basis block (row-major; columns are basis vectors)
  const double* gso;        // k x k Gram-Schmidt vectors (columns)
  const double* norm2;      // ||b~_i||^2
  const SampleZParams* sz;  // parameters of D_{Z, s_G/||b~_i||, .}
};

// find_solution_gadget_mat (gadget_classical.rs:219-229) as a standalone kernel: value[rows x cols] -> out[k rows x cols]
__global__ void k_digits(const uint64_t* __restrict__ value, size_t rows, size_t cols, uint64_t q, uint32_t k,
                         uint64_t base, int64_t* __restrict__ out) {
  const size_t total = rows * cols;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t j = g / cols, c = g % cols;
    uint64_t v = value[g] % q;
    for (uint32_t t = 0; t < k; ++t) {
      uint64_t d;
      if (base == 2) { d = v & 1; v >>= 1; }
      else { d = v % base; v = (v - d) / base; }
      out[((size_t)k * j + t) * cols + c] = (int64_t)d;
    }
  }
}

// One thread = one (row j of v, preimage b): x = digits(v_j); c = -x; for i = k-1..0: c' = <c,b~_i>/||b~_i||^2,
// z_i <- D_{Z,s_i,c'}, c -= z_i b_i; result -c  (mp_perturbation.rs:173-191 + GPV08 SampleD).
// c lives in LDS ([t][thread], conflict free), the k x k tables are broadcast reads from LDS.
// z is written as two int8 planes (z = lo + 256 hi) in the MFMA operand layout [c/16][b][16] (sixteen consecutive
// coordinates of one preimage = one 16-byte fragment element); fail[1] is set when any hi byte is non-zero.
__global__ __launch_bounds__(256) void k_gadget(uint64_t seed, uint64_t first_index, uint32_t n, uint32_t k, uint64_t q,
                                                uint64_t base, size_t B, size_t ld, const uint64_t* __restrict__ V,
                                                GadgetTables tb, int8_t* __restrict__ Zlo, int8_t* __restrict__ Zhi,
                                                int* __restrict__ fail) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* s_gso = reinterpret_cast<double*>(smem_raw);                   // k*k
  double* s_norm2 = s_gso + (size_t)k * k;                               // k
  SampleZParams* s_sz = reinterpret_cast<SampleZParams*>(s_norm2 + k);   // k
  int32_t* s_Sk = reinterpret_cast<int32_t*>(s_sz + k);                  // k*k
  int32_t* s_c = s_Sk + (size_t)k * k;                                   // k * 256
  const int tid = threadIdx.x;
  for (uint32_t e = tid; e < k * k; e += 256) { s_gso[e] = tb.gso[e]; s_Sk[e] = tb.Sk[e]; }
  for (uint32_t e = tid; e < k; e += 256) { s_norm2[e] = tb.norm2[e]; s_sz[e] = tb.sz[e]; }
  __syncthreads();
  const size_t b = (size_t)blockIdx.x * 256 + tid;
  const uint32_t j = blockIdx.y;
  if (b >= B) return;
  int f = 0, anyhi = 0;
  uint64_t v = V[(size_t)j * ld + b] % q;
  for (uint32_t t = 0; t < k; ++t) {
    uint64_t d;
    if (base == 2) { d = v & 1; v >>= 1; }
    else { d = v % base; v = (v - d) / base; }
    s_c[t * 256 + tid] = -(int32_t)d;
  }
  const uint64_t index = first_index + b;
  for (int i = (int)k - 1; i >= 0; --i) {
    double dot = 0.0;
    for (uint32_t t = 0; t < k; ++t) dot = fma((double)s_c[t * 256 + tid], s_gso[t * k + i], dot);
    const double c2 = dot / s_norm2[i];
    const long long zi = sample_z(seed, TAG_GADGET, index, j * k + (uint32_t)i, c2, s_sz[i], &f);
    const int32_t z32 = (int32_t)zi;
    if (zi > 0x3fffff || zi < -0x3fffff) f = 1;
    for (uint32_t t = 0; t < k; ++t) {
      const int32_t sk = s_Sk[t * k + i];
      if (sk) s_c[t * 256 + tid] -= z32 * sk;
    }
  }
  for (uint32_t t = 0; t < k; ++t) {
    const int32_t z = -s_c[t * 256 + tid];
    if (z > 32767 || z < -32768) f = 1;
    const int32_t lo = (int32_t)(int8_t)(z & 0xff);
    const int32_t hi = (z - lo) >> 8;
    const size_t c = (size_t)j * k + t;
    const size_t addr = ((c >> 4) * ld + b) * 16 + (c & 15);
    Zlo[addr] = (int8_t)lo;
    Zhi[addr] = (int8_t)hi;
    if (hi) anyhi = 1;
  }
  if (f) atomicOr(fail, 1);
  if (anyhi) atomicOr(fail + 1, 1);
}

// ---- the same sampler with task queues: no lane waits for another lane's rejection loop ---------------------------------
// A wave owns GQ_P problems (a problem = one (row j of v, preimage b) pair, k sequential steps).  Its state lives in LDS:
// c (int16, [row][problem]), the current step and centre, the drawn z.  Each problem is in exactly one place:
//   READY queue (centre known, waiting for a lane) -> held by a lane (rejection attempts, two per iteration)
//   -> PENDING queue (z drawn) -> advance pass (64 pending problems at once: c -= z b_i, next centre) -> READY ... -> done.
// Lanes that accept push their problem to PENDING and take the next READY one (ballot + mbcnt), so all 64 lanes attempt
// every iteration, and the sequential part (update + projection) always runs with a full wave.  Results are those of the
// lock-step kernel: each draw is the first accepted attempt of its own Philox stream, each projection the same fma chain.
struct GadgetTablesQ {
  const int32_t* Sk; const double* gso; const double* norm2; const SampleZParams* sz;
  const int32_t* rng;        // 4k: first / last non-zero row of b~_i, first / last non-zero row of b_i
};
constexpr int GQ_WAVES = 4;

// slots per wave (a power of two).  256 measured slower at k = 30 (LDS limits occupancy), 64 and 32 at k = 30 too (round 5: 3.54 / 6.20 ms against 2.49 at C3: the
// READY / PENDING rings need the depth); 64 at k = 60 (two workgroups per CU
// instead of one) measured 0.155 ns per draw against 0.124 with 128
__host__ __device__ inline int gq_problems_per_wave(uint32_t k) { (void)k; return 128; }
// Few problems in all (a single call: n of them, one preimage): a problem is a chain of k dependent draws, so the launch lasts as long as one chain
// however many problems a wave holds -- the host then gives every wave fewer of them (a power of two), down to one, to use more SIMDs.
__host__ inline int gq_problems_for(uint32_t k, size_t total) {
  int P = gq_problems_per_wave(k);
  while (P > 1 && total / (size_t)P < 2048) P >>= 1;
  return P;
}
__host__ inline size_t gadget_queue_lds_bytes(size_t k, int problems = 0) {
  const size_t P = (size_t)(problems > 0 ? problems : gq_problems_per_wave((uint32_t)k));
  const size_t tables = k * k * 8 + k * 8 + k * sizeof(SampleZParams) + 4 * k * 4 + k * k * 2;
  const size_t per_wave = k * P * 2 + P * 8 + P * 4 + P * 4 + P * 2 + P * 2;
  return tables + GQ_WAVES * per_wave + 64;
}

// FIXED: 128 problems per wave as a compile-time constant (full batches: the ring masks and strides fold; a run-time P cost 4 % there);
// otherwise Prt (a power of two below 128) for mid-size batches, see gq_problems_for
template <bool FIXED>
__global__ __launch_bounds__(256) void k_gadget_queue(uint64_t seed, uint64_t first_index, uint32_t n, uint32_t k, uint64_t q,
                                                      uint64_t base, size_t B, size_t ld, const uint64_t* __restrict__ V,
                                                      GadgetTablesQ tb, int8_t* __restrict__ Zlo, int8_t* __restrict__ Zhi,
                                                      int* __restrict__ fail, int Prt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gq_raw[];
  const int P = FIXED ? 128 : Prt;
  double* s_gso = reinterpret_cast<double*>(gq_raw);                       // k*k
  double* s_norm2 = s_gso + (size_t)k * k;                                 // k
  SampleZParams* s_sz = reinterpret_cast<SampleZParams*>(s_norm2 + k);     // k
  int32_t* s_rng = reinterpret_cast<int32_t*>(s_sz + k);                   // 4k
  int16_t* s_Sk = reinterpret_cast<int16_t*>(s_rng + 4 * k);               // k*k (entries of S_k are digits of q, base or -1: |.| < 2^15, checked by the host)
  unsigned char* wave_base = reinterpret_cast<unsigned char*>(s_Sk + (size_t)k * k);
  wave_base = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(wave_base) + 7) & ~(uintptr_t)7);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t per_wave = (size_t)k * P * 2 + (size_t)P * 8 + P * 4 + P * 4 + P * 2 + P * 2;
  unsigned char* wb = wave_base + (size_t)wave * per_wave;
  double* s_cen = reinterpret_cast<double*>(wb);                           // P
  int32_t* s_z = reinterpret_cast<int32_t*>(s_cen + P);                    // P
  int32_t* s_step = s_z + P;                                               // P
  uint16_t* s_ready = reinterpret_cast<uint16_t*>(s_step + P);             // P (ring)
  uint16_t* s_pend = s_ready + P;                                          // P (ring)
  int16_t* s_c = reinterpret_cast<int16_t*>(s_pend + P);                   // k * P, [row][problem]

  for (uint32_t e = tid; e < k * k; e += 256) { s_gso[e] = tb.gso[e]; s_Sk[e] = (int16_t)tb.Sk[e]; }
  for (uint32_t e = tid; e < k; e += 256) { s_norm2[e] = tb.norm2[e]; s_sz[e] = tb.sz[e]; }
  for (uint32_t e = tid; e < 4 * k; e += 256) s_rng[e] = tb.rng[e];
  __syncthreads();

  const size_t total = (size_t)n * B;
  const size_t seg0 = ((size_t)blockIdx.x * GQ_WAVES + wave) * (size_t)P;
  if (seg0 >= total) return;
  const int nprob = (int)(total - seg0 < (size_t)P ? total - seg0 : (size_t)P);
  int f = 0, anyhi = 0;
#define GQ_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

  auto centre = [&](int p, int col) -> double {     // <c, b~_col> / ||b~_col||^2 over the non-zero rows (ascending fma chain)
    double dot = 0.0;
    const int hi = s_rng[k + col];
    for (int r = s_rng[col]; r <= hi; ++r) dot = fma((double)s_c[r * P + p], s_gso[r * k + col], dot);
    return dot / s_norm2[col];
  };

  // ---- init: digits, first centre, everything READY
  for (int p = lane; p < nprob; p += 64) {
    const size_t pid = seg0 + p;
    const uint32_t j = (uint32_t)(pid / B);
    const size_t b = pid % B;
    uint64_t v = V[(size_t)j * ld + b] % q;
    for (uint32_t r = 0; r < k; ++r) {
      uint64_t d;
      if (base == 2) { d = v & 1; v >>= 1; }
      else { d = v % base; v = (v - d) / base; }
      s_c[r * P + p] = (int16_t)(-(int)d);
    }
    s_step[p] = (int)k - 1;
    s_cen[p] = centre(p, (int)k - 1);
    s_ready[p] = (uint16_t)p;
  }
  GQ_FENCE();
  int rhead = 0, rcount = nprob;          // READY ring (wave-uniform bookkeeping)
  int phead = 0, pcount = 0;              // PENDING ring
  int done = 0;

  const uint32_t pj0 = (uint32_t)(seg0 / B), pb0 = (uint32_t)(seg0 % B), B32 = (uint32_t)B;   // one 64-bit division per wave
  bool has = false;
  int myp = 0, mystep = 0;
  uint32_t t = 0, myj = 0;
  size_t myb = 0;
  double cen = 0.0;
  double inv_s = 0.0;                     // 1 / s of the problem's current step (the exact decisions); the other SampleZ parameters are used when the problem is taken only
  // everything of an attempt group that depends on the problem's current step only is formed when a lane takes the problem (round 5): the Philox key words,
  // the range in 32 bits and the fp32 offsets of the narrow screen (sz_group4_narrow) -- a group is then one Philox block and four screens
  SzRange rg{0, 1, 0, 16};
  uint32_t idx_lo = 0, tw = 0, coord = 0;
  float c_rel = 0.f, inv_s_f = 0.f;
  bool narrow = false;

  while (done < nprob) {
    // ---- idle lanes take READY problems
    {
      const uint64_t need = __ballot(!has);
      const int want = __popcll(need);
      const int take = want < rcount ? want : rcount;
      if (!has) {
        const int rk = lane_rank(need);
        if (rk < take) {
          myp = s_ready[(rhead + rk) & (P - 1)];
          mystep = s_step[myp];
          cen = s_cen[myp];
          const SampleZParams sp = s_sz[mystep];
          inv_s = sp.inv_s;
          const uint32_t o = pb0 + (uint32_t)myp;               // seg0 + myp = pj0 * B + o
          if (B32 >= (uint32_t)P) { const bool wrap = o >= B32; myj = pj0 + (wrap ? 1u : 0u); myb = wrap ? o - B32 : o; }
          else { myj = pj0 + o / B32; myb = o % B32; }
          t = 0;
          has = true;
          const uint64_t index = first_index + myb;
          idx_lo = (uint32_t)index;
          tw = tag_word(TAG_GADGET, index);
          coord = myj * k + (uint32_t)mystep;
          narrow = sp.sh == 16 && fabs(cen) < 0x1.0p30;
          if (narrow) {
            const double cc = ceil(cen);
            const bool integral = cc == cen;
            rg.lo = (long long)((int)cc - (int)sp.c6);
            rg.N = integral ? sp.n_int : sp.n_int - 1;
            rg.thr = integral ? sp.thr_int : sp.thr_frac;
            rg.sh = 16;
            c_rel = (float)((double)rg.lo - cen);
            inv_s_f = (float)sp.inv_s;
          } else rg = sz_range(cen, sp);
        }
      }
      rhead = (rhead + take) & (P - 1);
      rcount -= take;
    }
    // ---- four attempts (one Philox block when the range is narrow), screened in fp32: see sz_group4
    bool accept = false;
    long long x = 0;
    if (has) {
      accept = narrow ? sz_group4_narrow(seed, coord, idx_lo, tw, t, rg, cen, inv_s, c_rel, inv_s_f, &x)
                      : sz_group4(seed, coord, idx_lo, tw, t, rg, cen, inv_s, &x);
      if (!accept && ++t >= kMaxAttempts / 4) { accept = true; f = 1; x = (long long)floor(cen + 0.5); }
    }
    {
      const uint64_t mask = __ballot(accept);
      if (accept) {
        if (x > 16000 || x < -16000) f = 1;
        s_z[myp] = (int32_t)x;
        s_pend[(phead + pcount + lane_rank(mask)) & (P - 1)] = (uint16_t)myp;
        has = false;
      }
      pcount += __popcll(mask);
    }
    GQ_FENCE();
    // ---- advance a full wave of PENDING problems (or whatever is left once nothing else can run)
    const bool starving = !__ballot(has) && rcount == 0;
    // full batches advance 64 pending problems at a time; with P < 128 problems per wave (mid-size batches) the pass runs as soon as half of them are
    // pending -- otherwise it only ran when every problem had drawn, and a round lasted as long as its slowest draw
    const int adv = FIXED ? 64 : (P >= 128 ? 64 : (P >= 2 ? P / 2 : 1));
    while (pcount >= adv || (starving && pcount > 0)) {
      const int cnt = pcount < 64 ? pcount : 64;
      bool to_ready = false;
      int p = 0;
      if (lane < cnt) {
        p = s_pend[(phead + lane) & (P - 1)];
        int i = s_step[p];
        const int z = s_z[p];
        const int shi = s_rng[3 * k + i];
        for (int r = s_rng[2 * k + i]; r <= shi; ++r) {
          const int nv = (int)s_c[r * P + p] - z * (int)s_Sk[r * k + i];
          if (nv > 32767 || nv < -32768) f = 1;
          s_c[r * P + p] = (int16_t)nv;
        }
        --i;
        s_step[p] = i;
        if (i >= 0) {
          s_cen[p] = centre(p, i);
          to_ready = true;
        }
        // (a finished problem keeps its column of c in LDS; the digits are written once, behind the loop, with every lane at work -- round 5: written from here the
        // k-step store loop ran in nine passes of ten, for the one or two lanes whose problem had just finished, and was a third of the kernel's vector instructions)
      }
      const uint64_t rmask = __ballot(to_ready);
      if (to_ready) s_ready[(rhead + rcount + lane_rank(rmask)) & (P - 1)] = (uint16_t)p;
      const int nready = __popcll(rmask);
      rcount += nready;
      done += cnt - nready;
      phead = (phead + cnt) & (P - 1);
      pcount -= cnt;
      GQ_FENCE();
      if (!starving) break;                 // with lanes still sampling, one pass per iteration keeps everybody busy
    }
  }
  // ---- the digits z = -c of every problem of the wave
  GQ_FENCE();
  for (int p = lane; p < nprob; p += 64) {
    const uint32_t o = pb0 + (uint32_t)p;                     // seg0 + p = pj0 * B + o
    uint32_t j; size_t b;
    if (B32 >= (uint32_t)P) { const bool wrap = o >= B32; j = pj0 + (wrap ? 1u : 0u); b = wrap ? o - B32 : o; }
    else { j = pj0 + o / B32; b = o % B32; }
    for (uint32_t r = 0; r < k; ++r) {
      const int32_t zz = -(int32_t)s_c[r * P + p];
      const int32_t zl = (int32_t)(int8_t)(zz & 0xff);
      const int32_t zh = (zz - zl) >> 8;
      const size_t c = (size_t)j * k + r;
      const size_t addr = ((c >> 4) * ld + b) * 16 + (c & 15);
      Zlo[addr] = (int8_t)zl;
      Zhi[addr] = (int8_t)zh;
      if (zh) anyhi = 1;
    }
  }
#undef GQ_FENCE
  if (f) atomicOr(fail, 1);
  if (anyhi) atomicOr(fail + 1, 1);
}

// ---- e = p + [R; I] z, written preimage-major (B x m int64) ---------------------------------------------
// top part on the int8 matrix cores: C[b][i] = sum_c Z[c][b] R[i][c] with v_mfma_i32_16x16x64_i8, Z as the A operand
// (rows = preimages) and R as the B operand (columns = coordinates i), so that 16 lanes hold 16 consecutive i of one
// preimage and the int64 stores of e[b][i] are 128-byte runs.  Both operands use the same (lane group, byte) -> c map,
// which is all the dot product needs.  Workgroup tile 128 (b) x 128 (i), wave tile 64 x 64, K step 64, LDS-DMA staging
// (2 stages x (R tile 8 KiB | Zlo 8 KiB | Zhi 8 KiB)).  The hi plane is skipped when the gadget kernel saw no |z| > 127.
constexpr int RC_STAGE = 3 * 8192;
constexpr int RC_LDS = 3 * RC_STAGE;      // 72 KiB: three 24 KiB stages, or four 16 KiB stages when the hi plane is unused

__global__ __launch_bounds__(256, 2) void k_recombine_mfma(const int8_t* __restrict__ R, size_t ldr, size_t mbar, int nks,
                                                           const int8_t* __restrict__ Zlo, const int8_t* __restrict__ Zhi, size_t ld,
                                                           const int* __restrict__ flags, const int32_t* __restrict__ P, size_t B,
                                                           int64_t* __restrict__ E, size_t m, int only_hi, int ks_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const size_t b0 = (size_t)blockIdx.x * 128, i0 = (size_t)blockIdx.y * 128;
  const bool use_hi = flags[1] != 0;
  if (only_hi && !use_hi) return;                     // k_recombine_mfma_big (launched in front) has done the lo-plane-only case
  // gridDim.z > 1 (few preimages: a single call has one column tile, and the launch would be 121 workgroups streaming 2 MB of R each): blockIdx.z takes the
  // K steps [z ks_per_split, (z + 1) ks_per_split) and ADDS its integer partial sum to E (zeroed by the host; z = 0 also adds p) -- exact in any order.
  const bool split = gridDim.z > 1;
  if (split) {
    const int ks0 = (int)blockIdx.z * ks_per_split;
    if (ks0 >= nks) return;
    R += (size_t)ks0 * 64; Zlo += (size_t)ks0 * 4 * ld * 16; Zhi += (size_t)ks0 * 4 * ld * 16;
    nks = nks - ks0 < ks_per_split ? nks - ks0 : ks_per_split;
  }

  v4i alo[4][4], ahi[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) { alo[x][y] = v4i{0, 0, 0, 0}; ahi[x][y] = v4i{0, 0, 0, 0}; }

  // per-lane global sources of the two 1 KiB pieces each wave moves per array and stage
  const int8_t* srcR[2]; const int8_t* srcL[2]; const int8_t* srcH[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = (wave * 2 + j) * 64 + lane;                 // 16-byte piece index, 0..511: LDS slot (row p / 4, position p % 4) takes k group (position - row / 4) mod 4
    srcR[j] = R + (i0 + (size_t)(p >> 2)) * ldr + (size_t)(((p & 3) - (p >> 4)) & 3) * 16;
    const size_t zoff = ((size_t)(p >> 7) * ld + b0 + (size_t)(p & 127)) * 16;
    srcL[j] = Zlo + zoff;
    srcH[j] = Zhi + zoff;
  }
  // A K step is only 16 (32 with the hi plane) short MFMAs, far less than a DMA round trip, so the stages form a ring of depth
  // 4 (16 KiB stages, hi plane unused) or 3 (24 KiB stages) inside the same RC_LDS bytes: NS - 1 stages are always in flight.
  const int NS = use_hi ? 3 : 4;
  const int SS = use_hi ? 3 * 8192 : 2 * 8192;
  auto stage_load = [&](int ks, int buf) {
    unsigned char* base = rc_smem + buf * SS;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece0 = (wave * 2 + j) * 64;
      __builtin_amdgcn_global_load_lds(srcR[j] + (size_t)ks * 64, (lds_void_ptr)(base + piece0 * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(srcL[j] + (size_t)ks * 4 * ld * 16, (lds_void_ptr)(base + 8192 + piece0 * 16), 16, 0, 0);
      if (use_hi)
        __builtin_amdgcn_global_load_lds(srcH[j] + (size_t)ks * 4 * ld * 16, (lds_void_ptr)(base + 16384 + piece0 * 16), 16, 0, 0);
    }
  };
  const int r16 = lane & 15, g = lane >> 4;
  if (!use_hi && (nks & 1) == 0) {
    // Common case (no |z| > 127): K = 128 per stage -- 32 MFMAs per wave between two barriers instead of 16 -- in two 32 KiB stages:
    // R tile as [k half 2][row 128][64 B], Z tile as [k group 8][preimage 128][16 B].
    auto stage_load2 = [&](int ks2, int buf) {
      unsigned char* base = rc_smem + buf * 32768;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = (wave * 4 + j) * 64 + lane;                  // 16-byte piece, 0..1023
        const int kk = p >> 9, row = (p >> 2) & 127, col = ((p & 3) - (row >> 2)) & 3;      // the k group this LDS position holds (i8_slot inverted)
        __builtin_amdgcn_global_load_lds(R + (i0 + (size_t)row) * ldr + (size_t)ks2 * 128 + kk * 64 + col * 16,
                                         (lds_void_ptr)(base + (wave * 4 + j) * 1024), 16, 0, 0);
        const int kg = p >> 7, bb = p & 127;
        __builtin_amdgcn_global_load_lds(Zlo + (((size_t)ks2 * 8 + kg) * ld + b0 + (size_t)bb) * 16,
                                         (lds_void_ptr)(base + 16384 + (wave * 4 + j) * 1024), 16, 0, 0);
      }
    };
    const int n2 = nks / 2;
    stage_load2(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ks2 = 0; ks2 < n2; ++ks2) {
      const int cb = ks2 & 1;
      if (ks2 + 1 < n2) stage_load2(ks2 + 1, cb ^ 1);
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
          for (int it = 0; it < 4; ++it) alo[bt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fl[bt], fr[it], alo[bt][it], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
  for (int s0 = 0; s0 < NS - 1 && s0 < nks; ++s0) stage_load(s0, s0);
  int cur = 0;                                        // ks % NS
  for (int ks = 0; ks < nks; ++ks) {
    // stage ks has landed when at most the NS - 2 younger stages (4 or 6 DMA instructions each) are outstanding
    if (nks - 1 - ks >= NS - 2) {
      if (use_hi) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                  // everybody's part of stage ks is in LDS; everybody is done with stage ks - 1
    const int nxt = ks + NS - 1;
    if (nxt < nks) stage_load(nxt, cur == 0 ? NS - 1 : cur - 1);
    const unsigned char* sR = rc_smem + cur * SS;
    const unsigned char* sL = sR + 8192;
    const unsigned char* sH = sR + 16384;
    v4i fr[4], fl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fr[t] = *reinterpret_cast<const v4i*>(sR + ((wc * 64 + t * 16 + r16) * 64 + i8_slot(wc * 64 + t * 16 + r16, g) * 16));
      fl[t] = *reinterpret_cast<const v4i*>(sL + ((g * 128 + wr * 64 + t * 16 + r16) * 16));
    }
#pragma unroll
    for (int bt = 0; bt < 4; ++bt)
#pragma unroll
      for (int it = 0; it < 4; ++it) alo[bt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fl[bt], fr[it], alo[bt][it], 0, 0, 0);
    if (use_hi) {
      v4i fh[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fh[t] = *reinterpret_cast<const v4i*>(sH + ((g * 128 + wr * 64 + t * 16 + r16) * 16));
#pragma unroll
      for (int bt = 0; bt < 4; ++bt)
#pragma unroll
        for (int it = 0; it < 4; ++it) ahi[bt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fh[bt], fr[it], ahi[bt][it], 0, 0, 0);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  }
  // The 128 x 128 tile of p is brought in through LDS: rows of P (coordinate-major, preimages contiguous) are read as 16-byte
  // runs and turned by the padded LDS tile, instead of sixty-four 4-byte gathers per thread across sixteen rows each.
  __syncthreads();                                    // the stage ring is free
  int32_t* sP = reinterpret_cast<int32_t*>(rc_smem);  // [128][129]
  const bool add_p = !split || blockIdx.z == 0;       // workgroup-uniform
  for (int idx = tid; idx < 128 * 32; idx += 256) {
    const int ii = idx >> 5, c4 = idx & 31;
    const int4 v = add_p ? *reinterpret_cast<const int4*>(P + (i0 + (size_t)ii) * ld + b0 + (size_t)c4 * 4) : int4{0, 0, 0, 0};
    int32_t* d = sP + ii * 129 + c4 * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  // C/D map: column (here i) = lane & 15, row (here b) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int bt = 0; bt < 4; ++bt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bl = wr * 64 + bt * 16 + 4 * g + r, il = wc * 64 + it * 16 + r16;
        const size_t bb = b0 + bl, ii = i0 + il;
        if (bb < B && ii < mbar) {
          const int64_t v = (int64_t)sP[il * 129 + bl] + (int64_t)alo[bt][it][r] + 256 * (int64_t)ahi[bt][it][r];
          if (split) atomicAdd(reinterpret_cast<unsigned long long*>(E + bb * m + ii), (unsigned long long)v);
          else E[bb * m + ii] = v;
        }
      }
}

// The common case (no |z| > 127: one digit plane) on 256 x 256 workgroup tiles: eight waves, wave (wr, wc) the 128 (b) x 64 (i) piece in 32 accumulator
// tiles = 128 registers; one workgroup per CU; two 64 KiB stages (R tile [k half 2][row 256][64 B, k groups rotated by i8_slot] | Z tile [k group 8][preimage 256]
// [16 B]) filled by LDS-DMA, K = 128 (64 MFMAs per wave) between two barriers.  p is added straight from global memory: the four preimages of a lane's
// accumulator tile are one 16-byte run of a row of P.
// Measured at C3 (profiles/r03_notes.md, "int8 recombination: what bounds it"): 1.064 ms against 1.081 ms for the 128 x 128 kernel above -- half the bytes
// staged (7.6 instead of 15 GB) buy 1.6 %, so the L2 -> LDS stream round 2 blamed is not the bound.  Ablation of this kernel (stores off, 0.93 ms): without the
// LDS fragment reads 0.96, without the LDS-DMA 0.65, without both 0.51 (the MFMAs alone: 0.41 at 16 cycles each); the epilogue alone 0.19.  The eight DMA
// pieces a wave issues per K step are what the step waits for, however they are placed (in front of the MFMAs, spread over them by the scheduler, or replaced by
// global loads into registers + ds_write_b128: 1.08 / 1.08 / 1.13 ms); double-buffered fragment reads change nothing (1.08).
constexpr int RCB_STAGE = 65536;
constexpr int RCB_LDS = 2 * RCB_STAGE;
// packed != 0: R is the tile-packed copy (k_pack_R8: [row tile of 64][k step of 64][row][64 bytes, k groups rotated as in the LDS image]): the R half of a stage is
// thirty-two CONTIGUOUS 1 KiB pieces instead of 512 row segments of 64 bytes (tools/probe_ldsdma_l2.hip: LDS-DMA fills from L2 run at 65 GB/s per CU contiguous,
// 44 GB/s with the 64-byte segments of the row-major fetch; 76 GB/s is what this tile needs to keep its MFMAs busy)
__global__ __launch_bounds__(512, 1) void k_recombine_mfma_big(const int8_t* __restrict__ R, int packed, size_t ldr, size_t mbar, int n2,
                                                               const int8_t* __restrict__ Zlo, size_t ld, const int* __restrict__ flags,
                                                               const int32_t* __restrict__ P, size_t B, int64_t* __restrict__ E, size_t m,
                                                               unsigned nbx, unsigned nby) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rcb_smem[];
  if (flags[1] != 0) return;                          // a second digit plane is in use: k_recombine_mfma (launched behind) takes the call
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // the 32 workgroups an XCD holds at a time (blockIdx % 8 = XCD, one workgroup per CU) form a super-tile of 4 preimage tiles x 8 coordinate tiles: every R
  // tile is fetched into that XCD's L2 once for four workgroups, every Z tile once for eight
  const unsigned nbg = (nbx + 3) / 4, nig = (nby + 7) / 8;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned sup = (slot >> 5) * 8u + xcd, tt = slot & 31u;
  const unsigned bx = (sup % nbg) * 4u + (tt & 3u), by = (sup / nbg) * 8u + (tt >> 2);
  if (sup >= nbg * nig || bx >= nbx || by >= nby) return;
  const size_t b0 = (size_t)bx * 256, i0 = (size_t)by * 256;
  v4i acc[8][4];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = v4i{0, 0, 0, 0};
  // 16-byte pieces of a stage: 2048 of R and 2048 of Z, four of each per thread; piece p of R = (k half p / 1024, row (p / 4) % 256, position p % 4) and
  // holds k group (position - row / 4) mod 4; piece p of Z = (k group p / 256, preimage p % 256)
  const int8_t* srcR[4]; const int8_t* srcZ[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (wave * 4 + j) * 64 + lane;
    const int kk = p >> 10, row = (p >> 2) & 255, col = ((p & 3) - (row >> 2)) & 3;
    const int q = wave * 4 + j;                        // piece q of the stage's R half = LDS bytes [q KiB, q + 1 KiB): (k half q / 16, row tile (q / 4) % 4, quarter q % 4)
    srcR[j] = packed ? R + ((i0 / 64 + (size_t)((q >> 2) & 3)) * (ldr / 64) + (size_t)(q >> 4)) * 4096 + (size_t)(q & 3) * 1024 + (size_t)lane * 16
                     : R + (i0 + (size_t)row) * ldr + (size_t)(kk * 64 + col * 16);
    srcZ[j] = Zlo + ((size_t)(p >> 8) * ld + b0 + (size_t)(p & 255)) * 16;
  }
  const size_t rstep = packed ? 8192 : 128;            // a K step of 128: two 4 KiB tiles of the packed copy / 128 bytes of a row
  auto stage_load = [&](int ks2, int buf) {
    unsigned char* base = rcb_smem + buf * RCB_STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds(srcR[j] + (size_t)ks2 * rstep, (lds_void_ptr)(base + (wave * 4 + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(srcZ[j] + (size_t)ks2 * 8 * ld * 16, (lds_void_ptr)(base + 32768 + (wave * 4 + j) * 1024), 16, 0, 0);
    }
  };
  const int r16 = lane & 15, g = lane >> 4;
  stage_load(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks2 = 0; ks2 < n2; ++ks2) {
    const int cb = ks2 & 1;
    if (ks2 + 1 < n2) stage_load(ks2 + 1, cb ^ 1);
    const unsigned char* sR = rcb_smem + cb * RCB_STAGE;
    const unsigned char* sL = sR + 32768;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      v4i fr[4], fl[8];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ri = wc * 64 + t * 16 + r16;
        fr[t] = *reinterpret_cast<const v4i*>(sR + kk * 16384 + (ri * 64 + i8_slot(ri, g) * 16));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) fl[t] = *reinterpret_cast<const v4i*>(sL + (((kk * 4 + g) * 256 + wr * 128 + t * 16 + r16) * 16));
#pragma unroll
      for (int bt = 0; bt < 8; ++bt)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[bt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fl[bt], fr[it], acc[bt][it], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // C/D map: column (here i) = lane & 15, row (here b) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int bt = 0; bt < 8; ++bt) {
    const size_t bb = b0 + (size_t)(wr * 128 + bt * 16 + 4 * g);
    int4 pv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const size_t ii = i0 + (size_t)(wc * 64 + it * 16 + r16);
      pv[it] = (ii < mbar && bb < B) ? *reinterpret_cast<const int4*>(P + ii * ld + bb) : int4{0, 0, 0, 0};
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const size_t ii = i0 + (size_t)(wc * 64 + it * 16 + r16);
      if (ii >= mbar) continue;
      const int pr[4] = {pv[it].x, pv[it].y, pv[it].z, pv[it].w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (bb + r < B) E[(bb + r) * m + ii] = (int64_t)pr[r] + (int64_t)acc[bt][it][r];
    }
  }
}

// bottom part: e[b][mbar + c] = p[mbar + c][b] + z[c][b]
// zero_top: also clear e[b][c] for c < mbar (the split-K form of k_recombine_mfma, launched behind, adds its partial sums there); the grid then covers
// max(w, mbar) columns
__global__ __launch_bounds__(256) void k_recombine_bottom(size_t mbar, size_t w, const int8_t* __restrict__ Zlo,
                                                          const int8_t* __restrict__ Zhi, size_t ld, const int32_t* __restrict__ P,
                                                          size_t B, int64_t* __restrict__ E, size_t m, int zero_top) {
  // a 64 x 64 tile (preimages x coordinates) through LDS as int32 (|p| < 2^23, |z| < 2^15): z arrives as the planes' own 16-byte groups (one load per plane,
  // preimage and group, not sixteen byte loads), p coalesced along the preimages, e leaves as whole rows
  __shared__ int sE[64][65];
  const int tid = threadIdx.x;
  const size_t c0 = (size_t)blockIdx.y * 64, b0 = (size_t)blockIdx.x * 64;
  if (zero_top) {
    for (int e = tid; e < 64 * 64; e += 256) {
      const int bb = e >> 6, cc = e & 63;
      if (b0 + bb < B && c0 + cc < mbar) E[(b0 + bb) * m + c0 + cc] = 0;
    }
  }
  if (c0 >= w) return;                                // workgroup-uniform
  {
    const int g = tid >> 6, bb = tid & 63;            // group g of the tile's four, preimage bb
    const size_t at = ((c0 / 16 + (size_t)g) * ld + b0 + (size_t)bb) * 16;
    int4 zl = make_int4(0, 0, 0, 0), zh = make_int4(0, 0, 0, 0);
    if (c0 + 16 * (size_t)g < w) { zl = *reinterpret_cast<const int4*>(Zlo + at); zh = *reinterpret_cast<const int4*>(Zhi + at); }      // (b0 + bb < ld always: the planes are ld wide)
    const int lo[4] = {zl.x, zl.y, zl.z, zl.w}, hi[4] = {zh.x, zh.y, zh.z, zh.w};
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sE[bb][g * 16 + q4 * 4 + r] = (int)(int8_t)(lo[q4] >> (8 * r)) + 256 * (int)(int8_t)(hi[q4] >> (8 * r));
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int cc = e >> 6, bb = e & 63;
    const size_t c = c0 + cc;
    if (c < w) sE[bb][cc] += P[(mbar + c) * ld + b0 + bb];
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int bb = e >> 6, cc = e & 63;
    if (b0 + bb < B && c0 + cc < w) E[(b0 + bb) * m + mbar + c0 + cc] = (int64_t)sE[bb][cc];
  }
}

// ---- samp_d, check_domain, narrowing transpose for f_a --------------------------------------------------
__global__ void k_samp_d(uint64_t seed, uint64_t first_index, size_t m, size_t B, SampleZParams sp, int64_t* __restrict__ E,
                         int* __restrict__ fail) {
  const size_t total = m * B;
  int f = 0;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / m, i = g % m;
    E[g] = sample_z(seed, TAG_SAMPD, first_index + b, (uint32_t)i, 0.0, sp, &f);
  }
  if (f) atomicOr(fail, 1);
}

// ok[b] = (||e_b||^2 <= bound), exact 128-bit norm (mp_perturbation.rs:396-402)
__global__ __launch_bounds__(256) void k_check_domain(const int64_t* __restrict__ E, size_t len, size_t m, double bound,
                                                      uint8_t* __restrict__ ok) {
  __shared__ uint64_t s_lo[256];
  __shared__ uint64_t s_hi[256];
  const size_t b = blockIdx.x;
  uint64_t lo = 0, hi = 0;
  for (size_t i = threadIdx.x; i < len; i += 256) {
    const int64_t v = E[b * len + i];
    const uint64_t a = (uint64_t)(v < 0 ? -v : v);
    const uint64_t pl = a * a, ph = __umul64hi(a, a);
    const uint64_t nl = lo + pl;
    hi += ph + (nl < lo ? 1 : 0);
    lo = nl;
  }
  s_lo[threadIdx.x] = lo; s_hi[threadIdx.x] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const uint64_t nl = s_lo[threadIdx.x] + s_lo[threadIdx.x + s];
      s_hi[threadIdx.x] += s_hi[threadIdx.x + s] + (nl < s_lo[threadIdx.x] ? 1 : 0);
      s_lo[threadIdx.x] = nl;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double nn = (double)s_hi[0] * 18446744073709551616.0 + (double)s_lo[0];
    ok[b] = (len == m && nn <= bound) ? 1 : 0;
  }
}

// e (B x m int64) -> P layout (m x ld int32).  The int8-MFMA product cuts p into three balanced base-256 digits, i.e. |p| < 2^23;
// psfp_create refuses parameter sets with s r sqrt(m) >= 2^23, so a coordinate beyond that is outside D_n (its row already failed
// check_domain): it is clamped and the row's ok flag is cleared again here, so f_a can never report ok for a truncated row.
constexpr int64_t kDigitRange = (1ll << 23) - 1;
__global__ __launch_bounds__(256) void k_narrow_transpose(const int64_t* __restrict__ E, size_t m, size_t B, size_t ld,
                                                          int32_t* __restrict__ P, uint8_t* __restrict__ ok) {
  __shared__ int32_t s[64][65];
  const int tid = threadIdx.x;
  const size_t i0 = (size_t)blockIdx.y * 64, b0 = (size_t)blockIdx.x * 64;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int bb = e >> 6, ii = e & 63;
    int64_t v = 0;
    if (b0 + bb < B && i0 + ii < m) v = E[(b0 + bb) * m + i0 + ii];
    if (v > kDigitRange || v < -kDigitRange) {
      v = v > 0 ? kDigitRange : -kDigitRange;
      ok[b0 + bb] = 0;
    }
    s[ii][bb] = (int32_t)v;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int ii = e >> 6, bb = e & 63;
    if (i0 + ii < m && b0 + bb < ld) P[(i0 + ii) * ld + b0 + bb] = s[ii][bb];
  }
}

// P (m x ld int32) -> B x m int64 (stage export)
__global__ void k_export_P(const int32_t* __restrict__ P, size_t m, size_t B, size_t ld, int64_t* __restrict__ out) {
  const size_t total = m * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / m, i = g % m;
    out[g] = P[i * ld + b];
  }
}
template <typename T>
__global__ void k_export_T(const T* __restrict__ S, size_t rows, size_t B, size_t ld, T* __restrict__ out) {
  const size_t total = rows * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / rows, i = g % rows;
    out[g] = S[i * ld + b];
  }
}

__global__ void k_uniform_targets(uint64_t seed, uint64_t first_index, size_t n, size_t B, uint64_t q, uint64_t* __restrict__ U) {
  const size_t total = n * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g / n, i = g % n;
    U[g] = uniform_mod(seed, TAG_TARGET, (uint32_t)i, (uint32_t)(first_index + b), q);
  }
}

// ---- trap_gen pieces ---------------------------------------------------------------------------------------
// A[i][j] = uniform(Z_q), j < m_bar  (mp_perturbation.rs:222)
__global__ void k_sample_abar(uint64_t seed, size_t n, size_t mbar, size_t lda, uint64_t q, uint64_t* __restrict__ A) {
  const size_t total = n * mbar;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / mbar, j = g % mbar;
    A[i * lda + j] = uniform_mod(seed, TAG_ABAR, (uint32_t)j, (uint32_t)i, q);
  }
}
// PlusMinusOneZero (trapdoor_distribution.rs:82-86): 64 entries per Philox block
__global__ void k_sample_R(uint64_t seed, size_t mbar, size_t w, size_t ldr, int8_t* __restrict__ R) {
  const size_t total = mbar * ldr;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g / ldr, j = g % ldr;
    int8_t v = 0;
    if (j < w) {
      const U4 wd = philox(seed, (uint32_t)(j >> 6), (uint32_t)i, 0, TAG_R);
      const uint32_t sel = (uint32_t)(j & 63) >> 4;
      const uint32_t word = sel == 0 ? wd.x : sel == 1 ? wd.y : sel == 2 ? wd.z : wd.w;
      const uint32_t sh = 2 * (uint32_t)(j & 15);
      v = (int8_t)((int)((word >> sh) & 1) - (int)((word >> (sh + 1)) & 1));
    }
    R[g] = v;
  }
}

// Sigma_2 = (r^2/2pi) ((Sigma - (b^2+1) T T^t) - I), T = [R; I]  (mp_perturbation.rs:111-136), lower triangle of a
// dense row-major m x m matrix.  Sigma = s^2 I (Sig == nullptr: the form trap_gen passes, :227-231) or any symmetric matrix given as its
// packed lower triangle (row i: i + 1 entries).  The R R^t block is a dot4 product of 64 x 64 row pairs.
__global__ __launch_bounds__(256) void k_sigma2(const int8_t* __restrict__ R, size_t ldr, size_t mbar, size_t w, size_t m,
                                                double nf_r2, double s2, double b2p1, const double* __restrict__ Sig, double* __restrict__ S, size_t lds,
                                                size_t row_off, size_t col_off, int skip_rrt) {
  __shared__ uint32_t sRi[64][17];
  __shared__ uint32_t sRj[64][17];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  // (row_off, col_off): the tile grid covers rows row_off.. and columns col_off.. of Sigma_2 and S holds that window (the panel-wise Cholesky)
  const size_t i0 = row_off + (size_t)blockIdx.y * 64, j0 = col_off + (size_t)blockIdx.x * 64;
  if (j0 > i0 + 63) return;  // strictly upper tile
  if (skip_rrt && i0 + 63 < mbar) return;                // skip_rrt: the entries with i, j < m_bar come from k_sigma2_rrt (int8 matrix cores)
  int32_t acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0;
  if (!skip_rrt && i0 < mbar && j0 < mbar) {
    const uint32_t* R32 = reinterpret_cast<const uint32_t*>(R);
    const size_t ldr4 = ldr / 4, w4 = (w + 3) / 4;
    for (size_t q0 = 0; q0 < w4; q0 += 16) {
      for (int e = tid; e < 64 * 16; e += 256) {
        const int r = e >> 4, qq = e & 15;
        uint32_t vi = 0, vj = 0;
        if (q0 + qq < w4) {
          if (i0 + r < mbar) vi = R32[(i0 + r) * ldr4 + q0 + qq];
          if (j0 + r < mbar) vj = R32[(j0 + r) * ldr4 + q0 + qq];
        }
        sRi[r][qq] = vi; sRj[r][qq] = vj;
      }
      __syncthreads();
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        uint32_t rj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) rj[c] = sRj[tx * 4 + c][qq];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t ri = sRi[ty * 4 + r][qq];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_sdot4((int)ri, (int)rj[c], acc[r][c], false);
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t i = i0 + ty * 4 + r, j = j0 + tx * 4 + c;
      if (i >= m || j > i) continue;
      if (skip_rrt && i < mbar) continue;
      double tt;
      if (i < mbar) tt = (double)acc[r][c];
      else if (j < mbar) tt = (double)R[j * ldr + (i - mbar)];
      else tt = (i == j) ? 1.0 : 0.0;
      const double sg = Sig ? Sig[i * (i + 1) / 2 + j] : ((i == j) ? s2 : 0.0);
      double sp = sg - b2p1 * tt;
      if (i == j) sp = sp - 1.0;
      S[(i - row_off) * lds + (j - col_off)] = nf_r2 * sp;
    }
}

// The R R^t block of Sigma_2 (rows and columns below m_bar: at C3 half the entries and all of the arithmetic, m_bar^2 w / 2 = 1.8e12 MAC; C5: 1.2e14) on
// the int8 matrix cores: R is ternary, so v_mfma_i32_16x16x64_i8 sums exactly.  Both operands are row tiles of R itself -- a row of R is 16 consecutive k for
// the A fragment and for the B fragment alike.  64 x 64 output tile per workgroup (4 waves, 2 x 2 of 32 x 32), K steps of 64 staged straight from the
// row-major R by LDS-DMA (each lane moves one 16-byte k group; the groups of a row are stored rotated by row / 4, i8_slot, so that the fragment reads are
// conflict free), ring of three stages.  Writes the same window and the same expression as k_sigma2.
__global__ __launch_bounds__(256, 2) void k_sigma2_rrt(const int8_t* __restrict__ R, size_t ldr, size_t mbar, size_t m, double nf_r2, double s2, double b2p1,
                                                       const double* __restrict__ Sig, double* __restrict__ S, size_t lds, size_t row_off, size_t col_off) {
  constexpr int STAGE = 2 * 4096, NS = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char rr_smem[];
  const size_t i0 = row_off + (size_t)blockIdx.y * 64, j0 = col_off + (size_t)blockIdx.x * 64;
  if (j0 > i0 + 63 || i0 >= mbar || j0 >= mbar) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nks = (int)(ldr / 64);
  v4i acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = v4i{0, 0, 0, 0};
  // piece p = tid of a 64 x 64-byte tile: LDS slot (row p / 4, position p % 4) takes k group (position - row / 4) mod 4
  const size_t koff = (size_t)(((tid & 3) - (tid >> 4)) & 3) * 16;
  const int8_t* srcA = R + (i0 + (size_t)(tid >> 2)) * ldr + koff;
  const int8_t* srcB = R + (j0 + (size_t)(tid >> 2)) * ldr + koff;
  auto stage_load = [&](int ks, int buf) {
    unsigned char* base = rr_smem + buf * STAGE + wave * 1024;
    __builtin_amdgcn_global_load_lds(srcA + (size_t)ks * 64, (lds_void_ptr)base, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(srcB + (size_t)ks * 64, (lds_void_ptr)(base + 4096), 16, 0, 0);
  };
  for (int s0 = 0; s0 < NS - 1 && s0 < nks; ++s0) stage_load(s0, s0);
  const int r16 = lane & 15, gq = lane >> 4;
  int cur = 0;
  for (int ks = 0; ks < nks; ++ks) {
    // stage ks has landed when at most the NS - 2 younger stages (2 DMA instructions each) are outstanding
    if (nks - 1 - ks >= NS - 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // everybody's part of stage ks is in LDS; everybody is done with stage ks - 1
    const int nxt = ks + NS - 1;
    if (nxt < nks) stage_load(nxt, cur == 0 ? NS - 1 : cur - 1);
    const unsigned char* sb = rr_smem + cur * STAGE;
    v4i fa[2], fb[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      fa[x] = *reinterpret_cast<const v4i*>(sb + ((wr * 32 + x * 16 + r16) * 64 + i8_slot(wr * 32 + x * 16 + r16, gq) * 16));
      fb[x] = *reinterpret_cast<const v4i*>(sb + 4096 + ((wc * 32 + x * 16 + r16) * 64 + i8_slot(wc * 32 + x * 16 + r16, gq) * 16));
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[x], fb[y], acc[x][y], 0, 0, 0);
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  // C/D map: column (here j) = lane & 15, row (here i) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t i = i0 + wr * 32 + x * 16 + 4 * gq + r, j = j0 + wc * 32 + y * 16 + r16;
        if (i >= mbar || j > i || i >= m) continue;
        const double sg = Sig ? Sig[i * (i + 1) / 2 + j] : ((i == j) ? s2 : 0.0);
        double sp = sg - b2p1 * (double)acc[x][y][r];
        if (i == j) sp = sp - 1.0;
        S[(i - row_off) * lds + (j - col_off)] = nf_r2 * sp;
      }
}

}  // namespace psf
