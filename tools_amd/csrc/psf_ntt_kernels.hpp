// psf_ntt_kernels.hpp -- device back end and kernels of the wave-level negacyclic NTT (psf_ntt_core.hpp):
//   k_ntt_polymul      out = a * b in Z_q[X]/(X^n + 1), one product per wave (PolynomialRingZq products: gadget_ring.rs:78, gpv_ring.rs:245-246)
//   k_ntt_forward      a -> its leaf residues in the register image [r][lane] (a key polynomial is transformed ONCE)
//   k_ntt_mul_hat      out = a * b with a given by its image
//   k_ring_fa          u = sum_j a_j * sigma_j (PSFGPVRing::f_a, gpv_ring.rs:243-247) from the cached images of a: K forward transforms,
//                      K leaf products accumulated, ONE inverse transform per preimage
// 16-bit I/O (a: uint16 in [0, q), b: int16 in (-q, q), out: uint16) beside the 64-bit ABI of psf_poly_mul_negacyclic.
#pragma once
#include <hip/hip_runtime.h>
#include "psf_ntt_core.hpp"

namespace psf {
namespace ntt {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct DevWave {
  using I = int;
  using U = unsigned;
  using Tab = const uint32_t*;                                        // LDS
  static __device__ __forceinline__ I lane() { return (I)__lane_id(); }
  static __device__ __forceinline__ I izero() { return 0; }
  static __device__ __forceinline__ U uzero() { return 0u; }
  static __device__ __forceinline__ I sra(I x, int s) { return x >> s; }
  static __device__ __forceinline__ I srl(I x, int s) { return (I)((U)x >> s); }
  static __device__ __forceinline__ I shl(I x, int s) { return (I)((U)x << s); }
  // Montgomery reduction with R = 2^16: two full-rate multiplies and a shift.  v_mad_i32_i16 reads the LOW HALVES of its factors as signed 16-bit
  // values, so the low half of t q^-1 needs no sign extension; as an asm statement it also keeps hipcc from turning the 24-bit multiply in front of
  // it into a quarter-rate v_mul_lo_u32 ("only 16 bits are used").
  static __device__ __forceinline__ I mont16(I t, I qinv, I nq) {
    const I m = __mul24(t, qinv);
    I r;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(m), "s"(nq), "v"(t));
    return r >> 16;
  }
  static __device__ __forceinline__ I mul24(I a, I b) { return __mul24(a, b); }
  static __device__ __forceinline__ I mad24(I a, I b, I c) { return __mul24(a, b) + c; }
  static __device__ __forceinline__ U mullo_u(U a, U b) { return a * b; }
  static __device__ __forceinline__ U mulhi_u(U a, U b) { return __umulhi(a, b); }
  static __device__ __forceinline__ U nonzero(U x) { return x != 0u ? 1u : 0u; }
  static __device__ __forceinline__ U csub(U r, U q) { return r >= q ? r - q : r; }
  static __device__ __forceinline__ U cadd(U x, U q) { return x + (q & (U)((I)x >> 31)); }
  template <class V> static __device__ __forceinline__ V tab(Tab t, I idx, int off) { return (V)t[idx + off]; }
  template <class V> static __device__ __forceinline__ V tab_const(Tab t, int idx) { return (V)t[idx]; }
  template <class V> static __device__ __forceinline__ V sel_odd(I lane, V a, V b) { return (lane & 1) ? a : b; }
  static __device__ __forceinline__ I umin(I a, I b) { return (I)((U)a < (U)b ? (U)a : (U)b); }
  // pairs [(z | -q), z q^-1] behind `zoff` words of plain tables: one 8-byte LDS read
  template <class V> static __device__ __forceinline__ void tab_pair(Tab t, int zoff, I idx, int off, V& pk, V& zq) {
    const uint2 v = *reinterpret_cast<const uint2*>(t + zoff + 2 * (idx + off));
    pk = (V)v.x; zq = (V)v.y;
  }
  template <class V> static __device__ __forceinline__ void tab_pair_const(Tab t, int zoff, int idx, V& pk, V& zq) {
    const uint2 v = *reinterpret_cast<const uint2*>(t + zoff + 2 * idx);
    pk = (V)v.x; zq = (V)v.y;
  }
  // (z x - m q) >> 16 with m = low half of x (z q^-1): m goes into the upper half of x's register (SDWA, lower half preserved), then one dot product
  // of the halves with (z | -q).  Wait states by hand (asm is opaque to the hazard recogniser): one between an SDWA write of a half and its reader,
  // three between a dot product and a vector read of its result.
  static __device__ __forceinline__ I dot2mont(I x, I zq, I pk) {
    asm("v_mul_lo_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n\ts_nop 0\n\t"
        "v_dot2_i32_i16 %0, %0, %2, 0\n\ts_nop 2"
        : "+v"(x) : "v"(zq), "v"(pk));
    return x >> 16;
  }

  // 2 x 2 transpose of (register pair (a, b), lane bit K): afterwards a holds [a where bit K = 0 | b of the partner lane where bit K = 1],
  // b holds [a of the partner lane where bit K = 0 | b where bit K = 1].  Bits 5 and 4 are true swaps (v_permlane32_swap / v_permlane16_swap: one
  // instruction per pair); bits 3 ... 0 are two v_cndmask_b32_dpp per pair -- the partner's value comes in through the DPP operand (row_ror:8, row
  // shifts by 4, quad permutes), the lane-bit mask through VCC.  Asm statements: hipcc has no cndmask-with-DPP builtin and lowers the same selection to
  // v_mov_b32_dpp + v_cndmask + copies (twice the instructions of a kernel that is bound by vector issue); `s_nop 1` covers the two wait states
  // between a vector write of an operand and its DPP read, which the hazard recogniser cannot see inside an asm statement.
  template <int K> struct BitMask { static constexpr unsigned long long v = K == 3 ? 0xff00ff00ff00ff00ull : K == 2 ? 0xf0f0f0f0f0f0f0f0ull : K == 1 ? 0xccccccccccccccccull : 0xaaaaaaaaaaaaaaaaull; };
#define PSF_NTT_DPP_PAIR(CTRL_A, CTRL_B)                                                                                                              \
  asm("s_nop 1\n\ts_mov_b64 vcc, %4\n\tv_cndmask_b32_dpp %0, %3, %2, vcc " CTRL_A " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                       \
      "s_mov_b64 vcc, %5\n\tv_cndmask_b32_dpp %1, %2, %3, vcc " CTRL_B " row_mask:0xf bank_mask:0xf bound_ctrl:0"                                      \
      : "=&v"(na), "=&v"(nb) : "v"(a), "v"(b), "s"(~BitMask<K>::v), "s"(BitMask<K>::v) : "vcc")
#define PSF_NTT_DPP_TWO(CTRL_A, CTRL_B)                                                                                                               \
  asm("s_nop 1\n\ts_mov_b64 vcc, %8\n\tv_cndmask_b32_dpp %0, %5, %4, vcc " CTRL_A " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                       \
      "v_cndmask_b32_dpp %2, %7, %6, vcc " CTRL_A " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                                      \
      "s_mov_b64 vcc, %9\n\tv_cndmask_b32_dpp %1, %4, %5, vcc " CTRL_B " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                  \
      "v_cndmask_b32_dpp %3, %6, %7, vcc " CTRL_B " row_mask:0xf bank_mask:0xf bound_ctrl:0"                                                          \
      : "=&v"(na), "=&v"(nb), "=&v"(nc), "=&v"(nd) : "v"(a), "v"(b), "v"(c), "v"(d), "s"(~BitMask<K>::v), "s"(BitMask<K>::v) : "vcc")
  template <int K> static __device__ __forceinline__ void swap(U& a, U& b) {
    if constexpr (K == 5) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r.x; b = r.y; }
    else if constexpr (K == 4) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r.x; b = r.y; }
    else {
      U na, nb;
      if constexpr (K == 3) PSF_NTT_DPP_PAIR("row_ror:8", "row_ror:8");
      else if constexpr (K == 2) PSF_NTT_DPP_PAIR("row_shr:4", "row_shl:4");
      else if constexpr (K == 1) PSF_NTT_DPP_PAIR("quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]");
      else PSF_NTT_DPP_PAIR("quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]");
      a = na; b = nb;
    }
  }
  template <int K> static __device__ __forceinline__ void swap2(U& a, U& b, U& c, U& d) {       // two pairs behind one pair of mask moves
    U na, nb, nc, nd;
    if constexpr (K == 3) PSF_NTT_DPP_TWO("row_ror:8", "row_ror:8");
    else if constexpr (K == 2) PSF_NTT_DPP_TWO("row_shr:4", "row_shl:4");
    else if constexpr (K == 1) PSF_NTT_DPP_TWO("quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]");
    else PSF_NTT_DPP_TWO("quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]");
    a = na; b = nb; c = nc; d = nd;
  }
#undef PSF_NTT_DPP_PAIR
#undef PSF_NTT_DPP_TWO
  template <int K> static __device__ __forceinline__ void swap(I& a, I& b) {
    U ua = (U)a, ub = (U)b;
    swap<K>(ua, ub);
    a = (I)ua; b = (I)ub;
  }
  // lane bit K against register bit J of the C registers of a lane
  template <int K, int C, int J, class V> static __device__ __forceinline__ void exchange(V (&x)[C]) {
    if constexpr (K >= 4 || C < 4) {
#pragma unroll
      for (int r = 0; r < C; ++r)
        if (!((r >> J) & 1)) swap<K>(x[r], x[r | (1 << J)]);
    } else {
      int lo[C / 2];
      int np = 0;
#pragma unroll
      for (int r = 0; r < C; ++r)
        if (!((r >> J) & 1)) lo[np++] = r;
#pragma unroll
      for (int i = 0; i < C / 2; i += 2) {
        U a = (U)x[lo[i]], b = (U)x[lo[i] | (1 << J)], c = (U)x[lo[i + 1]], d = (U)x[lo[i + 1] | (1 << J)];
        swap2<K>(a, b, c, d);
        x[lo[i]] = (V)a; x[lo[i] | (1 << J)] = (V)b; x[lo[i + 1]] = (V)c; x[lo[i + 1] | (1 << J)] = (V)d;
      }
    }
  }
};

// what the host hands every kernel: the modulus in both forms and the table of zetas (forward [2^L] | inverse [2^L], policy form) in global memory
struct NttDev {
  uint32_t q;
  int32_t qinv16;          // q^-1 mod 2^16, signed
  uint32_t nqinv32;        // -q^-1 mod 2^32
  uint32_t fin;            // 2^-L R^(e+1) mod q (centred for the 16-bit form), e = R^-1 factors of a product
  uint32_t fin_fa;         // the same for k_ring_fa (one more uniform reduction)
  uint32_t r2;             // R^2 mod q (input reduction)
  const uint32_t* zetas;
};

template <int QB> struct PolicyOf { using M = Mod16<DevWave, QB>; template <int LOGN, int LD> using BD = Bounds16<QB, LOGN, LD>; };
template <> struct PolicyOf<12> { using M = Mod16D<DevWave>; template <int LOGN, int LD> using BD = Bounds16<12, LOGN, LD>; };
template <> struct PolicyOf<0> { using M = Mod32<DevWave>; template <int LOGN, int LD> using BD = NoBounds; };

template <int QB> __device__ __forceinline__ typename PolicyOf<QB>::M make_policy(const NttDev& p, int levels) {
  typename PolicyOf<QB>::M md;
  if constexpr (QB != 0) { md.q = (int)p.q; md.nq = -(int)p.q; md.qinv = p.qinv16; }
  else { md.q = p.q; md.nqinv = p.nqinv32; }
  if constexpr (QB == 12) md.zoff = 2 << levels;
  return md;
}

// ---- input reduction (the butterflies never divide; a 64-bit operand outside 32 bits pays the one division here) -----------------------------------
template <int QB, class M> __device__ __forceinline__ typename M::V reduce_i64(int64_t x, const M& md, const NttDev& p) {
  if constexpr (QB != 0) {
    if (__all(x >= -(1ll << 30) && x < (1ll << 30))) return md.mul((int)p.r2, md.mont((int)x));   // x R^-1, then times R^2 R^-1 (|x| + 2^15 q < 2^31)
    const int64_t v = x % (int64_t)p.q;
    return (int)v;                                                                           // (-q, q)
  } else {
    if (__all(x >= 0 && x < (int64_t)p.q)) return (uint32_t)x;
    const int64_t v = x % (int64_t)p.q;
    return (uint32_t)(v < 0 ? v + (int64_t)p.q : v);
  }
}
template <int QB, class M> __device__ __forceinline__ typename M::V reduce_u64(uint64_t x, const M& md, const NttDev& p) {
  if constexpr (QB != 0) {
    if (__all(x < (1ull << 30))) return md.mul((int)p.r2, md.mont((int)x));
    return (int)(x % p.q);
  } else {
    if (__all(x < (uint64_t)p.q)) return (uint32_t)x;
    return (uint32_t)(x % p.q);
  }
}

// the wave's number inside its workgroup as a SCALAR: row addresses are then computed once per wave, in SGPRs
static __device__ __forceinline__ unsigned wave_in_block() { return (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <int LOGN, int LD, int QB> struct Kern {
  using M = typename PolicyOf<QB>::M;
  using BD = typename PolicyOf<QB>::template BD<LOGN, LD>;
  using K = Core<DevWave, M, BD, LOGN, LD>;
  using V = typename M::V;
  static constexpr int C = K::C, N = 1 << LOGN, L = LOGN - LD, ZN = (QB == 12 ? 4 : 2) << L;     // forward | inverse (| forward pairs of the dot-product form)
  static_assert(QB != 12 || (BD::r.xf < 32768 && BD::r.nrf == 0), "dot-product form: 16-bit operands in every forward butterfly");
  static constexpr int E = 1 + 2 * BD::r.nrf + BD::r.nri;            // powers of R^-1 a product carries before the final scale

  static __device__ __forceinline__ void load_tables(uint32_t* zt, const NttDev& p) {
    for (int i = threadIdx.x; i < ZN; i += blockDim.x) zt[i] = p.zetas[i];
    __syncthreads();
  }
  // IO = 16: T16 rows of n 16-bit values; IO = 64: rows of n 64-bit values
  template <int IO, bool SIGNED> static __device__ __forceinline__ void load(V (&x)[C], const void* base, size_t row, int lane, const M& md, const NttDev& p) {
#pragma unroll
    for (int r = 0; r < C; ++r) {
      const size_t i = row * N + (size_t)r * 64 + lane;
      if constexpr (IO == 16) {
        if constexpr (SIGNED) x[r] = (V)(int)reinterpret_cast<const int16_t*>(base)[i];
        else x[r] = (V)reinterpret_cast<const uint16_t*>(base)[i];
        if constexpr (QB == 0) { if constexpr (SIGNED) x[r] = DevWave::cadd(x[r], p.q); }
      } else {
        if constexpr (SIGNED) x[r] = reduce_i64<QB>(reinterpret_cast<const int64_t*>(base)[i], md, p);
        else x[r] = reduce_u64<QB>(reinterpret_cast<const uint64_t*>(base)[i], md, p);
      }
    }
  }
  template <int IO> static __device__ __forceinline__ void store(const V (&x)[C], void* base, size_t row, int lane) {
#pragma unroll
    for (int r = 0; r < C; ++r) {
      const size_t i = row * N + (size_t)r * 64 + lane;
      if constexpr (IO == 16) reinterpret_cast<uint16_t*>(base)[i] = (uint16_t)x[r];
      else reinterpret_cast<uint64_t*>(base)[i] = (uint64_t)(uint32_t)x[r];
    }
  }
};

// grid: any; every wave takes the products wave, wave + waves, ...
template <int LOGN, int LD, int QB, int IO>
__global__ __launch_bounds__(256) void k_ntt_polymul(NttDev p, const void* __restrict__ A, const void* __restrict__ B, void* __restrict__ out, size_t count) {
  using KN = Kern<LOGN, LD, QB>;
  using V = typename KN::V;
  __shared__ __attribute__((aligned(16))) uint32_t zt[KN::ZN];
  KN::load_tables(zt, p);
  const auto md = make_policy<QB>(p, KN::L);
  const int lane = DevWave::lane();
  const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6);
  const uint32_t* zf = zt;
  const uint32_t* zi = zt + (1 << KN::L);
  for (size_t pr = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block(); pr < count; pr += waves) {
    V a[KN::C], b[KN::C], c[KN::C];
    KN::template load<IO, false>(a, A, pr, lane, md, p);
    KN::template load<IO, true>(b, B, pr, lane, md, p);
    KN::K::forward(a, md, zf, lane);
    KN::K::forward(b, md, zf, lane);
    KN::K::leafmul(c, a, b, md, zf, lane);
    KN::K::inverse(c, md, zi, lane);
    KN::K::finish(c, md, (V)p.fin);
    KN::template store<IO>(c, out, pr, lane);
  }
}

// hat[(row*C + r)*64 + lane]: the register image of the leaf residues (times R^-nrf in the 16-bit form)
template <int LOGN, int LD, int QB, int IO, bool SIGNED>
__global__ __launch_bounds__(256) void k_ntt_forward(NttDev p, const void* __restrict__ A, uint32_t* __restrict__ hat, size_t count) {
  using KN = Kern<LOGN, LD, QB>;
  using V = typename KN::V;
  __shared__ __attribute__((aligned(16))) uint32_t zt[KN::ZN];
  KN::load_tables(zt, p);
  const auto md = make_policy<QB>(p, KN::L);
  const int lane = DevWave::lane();
  const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6);
  for (size_t pr = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block(); pr < count; pr += waves) {
    V a[KN::C];
    KN::template load<IO, SIGNED>(a, A, pr, lane, md, p);
    KN::K::forward(a, md, zt, lane);
#pragma unroll
    for (int r = 0; r < KN::C; ++r) hat[(pr * KN::C + r) * 64 + lane] = (uint32_t)a[r];
  }
}

// out = a * b with a given by its image; hat_stride = 0: one image for every product
template <int LOGN, int LD, int QB, int IO>
__global__ __launch_bounds__(256) void k_ntt_mul_hat(NttDev p, const uint32_t* __restrict__ hat, size_t hat_stride, const void* __restrict__ B, void* __restrict__ out, size_t count) {
  using KN = Kern<LOGN, LD, QB>;
  using V = typename KN::V;
  __shared__ __attribute__((aligned(16))) uint32_t zt[KN::ZN];
  KN::load_tables(zt, p);
  const auto md = make_policy<QB>(p, KN::L);
  const int lane = DevWave::lane();
  const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6);
  const uint32_t* zf = zt;
  const uint32_t* zi = zt + (1 << KN::L);
  for (size_t pr = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block(); pr < count; pr += waves) {
    V a[KN::C], b[KN::C], c[KN::C];
#pragma unroll
    for (int r = 0; r < KN::C; ++r) a[r] = (V)hat[pr * hat_stride + (size_t)r * 64 + lane];
    KN::template load<IO, true>(b, B, pr, lane, md, p);
    KN::K::forward(b, md, zf, lane);
    KN::K::leafmul(c, a, b, md, zf, lane);
    KN::K::inverse(c, md, zi, lane);
    KN::K::finish(c, md, (V)p.fin);
    KN::template store<IO>(c, out, pr, lane);
  }
}

// PSFGPVRing::f_a (gpv_ring.rs:243-247): u_b = sum_{j < K} a_j * sigma_{b,j} mod (X^n + 1, q).  sigma: B rows of K*n int64 (the preimages as
// psfring_samp_p returns them), hat: the K images of a (in LDS for the whole launch), u: B rows of n uint64.  One wave per preimage.
template <int LOGN, int LD, int QB>
__global__ __launch_bounds__(256) void k_ring_fa(NttDev p, const uint32_t* __restrict__ hat, uint32_t K, const int64_t* __restrict__ sigma, uint64_t* __restrict__ u, size_t count) {
  using KN = Kern<LOGN, LD, QB>;
  using V = typename KN::V;
  extern __shared__ __attribute__((aligned(16))) uint32_t fa_smem[];   // zetas [ZN] | images [K][C][64]
  uint32_t* zt = fa_smem;
  uint32_t* ah = fa_smem + KN::ZN;
  for (uint32_t i = threadIdx.x; i < K * (uint32_t)KN::N; i += blockDim.x) ah[i] = hat[i];
  KN::load_tables(zt, p);
  const auto md = make_policy<QB>(p, KN::L);
  const int lane = DevWave::lane();
  const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6);
  const uint32_t* zf = zt;
  const uint32_t* zi = zt + (1 << KN::L);
  for (size_t pr = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block(); pr < count; pr += waves) {
    V acc[KN::C];
#pragma unroll
    for (int r = 0; r < KN::C; ++r) acc[r] = 0;
    for (uint32_t j = 0; j < K; ++j) {
      V a[KN::C], b[KN::C], c[KN::C];
      KN::template load<64, true>(b, sigma, pr * K + j, lane, md, p);
#pragma unroll
      for (int r = 0; r < KN::C; ++r) a[r] = (V)ah[(j * KN::C + r) * 64 + lane];
      KN::K::forward(b, md, zf, lane);
      KN::K::leafmul(c, a, b, md, zf, lane);
#pragma unroll
      for (int r = 0; r < KN::C; ++r) acc[r] = md.add(acc[r], c[r]);
    }
    if constexpr (QB != 0) KN::K::reduce_all(acc, md);                // K summands of at most xc each: back under xc (one more R^-1, in fin_fa)
    KN::K::inverse(acc, md, zi, lane);
    KN::K::finish(acc, md, (V)(QB != 0 ? p.fin_fa : p.fin));
    KN::template store<64>(acc, u, pr, lane);
  }
}

}  // namespace ntt
}  // namespace psf
