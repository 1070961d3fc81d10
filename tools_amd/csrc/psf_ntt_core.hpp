// psf_ntt_core.hpp -- the wave-level negacyclic NTT of Z_q[X]/(X^n + 1) behind the R_q products of PSFGPVRing
// (PolynomialRingZq multiplication: gadget_ring.rs:78, gpv_ring.rs:243-247; moduli of common_moduli.rs:41-48).
//
// ONE transform = ONE wave, C = n/64 coefficients per lane, no LDS traffic for the data and no barrier:
//   * X^n + 1 splits into 2^L factors X^d - gamma (L = LOGN - LD levels, leaf degree d = 2^LD; q = 3329, n = 256: L = 7, d = 2 as in ML-KEM).
//     Level p (p = LOGN-1 ... LD) pairs the coefficients whose indices differ in bit p (Cooley-Tukey, zetas in bit-reversed order:
//     block blk = i >> (p+1) uses zetas[2^l + blk], l = LOGN-1-p -- the table of make_ntt_plan).
//   * Every butterfly is IN REGISTERS.  A wave starts with index bits 6..LOGN-1 in the register number and bits 0..5 in the lane number
//     (coefficient i = r*64 + lane: every global access is one contiguous row).  Before level p <= 5 the lane bit p is exchanged with the
//     register bit that holds the oldest finished index bit: a 2 x 2 transpose of (register pair, lane bit), which is ONE instruction per
//     register for every lane bit -- v_permlane32_swap (bit 5), v_permlane16_swap (bit 4), a bank-masked DPP row rotate / shift (bits 3, 2),
//     a DPP quad permute under v_cndmask (bits 1, 0).  Register bit j is used by the levels p = j + 6 (mod RB), so after the exchange at
//     level p register bit j holds index bit p + ((j - jof(p)) mod RB) and lane bit k >= p holds index bit k + RB; after the last exchange
//     (level 0) lane L holds the C consecutive coefficients L*C ... L*C + C - 1: the leaves of the pointwise product lie inside a lane.
//   * Arithmetic: no division anywhere.  q < 2^14 (3329, 7681, 12289 ...): signed Montgomery with R = 2^16 on 24-bit multiplies
//     (v_mul_i32_i24 / v_mad_i32_i24 are full rate), values kept UNREDUCED between levels; a compile-time bound analysis (Bounds16) places the
//     few uniform reductions x -> x R^-1 that keep every product below 2^31 and folds their powers of R into the final scale.  q < 2^31:
//     unsigned Montgomery with R = 2^32, canonical values.
// The code is written over a wave back end W (lane id, the exchange of a lane bit, a table read): the device back end is at the end of
// psf_ntt_kernels.hpp; tests/ntt_model/ instantiates the same templates over a 64-lane array on the CPU, so the index plumbing and the
// bound analysis are checked without a GPU (tests/test_ntt_model.py).
#pragma once
#include <stdint.h>
#include <utility>

#if defined(__HIPCC__)
#define PSF_NTT_FN __device__ __forceinline__
#else
#define PSF_NTT_FN inline
#endif

namespace psf {
namespace ntt {

constexpr int cmod(int a, int m) { return ((a % m) + m) % m; }

// ---- the compile-time exchange schedule ---------------------------------------------------------------------------------------------------------
template <int LOGN> struct Sched {
  static_assert(LOGN >= 7 && LOGN <= 10, "one wave holds 128 ... 1024 coefficients");
  static constexpr int RB = LOGN - 6, C = 1 << RB;
  static constexpr int jof(int p) { return cmod(p - 6, RB); }                     // register bit of level p
  // contribution of the register number to the zeta block index at level p <= 5 (register bit j != jof(p) holds index bit p + ((j - jof(p)) mod RB))
  static constexpr int regpart(int r, int p) {
    int v = 0;
    for (int j = 0; j < RB; ++j)
      if (j != jof(p) && ((r >> j) & 1)) v |= 1 << (cmod(j - jof(p), RB) - 1);
    return v;
  }
  // after the exchange at level 0: register r of lane L holds coefficient L*C + nat(r)
  static constexpr int nat(int r) {
    int c = 0;
    for (int j = 0; j < RB; ++j)
      if ((r >> j) & 1) c |= 1 << cmod(j - jof(0), RB);
    return c;
  }
  static constexpr int reg_of_nat(int c) {
    for (int r = 0; r < C; ++r)
      if (nat(r) == c) return r;
    return -1;
  }
};

// ---- bound analysis of the unreduced 16-bit Montgomery form --------------------------------------------------------------------------------------
// QB: q < 2^QB <= 2^14.  Inputs |x| <= 2^QB.  mont(t) = (t - m q) / 2^16 with |m| <= 2^15, so |mont(t)| <= |t| / 2^16 + q/2, for |t| <= LIM.
struct BoundsResult {
  bool fwd_red[11] = {}, inv_red[11] = {};
  bool leaf_red = false, fin_red = false, ok = true;
  int nrf = 0, nri = 0;
  long long xf = 0, xc = 0, xi = 0;
};
struct NoBounds { static constexpr BoundsResult r{}; };           // canonical 32-bit form: nothing to place
template <int QB, int LOGN, int LD> struct Bounds16 {
  static_assert(QB <= 14, "signed 16-bit Montgomery form: q < 2^14");
  // a reduction computes t - m q in 32 bits before the shift: |t| + 2^15 q must stay below 2^31
  static constexpr long long QM = 1ll << QB, ZM = QM / 2, LIM = (1ll << 31) - 1 - (QM << 15), I24 = (1ll << 23) - 1;
  static constexpr long long rb(long long t) { return t / 65536 + ZM + 1; }
  using R = BoundsResult;
  static constexpr long long leaf_t(long long x, long long& u_t) {     // largest |t| of the leaf product, u_t = largest |t| of its inner reduction
    if (LD == 0) { u_t = 0; return x * x; }
    if (LD == 1) { u_t = x * x; const long long t0 = x * x + rb(u_t) * ZM, t1 = 2 * x * x; return t0 > t1 ? t0 : t1; }
    u_t = 3 * x * x;
    return 4 * x * x + rb(u_t) * ZM;
  }
  static constexpr R make() {
    R o;
    long long x = QM;
    for (int p = LOGN - 1; p >= LD; --p) {
      if (x * ZM > LIM || x > I24) { o.fwd_red[p] = true; ++o.nrf; x = rb(x); }
      x = x + rb(x * ZM);
    }
    long long ut = 0;
    if (x > I24 || leaf_t(x, ut) > LIM || ut > LIM) { o.leaf_red = true; ++o.nrf; x = rb(x); }
    o.xf = x;
    const long long lt = leaf_t(x, ut);
    if (lt > LIM || ut > LIM || x > I24) o.ok = false;
    x = rb(lt);
    o.xc = x;
    for (int p = LD; p < LOGN; ++p) {
      if (2 * x * ZM > LIM || 2 * x > I24) { o.inv_red[p] = true; ++o.nri; x = rb(x); }
      if (2 * x * ZM > LIM) o.ok = false;
      x = 2 * x;
    }
    if (x >= 65536) { o.fin_red = true; ++o.nri; x = rb(x); }
    if (x >= 65536) o.ok = false;
    o.xi = x;
    return o;
  }
  static constexpr R r = make();
  static_assert(make().ok, "bound analysis failed");
};

// ---- arithmetic policies --------------------------------------------------------------------------------------------------------------------------
// Both expose: V (lane value), mul(z, x) = z x R^-1, add, sub, mont(x) = x R^-1 (uniform reduction), canon(x) in [0, q).
template <class W, int QB_> struct Mod16 {
  static constexpr int QB = QB_;
  static constexpr bool lazy = true;
  using V = typename W::I;
  int q, nq, qinv;                        // q, -q, q^-1 mod 2^16 (as a signed 16-bit value)
  PSF_NTT_FN V mont(V t) const { return W::mont16(t, qinv, nq); }              // (t - m q) >> 16, m = the signed low half of t q^-1
  PSF_NTT_FN V mul(V z, V x) const { return mont(W::mul24(z, x)); }
  PSF_NTT_FN V add(V a, V b) const { return a + b; }
  PSF_NTT_FN V sub(V a, V b) const { return a - b; }
  PSF_NTT_FN V neg(V a) const { return W::izero() - a; }
  PSF_NTT_FN V canon(V x) const { return W::umin(x, x + q); }                  // x in (-q, q): as unsigned numbers the smaller of x and x + q
  // sums of products reduced once (leaf products): t = a b (+ c d ...)
  PSF_NTT_FN V prod(V a, V b) const { return W::mul24(a, b); }
  PSF_NTT_FN V prod_add(V a, V b, V t) const { return W::mad24(a, b, t); }
  // the zeta of a forward butterfly and its product with the upper element
  using FZ = V;
  PSF_NTT_FN FZ fz(const typename W::Tab& zf, typename W::I idx, int off) const { return W::template tab<V>(zf, idx, off); }
  PSF_NTT_FN FZ fz_const(const typename W::Tab& zf, int idx) const { return W::template tab_const<V>(zf, idx); }
  PSF_NTT_FN V mulfz(FZ z, V x) const { return mul(z, x); }
};
// q < 2^12 with seven levels: every operand of a forward butterfly stays below 2^15 (Bounds16: xf = 20 839), so the product and its reduction are
// ONE two-element dot product:  z x - m q = (x | m) . (z | -q)  with m = the low half of x (z q^-1), written into the upper half of x's register.
// Two instructions per Montgomery product instead of three.  The table holds (z | -q) and z q^-1 mod 2^16 per zeta behind the plain tables.
template <class W> struct Mod16D : Mod16<W, 12> {
  using V = typename W::I;
  int zoff;                               // words in front of the pairs: forward [2^L] | inverse [2^L]
  struct FZ { V pk, zq; };
  PSF_NTT_FN FZ fz(const typename W::Tab& zf, typename W::I idx, int off) const {
    FZ z;
    W::template tab_pair<V>(zf, zoff, idx, off, z.pk, z.zq);
    return z;
  }
  PSF_NTT_FN FZ fz_const(const typename W::Tab& zf, int idx) const {
    FZ z;
    W::template tab_pair_const<V>(zf, zoff, idx, z.pk, z.zq);
    return z;
  }
  PSF_NTT_FN V mulfz(const FZ& z, V x) const { return W::dot2mont(x, z.zq, z.pk); }
};
template <class W> struct Mod32 {
  static constexpr bool lazy = false;
  using V = typename W::U;
  uint32_t q, nqinv;                      // q, -q^-1 mod 2^32
  PSF_NTT_FN V mul(V a, V b) const {      // a, b < q < 2^31: (a b + m q) / 2^32 < 2 q
    const V lo = W::mullo_u(a, b), hi = W::mulhi_u(a, b);
    const V m = W::mullo_u(lo, nqinv);
    const V r = hi + W::mulhi_u(m, q) + W::nonzero(lo);                        // lo + lo(m q) = 2^32 exactly unless lo = 0
    return W::csub(r, q);
  }
  PSF_NTT_FN V mont(V x) const { return x; }
  PSF_NTT_FN V add(V a, V b) const { return W::csub(a + b, q); }
  PSF_NTT_FN V sub(V a, V b) const { return W::cadd(a - b, q); }               // a - b wraps below zero: add q back
  PSF_NTT_FN V neg(V a) const { return W::cadd(W::uzero() - a, q); }
  PSF_NTT_FN V canon(V x) const { return x; }
  using FZ = V;
  PSF_NTT_FN FZ fz(const typename W::Tab& zf, typename W::I idx, int off) const { return W::template tab<V>(zf, idx, off); }
  PSF_NTT_FN FZ fz_const(const typename W::Tab& zf, int idx) const { return W::template tab_const<V>(zf, idx); }
  PSF_NTT_FN V mulfz(FZ z, V x) const { return mul(z, x); }
};

// ---- the transforms -------------------------------------------------------------------------------------------------------------------------------
template <class W, class M, class BD, int LOGN, int LD> struct Core {
  using S = Sched<LOGN>;
  using V = typename M::V;
  using I = typename W::I;
  static constexpr int C = S::C, RB = S::RB, L = LOGN - LD;
  static_assert(LD >= 0 && LD <= 2 && LD <= RB, "leaf degree 1, 2 or 4, inside a lane");

  template <int P> static PSF_NTT_FN void exchange(V (&x)[C]) { W::template exchange<P, C, S::jof(P)>(x); }
  static PSF_NTT_FN void reduce_all(V (&x)[C], const M& md) {
#pragma unroll
    for (int r = 0; r < C; ++r) x[r] = md.mont(x[r]);
  }
  template <int P> static PSF_NTT_FN I zeta_lane(I lane) {            // lane part of the zeta index at level P <= 5
    return W::shl(W::srl(lane, P), RB - 1) + (1 << (LOGN - 1 - P));
  }
  template <int P> static PSF_NTT_FN V zeta_at(const typename W::Tab& zt, I zl, int r) {
    if constexpr (P >= 6) return W::template tab_const<V>(zt, (1 << (LOGN - 1 - P)) + (r >> (P - 5)));
    else return W::template tab<V>(zt, zl, S::regpart(r, P));
  }

  template <int P> static PSF_NTT_FN void fwd_level(V (&x)[C], const M& md, const typename W::Tab& zf, I lane) {
    constexpr int J = S::jof(P);
    if constexpr (M::lazy) { if constexpr (BD::r.fwd_red[P]) reduce_all(x, md); }
    if constexpr (P <= 5) exchange<P>(x);
    I zl = lane;
    if constexpr (P <= 5) zl = zeta_lane<P>(lane);
#pragma unroll
    for (int r = 0; r < C; ++r)
      if (!((r >> J) & 1)) {
        typename M::FZ z;
        if constexpr (P >= 6) z = md.fz_const(zf, (1 << (LOGN - 1 - P)) + (r >> (P - 5)));
        else z = md.fz(zf, zl, S::regpart(r, P));
        const V t = md.mulfz(z, x[r | (1 << J)]);
        x[r | (1 << J)] = md.sub(x[r], t);
        x[r] = md.add(x[r], t);
      }
  }
  template <int P> static PSF_NTT_FN void inv_level(V (&x)[C], const M& md, const typename W::Tab& zi, I lane) {
    constexpr int J = S::jof(P);
    if constexpr (M::lazy) { if constexpr (BD::r.inv_red[P]) reduce_all(x, md); }
    I zl = lane;
    if constexpr (P <= 5) zl = zeta_lane<P>(lane);
#pragma unroll
    for (int r = 0; r < C; ++r)
      if (!((r >> J) & 1)) {
        const V z = zeta_at<P>(zi, zl, r);
        const V u = x[r], v = x[r | (1 << J)];
        x[r] = md.add(u, v);
        x[r | (1 << J)] = md.mul(z, md.sub(u, v));
      }
    if constexpr (P <= 5) exchange<P>(x);
  }
  template <int... K> static PSF_NTT_FN void fwd_levels(V (&x)[C], const M& md, const typename W::Tab& zf, I lane, std::integer_sequence<int, K...>) {
    (fwd_level<LOGN - 1 - K>(x, md, zf, lane), ...);
  }
  template <int... K> static PSF_NTT_FN void inv_levels(V (&x)[C], const M& md, const typename W::Tab& zi, I lane, std::integer_sequence<int, K...>) {
    (inv_level<LD + K>(x, md, zi, lane), ...);
  }
  template <int... K> static PSF_NTT_FN void tail_down(V (&x)[C], std::integer_sequence<int, K...>) { (exchange<LD - 1 - K>(x), ...); }
  template <int... K> static PSF_NTT_FN void tail_up(V (&x)[C], std::integer_sequence<int, K...>) { (exchange<K>(x), ...); }

  // x: coefficients i = r*64 + lane, |x| <= 2^QB (lazy form) / canonical  ->  leaf residues, register r of lane l = position l*C + nat(r)
  static PSF_NTT_FN void forward(V (&x)[C], const M& md, const typename W::Tab& zf, I lane) {
    fwd_levels(x, md, zf, lane, std::make_integer_sequence<int, L>());
    tail_down(x, std::make_integer_sequence<int, LD>());
    if constexpr (M::lazy) { if constexpr (BD::r.leaf_red) reduce_all(x, md); }
  }
  // leaf residues (times R^-e) -> coefficients i = r*64 + lane of 2^L x (times R^-e'), not yet scaled
  static PSF_NTT_FN void inverse(V (&x)[C], const M& md, const typename W::Tab& zi, I lane) {
    tail_up(x, std::make_integer_sequence<int, LD>());
    inv_levels(x, md, zi, lane, std::make_integer_sequence<int, L>());
    if constexpr (M::lazy) { if constexpr (BD::r.fin_red) reduce_all(x, md); }
  }
  // scale by fin (= 2^-L R^(e+1) in the policy's form) and bring into [0, q)
  static PSF_NTT_FN void finish(V (&x)[C], const M& md, V fin) {
#pragma unroll
    for (int r = 0; r < C; ++r) x[r] = md.canon(md.mul(fin, x[r]));
  }

  // c = a * b in every leaf ring Z_q[X]/(X^d - gamma); gamma = +-(the zeta of the last level), all times R^-1
  static PSF_NTT_FN void leafmul(V (&c)[C], const V (&a)[C], const V (&b)[C], const M& md, const typename W::Tab& zf, I lane) {
    constexpr int D = 1 << LD, NL = C >> LD;                      // leaves per lane
    if constexpr (LD == 0) {
#pragma unroll
      for (int r = 0; r < C; ++r) c[r] = md.mul(a[r], b[r]);
    } else {
#pragma unroll
      for (int t = 0; t < NL; ++t) {
        // global leaf number g = lane*NL + t; zeta index 2^(L-1) + (g >> 1), sign (-1)^g
        V z;
        if constexpr (NL >= 2) {
          z = W::template tab<V>(zf, W::shl(lane, RB - LD - 1) + (1 << (L - 1)), t >> 1);
          if (t & 1) z = md.neg(z);
        } else {
          const V zp = W::template tab<V>(zf, W::srl(lane, 1) + (1 << (L - 1)), 0);
          z = W::sel_odd(lane, md.neg(zp), zp);
        }
        V av[D], bv[D];
#pragma unroll
        for (int e = 0; e < D; ++e) { av[e] = a[S::reg_of_nat(t * D + e)]; bv[e] = b[S::reg_of_nat(t * D + e)]; }
#pragma unroll
        for (int k = 0; k < D; ++k) {
          V res;
          if constexpr (M::lazy) {
            // hi = sum_{i+j = k+D} a_i b_j (reduced once), then lo + gamma hi in one more reduction
            V t_lo = md.prod(av[0], bv[k]);
#pragma unroll
            for (int i = 1; i <= k; ++i) t_lo = md.prod_add(av[i], bv[k - i], t_lo);
            if (k + 1 < D) {
              V t_hi = md.prod(av[k + 1], bv[D - 1]);
#pragma unroll
              for (int i = k + 2; i < D; ++i) t_hi = md.prod_add(av[i], bv[D + k - i], t_hi);
              t_lo = md.prod_add(z, md.mont(t_hi), t_lo);
            }
            res = md.mont(t_lo);
          } else {
            res = md.mul(av[0], bv[k]);
#pragma unroll
            for (int i = 1; i <= k; ++i) res = md.add(res, md.mul(av[i], bv[k - i]));
            if (k + 1 < D) {
              V hi = md.mul(av[k + 1], bv[D - 1]);
#pragma unroll
              for (int i = k + 2; i < D; ++i) hi = md.add(hi, md.mul(av[i], bv[D + k - i]));
              res = md.add(res, md.mul(z, hi));                   // z in Montgomery form: gamma hi, still times the one R^-1 of the products
            }
          }
          c[S::reg_of_nat(t * D + k)] = res;
        }
      }
    }
  }
};

}  // namespace ntt
}  // namespace psf
