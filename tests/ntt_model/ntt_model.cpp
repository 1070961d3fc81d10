// CPU model of the wave-level NTT: the templates of tools_amd/csrc/psf_ntt_core.hpp instantiated over a 64-lane array instead of a wavefront.
// Test infrastructure (built and run by tests/test_ntt_model.py): it checks the exchange schedule, the zeta indexing, the leaf products and the
// bound analysis of the unreduced 16-bit form (every 24-bit multiply asserts its operand ranges) against a schoolbook product -- without a GPU.
// The exchange of a lane bit is modelled by its definition (2 x 2 transpose of register pair and lane bit); that the device instructions
// implement this definition is what tests/test_gpu_ntt.py checks on the GPU.
#include <array>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "../../tools_amd/csrc/psf_host.hpp"
#include "../../tools_amd/csrc/psf_ntt_core.hpp"

using namespace psf;
using namespace psf::ntt;

template <class T> struct HV {
  std::array<T, 64> v;
  HV() { v.fill(0); }
  HV(T s) { v.fill(s); }
  template <class O> explicit HV(const HV<O>& o) { for (int l = 0; l < 64; ++l) v[l] = (T)o.v[l]; }
};
#define HV_OP(op)                                                                                              \
  template <class T> HV<T> operator op(const HV<T>& a, const HV<T>& b) { HV<T> r; for (int l = 0; l < 64; ++l) r.v[l] = (T)(a.v[l] op b.v[l]); return r; } \
  template <class T> HV<T> operator op(const HV<T>& a, T b) { HV<T> r; for (int l = 0; l < 64; ++l) r.v[l] = (T)(a.v[l] op b); return r; }
HV_OP(+) HV_OP(-) HV_OP(&)
static HV<int32_t> operator&(const HV<int32_t>& a, int b) { return a & HV<int32_t>(b); }
static HV<uint32_t> operator+(const HV<uint32_t>& a, int b) { return a + HV<uint32_t>((uint32_t)b); }

static long long g_max_prod = 0;
struct HostWave {
  using I = HV<int32_t>;
  using U = HV<uint32_t>;
  using Tab = const uint32_t*;
  static I lane() { I r; for (int l = 0; l < 64; ++l) r.v[l] = l; return r; }
  static I izero() { return I(0); }
  static U uzero() { return U(0u); }
  static I sra(I x, int s) { for (auto& e : x.v) e >>= s; return x; }
  static I srl(I x, int s) { for (auto& e : x.v) e = (int32_t)((uint32_t)e >> s); return x; }
  static I shl(I x, int s) { for (auto& e : x.v) e = (int32_t)((uint32_t)e << s); return x; }
  static I mont16(I t, int qinv, int nq) {
    for (auto& e : t.v) {
      const int16_t m = (int16_t)(uint16_t)((uint32_t)e * (uint32_t)qinv);
      const long long r = (long long)e + (long long)m * nq;
      if (r & 0xffff) { std::fprintf(stderr, "Montgomery step not exact\n"); std::abort(); }
      if (r < -(1ll << 31) || r >= (1ll << 31)) { std::fprintf(stderr, "Montgomery step overflows 32 bits: t = %d\n", e); std::abort(); }
      e = (int32_t)(r >> 16);
    }
    return t;
  }
  static int32_t chk24(long long a, long long b, long long c) {
    if (a < -(1ll << 23) || a >= (1ll << 23) || b < -(1ll << 23) || b >= (1ll << 23)) { std::fprintf(stderr, "24-bit operand out of range: %lld * %lld\n", a, b); std::abort(); }
    const long long t = a * b + c;
    if (t < -(1ll << 31) || t >= (1ll << 31)) { std::fprintf(stderr, "32-bit overflow: %lld * %lld + %lld\n", a, b, c); std::abort(); }
    if (std::llabs(t) > g_max_prod) g_max_prod = std::llabs(t);
    return (int32_t)t;
  }
  static I mul24(I a, I b) { I r; for (int l = 0; l < 64; ++l) r.v[l] = chk24(a.v[l], b.v[l], 0); return r; }
  static I mul24(int a, I b) { return mul24(I(a), b); }
  static I mad24(I a, I b, I c) { I r; for (int l = 0; l < 64; ++l) r.v[l] = chk24(a.v[l], b.v[l], c.v[l]); return r; }
  static I mad24(I a, int b, I c) { return mad24(a, I(b), c); }
  static U mullo_u(U a, U b) { for (int l = 0; l < 64; ++l) a.v[l] *= b.v[l]; return a; }
  static U mullo_u(U a, uint32_t b) { return mullo_u(a, U(b)); }
  static U mulhi_u(U a, U b) { for (int l = 0; l < 64; ++l) a.v[l] = (uint32_t)(((uint64_t)a.v[l] * b.v[l]) >> 32); return a; }
  static U mulhi_u(U a, uint32_t b) { return mulhi_u(a, U(b)); }
  static U nonzero(U x) { for (auto& e : x.v) e = e != 0; return x; }
  static U csub(U r, uint32_t q) { for (auto& e : r.v) e = e >= q ? e - q : e; return r; }
  static U cadd(U x, uint32_t q) { for (auto& e : x.v) e = e + (q & (uint32_t)((int32_t)e >> 31)); return x; }
  template <class V> static V tab(Tab t, I idx, int off) { V r; for (int l = 0; l < 64; ++l) r.v[l] = (decltype(r.v[0]))t[idx.v[l] + off]; return r; }
  template <class V> static V tab_const(Tab t, int idx) { V r; for (int l = 0; l < 64; ++l) r.v[l] = (decltype(r.v[0]))t[idx]; return r; }
  static I umin(I a, I b) { I r; for (int l = 0; l < 64; ++l) r.v[l] = (uint32_t)a.v[l] < (uint32_t)b.v[l] ? a.v[l] : b.v[l]; return r; }
  template <class V> static void tab_pair(Tab t, int zoff, I idx, int off, V& pk, V& zq) {
    for (int l = 0; l < 64; ++l) { pk.v[l] = (int32_t)t[zoff + 2 * (idx.v[l] + off)]; zq.v[l] = (int32_t)t[zoff + 2 * (idx.v[l] + off) + 1]; }
  }
  template <class V> static void tab_pair_const(Tab t, int zoff, int idx, V& pk, V& zq) { tab_pair<V>(t, zoff, I(0), idx, pk, zq); }
  static I dot2mont(I x, I zq, I pk) {
    for (int l = 0; l < 64; ++l) {
      const int32_t xv = x.v[l];
      if (xv < -32768 || xv > 32767) { std::fprintf(stderr, "dot-product form: operand %d outside 16 bits\n", xv); std::abort(); }
      const int16_t m = (int16_t)(uint16_t)((uint16_t)xv * (uint16_t)zq.v[l]);
      const long long S = (long long)xv * (int16_t)(pk.v[l] & 0xffff) + (long long)m * (int16_t)((uint32_t)pk.v[l] >> 16);
      if (S & 0xffff) { std::fprintf(stderr, "dot-product form: not exact\n"); std::abort(); }
      if (S < -(1ll << 31) || S >= (1ll << 31)) { std::fprintf(stderr, "dot-product form: overflow\n"); std::abort(); }
      x.v[l] = (int32_t)(S >> 16);
    }
    return x;
  }
  template <class V> static V sel_odd(I lane, V a, V b) { V r; for (int l = 0; l < 64; ++l) r.v[l] = (lane.v[l] & 1) ? a.v[l] : b.v[l]; return r; }
  template <int K, int C, int J, class V> static void exchange(V (&x)[C]) {
    for (int r = 0; r < C; ++r)
      if (!((r >> J) & 1)) swap<K>(x[r], x[r | (1 << J)]);
  }
  template <int K, class V> static void swap(V& a, V& b) {
    V na, nb;
    for (int l = 0; l < 64; ++l) {
      const int pl = l ^ (1 << K);
      const bool hi = (l >> K) & 1;
      na.v[l] = hi ? b.v[pl] : a.v[l];
      nb.v[l] = hi ? b.v[l] : a.v[pl];
    }
    a = na; b = nb;
  }
};

static std::vector<uint64_t> schoolbook(const std::vector<uint64_t>& a, const std::vector<int64_t>& b, uint64_t q) {
  const size_t n = a.size();
  std::vector<uint64_t> out(n);
  for (size_t c = 0; c < n; ++c) {
    i128 acc = 0;
    for (size_t i = 0; i < n; ++i) {
      const size_t j = (c + n - i) % n;
      const i128 t = (i128)(a[i] % q) * (i128)(b[j] % (int64_t)q);
      acc += (i <= c) ? t : -t;
    }
    acc %= (i128)q;
    if (acc < 0) acc += q;
    out[c] = (uint64_t)acc;
  }
  return out;
}

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <int LOGN, int LD, int QB> static int run_case(uint64_t q, bool extreme) {
  using W = HostWave;
  constexpr int N = 1 << LOGN, C = N / 64;
  const NttPlan pl = make_ntt_plan(q, N);
  const NttTables tb = make_ntt_tables(pl);
  if (!pl.ok || !tb.wave || tb.logn != LOGN || tb.ld != LD || tb.qb != QB) { std::printf("plan mismatch q=%llu n=%d\n", (unsigned long long)q, N); return 1; }
  std::vector<uint64_t> a(N);
  std::vector<int64_t> b(N);
  for (int i = 0; i < N; ++i) {
    a[i] = extreme ? q - 1 : rnd() % q;
    b[i] = extreme ? ((i & 1) ? (int64_t)q - 1 : -((int64_t)q - 1)) : (int64_t)(rnd() % (2 * q - 1)) - (int64_t)(q - 1);
  }
  const std::vector<uint64_t> want = schoolbook(a, b, q);
  const uint32_t* zf = tb.zetas.data();
  const uint32_t* zi = zf + (1u << pl.L);
  std::vector<uint64_t> got(N);
  const auto lane = W::lane();
  if constexpr (QB != 0) {
    using M = std::conditional_t<QB == 12, Mod16D<W>, Mod16<W, QB>>;
    using BD = Bounds16<QB, LOGN, LD>;
    using K = Core<W, M, BD, LOGN, LD>;
    M md; md.q = (int)q; md.nq = -(int)q; md.qinv = tb.qinv16;
    if constexpr (QB == 12) md.zoff = 2 << pl.L;
    typename M::V x[C], y[C], c[C];
    for (int r = 0; r < C; ++r)
      for (int l = 0; l < 64; ++l) {
        x[r].v[l] = (int32_t)a[r * 64 + l];                                     // canonical
        y[r].v[l] = (int32_t)(b[r * 64 + l] % (int64_t)q);                      // (-q, q)
      }
    K::forward(x, md, zf, lane);
    K::forward(y, md, zf, lane);
    K::leafmul(c, x, y, md, zf, lane);
    K::inverse(c, md, zi, lane);
    const int e = 1 + 2 * BD::r.nrf + BD::r.nri;
    K::finish(c, md, typename M::V((int32_t)ntt_final_scale(tb, pl, e)));
    for (int r = 0; r < C; ++r)
      for (int l = 0; l < 64; ++l) got[r * 64 + l] = (uint64_t)(int64_t)c[r].v[l];
    std::printf("  bounds: nrf=%d nri=%d leaf_red=%d fin_red=%d xf=%lld xc=%lld max|t|=%lld\n", BD::r.nrf, BD::r.nri, (int)BD::r.leaf_red, (int)BD::r.fin_red, BD::r.xf, BD::r.xc, g_max_prod);
  } else {
    using M = Mod32<W>;
    using K = Core<W, M, NoBounds, LOGN, LD>;
    M md; md.q = (uint32_t)q; md.nqinv = tb.nqinv32;
    typename M::V x[C], y[C], c[C];
    for (int r = 0; r < C; ++r)
      for (int l = 0; l < 64; ++l) {
        x[r].v[l] = (uint32_t)a[r * 64 + l];
        const int64_t v = b[r * 64 + l] % (int64_t)q;
        y[r].v[l] = (uint32_t)(v < 0 ? v + (int64_t)q : v);
      }
    K::forward(x, md, zf, lane);
    K::forward(y, md, zf, lane);
    K::leafmul(c, x, y, md, zf, lane);
    K::inverse(c, md, zi, lane);
    K::finish(c, md, typename M::V(ntt_final_scale(tb, pl, 1)));
    for (int r = 0; r < C; ++r)
      for (int l = 0; l < 64; ++l) got[r * 64 + l] = c[r].v[l];
  }
  int bad = 0;
  for (int i = 0; i < N; ++i) bad += got[i] != want[i];
  std::printf("q=%llu n=%d L=%u d=%u qb=%d %s: %s (%d mismatches)\n", (unsigned long long)q, N, pl.L, pl.d, QB, extreme ? "extreme" : "random", bad ? "FAIL" : "ok", bad);
  return bad != 0;
}

int main() {
  int bad = 0;
  for (int ex = 0; ex < 2; ++ex) {
    bad += run_case<8, 1, 12>(3329, ex);       // ML-KEM: L = 7, d = 2
    bad += run_case<7, 0, 12>(3329, ex);       // complete
    bad += run_case<9, 2, 12>(3329, ex);       // d = 4
    bad += run_case<8, 0, 14>(7681, ex);       // complete, 8 levels
    bad += run_case<10, 0, 14>(12289, ex);     // complete, 10 levels, 16 coefficients per lane
    bad += run_case<9, 0, 14>(12289, ex);
    bad += run_case<7, 0, 12>(257, ex);
    bad += run_case<8, 0, 0>(2013265921u, ex);             // 2^31 - 2^27 + 1
    bad += run_case<9, 0, 0>(1073479681u, ex);
    bad += run_case<8, 1, 0>(22273, ex);                   // v2(q-1) = 8: d = 2 in the 32-bit form
    bad += run_case<8, 2, 0>(20353, ex);                   // v2 = 7: d = 4
    bad += run_case<8, 1, 14>(7937, ex);
    bad += run_case<9, 2, 14>(7937, ex);
    bad += run_case<7, 1, 14>(1153, ex);                   // one register bit, d = 2: the leaf sign is a lane bit
    bad += run_case<8, 2, 14>(1153, ex);                   // two register bits, d = 4: one leaf per lane
    bad += run_case<10, 1, 14>(13313, ex);
  }
  std::printf("NTT_MODEL %s\n", bad ? "FAIL" : "OK");
  return bad != 0;
}
