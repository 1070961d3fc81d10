// psf_rng.hpp -- randomness contract of the library, device side (gfx950).
//
// The reference samples through qfall-math's thread-local RNG and takes no seed (psf.rs:48-80), so the
// contract is this library's own (DESIGN.md "Randomness contract"): every random decision is a pure
// function of (seed, stream tag, global preimage index, coordinate, attempt) through Philox4x32-10, and
// all floating point is IEEE binary64 with explicit fma -- compiled with -ffp-contract=off -- so that any
// evaluation order / lane assignment gives the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace psf {

enum StreamTag : uint32_t {
  TAG_ABAR = 1, TAG_R = 2, TAG_NORMAL = 3, TAG_PERTURB = 4, TAG_GADGET = 5, TAG_SAMPD = 6,
  TAG_TARGET = 7, TAG_GPV = 8, TAG_RING_R = 9, TAG_RING_E = 10, TAG_RING_A = 11, TAG_GPV2 = 12
};

constexpr uint32_t kMaxAttempts = 65536u;

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

__host__ __device__ inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// Philox4x32-10, key = seed, counter = (c0, c1, c2, c3)
__host__ __device__ inline U4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32) instead of a mul_hi / mul_lo pair
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t lo0 = (uint32_t)p0, hi0 = (uint32_t)(p0 >> 32);
    const uint32_t lo1 = (uint32_t)p1, hi1 = (uint32_t)(p1 >> 32);
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

__host__ __device__ inline uint32_t tag_word(uint32_t tag, uint64_t index) {
  return tag | ((uint32_t)(index >> 32) << 8);
}

// exp(y), y <= 0: k = floor(y log2e + 1/2); r = y - k ln2 (hi/lo); degree-13 Taylor by fma; 2^k by exponent add.
__host__ __device__ inline double det_exp(double y) {
  if (!(y > -708.0)) return 0.0;
  if (y > 0.0) y = 0.0;
  const double kf = floor(y * 1.4426950408889634 + 0.5);
  double r = fma(kf, -6.93147180369123816490e-01, y);
  r = fma(kf, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const long long k = (long long)kf;
  const unsigned long long ubits = __builtin_bit_cast(unsigned long long, p) + ((unsigned long long)k << 52);   // k may be negative
  long long bits = (long long)ubits;
  return __builtin_bit_cast(double, bits);
}

// Parameters of D_{Z,s,.} that depend only on s; computed once on the host.
struct SampleZParams {
  double inv_s;     // 1/s
  long long c6;     // ceil(6 s)
  long long f6;     // floor(6 s)
  uint32_t n_int;   // candidates when the centre is an integer: c6 + f6 + 1 (one fewer otherwise)
  uint32_t thr_int, thr_frac;   // 2^sh mod N for N = n_int and N = n_int - 1 (Lemire rejection thresholds)
  uint32_t sh;      // bits per attempt word: 16 (narrow, n_int <= 4096) or 32 (wide)
};

constexpr uint32_t kNarrowMaxN = 4096;

inline SampleZParams make_sample_z_params(double s) {
  SampleZParams p;
  p.inv_s = 1.0 / s;
  p.c6 = (long long)ceil(6.0 * s);
  p.f6 = (long long)floor(6.0 * s);
  p.n_int = (uint32_t)(p.c6 + p.f6 + 1);
  p.sh = p.n_int <= kNarrowMaxN ? 16 : 32;
  const uint64_t two_sh = 1ull << p.sh;
  p.thr_int = (uint32_t)(two_sh % p.n_int);
  p.thr_frac = p.n_int > 1 ? (uint32_t)(two_sh % (p.n_int - 1)) : 0;
  return p;
}

// SampleZ of GPV08 as the reference documents it (CONTRIBUTING.md:35-45): candidates uniform in
// [c - ceil(6s), c + floor(6s)], accepted with probability exp(-pi (x-c)^2 / s^2).
// An attempt consumes a candidate word wa and an acceptance word wb of sh bits each:
//   narrow (sh = 16, at most 4096 candidates): Philox block g (coord, index_lo, g, tag|index_hi) serves attempts 4g + j,
//          j = 0..3, from its word j: wa = high half, wb = low half;
//   wide   (sh = 32): block b serves attempts 2b (words x, y) and 2b + 1 (words z, w).
//   candidate : index = (wa * N) >> sh, attempt void if the low sh bits of the product are below 2^sh mod N (Lemire: exactly uniform);
//   acceptance: U = wb * 2^32 + ext against floor(rho * 2^(sh+32)); ext comes from block (0x80000000 | attempt), drawn only on a tie.
// The value of a sample is its first accepted attempt, whatever lane or order evaluates the attempts.
struct SzRange { long long lo; uint32_t N, thr, sh; };
__host__ __device__ inline SzRange sz_range(double center, const SampleZParams sp) {
  SzRange r;
  const double cc = ceil(center), cf = floor(center);
  const bool integral = cc == cf;                 // N = floor(c) + f6 - (ceil(c) - c6) + 1
  r.lo = (long long)cc - sp.c6;
  r.N = integral ? sp.n_int : sp.n_int - 1;
  r.thr = integral ? sp.thr_int : sp.thr_frac;
  r.sh = sp.sh;
  return r;
}
// the (wa, wb) words of attempts 4g .. 4g+3
__host__ __device__ inline void sz_group_words(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t g, uint32_t sh,
                                               uint32_t wa[4], uint32_t wb[4]) {
  if (sh == 16) {
    const U4 w = philox(seed, coord, idx_lo, g, tw);
    wa[0] = w.x >> 16; wb[0] = w.x & 0xffffu;
    wa[1] = w.y >> 16; wb[1] = w.y & 0xffffu;
    wa[2] = w.z >> 16; wb[2] = w.z & 0xffffu;
    wa[3] = w.w >> 16; wb[3] = w.w & 0xffffu;
  } else {
    const U4 u = philox(seed, coord, idx_lo, 2 * g, tw), v = philox(seed, coord, idx_lo, 2 * g + 1, tw);
    wa[0] = u.x; wb[0] = u.y; wa[1] = u.z; wb[1] = u.w;
    wa[2] = v.x; wb[2] = v.y; wa[3] = v.z; wb[3] = v.w;
  }
}
// ... of the single attempt t
__host__ __device__ inline void sz_attempt_words(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t t, uint32_t sh,
                                                 uint32_t* wa, uint32_t* wb) {
  if (sh == 16) {
    const U4 w = philox(seed, coord, idx_lo, t >> 2, tw);
    const uint32_t word = (t & 2) ? ((t & 1) ? w.w : w.z) : ((t & 1) ? w.y : w.x);
    *wa = word >> 16; *wb = word & 0xffffu;
  } else {
    const U4 w = philox(seed, coord, idx_lo, t >> 1, tw);
    *wa = (t & 1) ? w.z : w.x; *wb = (t & 1) ? w.w : w.y;
  }
}
// exact acceptance decision for candidate x of attempt t (acceptance word wb): wb vs floor(rho 2^sh), tie -> side block
__host__ __device__ inline bool sz_decide(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t t, long long x, uint32_t wb,
                                          double center, double inv_s, uint32_t sh) {
  const double a = ((double)x - center) * inv_s;
  const double rs = det_exp(-3.14159265358979323846 * (a * a)) * (sh == 16 ? 65536.0 : 4294967296.0);
  const double rf = floor(rs);
  const uint64_t ru = (uint64_t)rf;
  if ((uint64_t)wb < ru) return true;
  if ((uint64_t)wb > ru) return false;
  const U4 w2 = philox(seed, coord, idx_lo, 0x80000000u | t, tw);
  return (double)w2.x < floor((rs - rf) * 4294967296.0);
}
__host__ __device__ inline bool sz_attempt(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t t, uint32_t wa, uint32_t wb,
                                           const SzRange rg, double center, double inv_s, long long* x_out) {
  const uint64_t prod = (uint64_t)wa * rg.N;
  if ((uint32_t)(prod & ((1ull << rg.sh) - 1)) < rg.thr) return false;
  const long long x = rg.lo + (long long)(prod >> rg.sh);
  *x_out = x;
  return sz_decide(seed, coord, idx_lo, tw, t, x, wb, center, inv_s, rg.sh);
}
#if defined(__HIPCC__)
// Conservative single-precision screen for one attempt: false only when sz_attempt is certainly false (index rejected, or
// wb 2^-sh above an upper bound of rho: fp32 exp of the fp32-rounded argument is within 1e-4 relative of det_exp for
// |arg| <= 36 pi, the factor 1.001 and the absolute slack cover that, the rounding of wb and flushed denormals).
// A "maybe" is settled by sz_decide, so the accepted attempt and value are those of the exact sampler.
__device__ inline bool sz_maybe(uint32_t wa, uint32_t wb, const SzRange rg, double center, double inv_s, long long* x_out) {
  const uint64_t prod = (uint64_t)wa * rg.N;
  if ((uint32_t)(prod & ((1ull << rg.sh) - 1)) < rg.thr) return false;
  const long long x = rg.lo + (long long)(prod >> rg.sh);
  const float a = (float)(((double)x - center) * inv_s);
  const float rho_hi = __expf(-3.14159274f * (a * a)) * 1.001f + 1e-9f;
  *x_out = x;
  return (float)wb * (rg.sh == 16 ? 0x1.0p-16f : 0x1.0p-32f) <= rho_hi;
}
// The screen for narrow words, in fp32 from the candidate word on: idx < 4096 and cand < 2^16, so the product is a 24-bit
// multiply, idx converts exactly, and a = (idx + (lo - c)) / s is evaluated with |error| < 2e-6 (c_rel = fp32(lo - c), |lo - c| <= 6s+1),
// which moves rho by < 1e-4 relative -- inside the 0.1 % margins used below.  Returns the candidate index; x = lo + idx is formed
// only for the attempt that survives.  Outcome classes: 0 = certainly rejected; 1 = certainly accepted (wb + 1 <= 0.999 rho_f 2^16
// < rho 2^16, so wb is below floor(rho 2^16) and no tie can occur); 2 = inside the +-0.1 % band around the threshold -- only
// these are settled in f64.
// Straight-line since round 4 (no early return: four screens behind four exec-mask branches were the largest single item of the rounding kernel's
// iteration, profiles/r03_notes.md): every lane evaluates the whole screen and the class is selected at the end.  rho = 2^(-pi log2(e) a^2) with the two
// constants merged into one multiply (the argument moves by < 1e-6 relative, rho by < 1.2e-4 relative at |arg| <= 113: inside the 0.1 % margins).
__device__ inline int sz_screen16(uint32_t word, const SzRange rg, float c_rel, float inv_s_f, uint32_t* idx_out) {
  const uint32_t prod = __umul24(word >> 16, rg.N);
  const bool valid = (prod & 0xffffu) >= rg.thr;                       // Lemire: the attempt is void below the threshold
  const uint32_t idx = prod >> 16;
  const float a = ((float)idx + c_rel) * inv_s_f;
  const float rho = __builtin_amdgcn_exp2f(-4.53236014f * (a * a));    // exp(-pi a^2); v_exp_f32 flushes denormals to zero: covered by the absolute slack
  const float wbf = (float)(word & 0xffffu);
  const bool maybe = wbf <= (rho * 1.001f + 1e-9f) * 65536.0f;         // not certainly rejected
  const bool sure = wbf + 1.0f <= rho * 0.999f * 65536.0f;             // certainly accepted
  *idx_out = idx;
  return (valid && maybe) ? (sure ? 1 : 2) : 0;
}
// attempts 4g .. 4g+3 of one sample: screened in fp32, the first "maybe" settled exactly; the rare screened-in-but-rejected
// case finishes the group with exact attempts, so the outcome is that of four sequential sz_attempt calls
__device__ inline bool sz_group4(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t g, const SzRange rg, double center,
                                 double inv_s, long long* x_out) {
  uint32_t wa[4], wb[4];
  int tm = -1;
  uint32_t wbm = 0;
  long long x = 0;
  if (rg.sh == 16 && fabs(center) < 0x1.0p40) {
    const U4 w = philox(seed, coord, idx_lo, g, tw);
    const uint32_t word[4] = {w.x, w.y, w.z, w.w};
    const float c_rel = (float)((double)rg.lo - center), inv_s_f = (float)inv_s;
    uint32_t idxm = 0;
    int cls = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wa[j] = word[j] >> 16; wb[j] = word[j] & 0xffffu;
      uint32_t idx = 0;
      const int cj = sz_screen16(word[j], rg, c_rel, inv_s_f, &idx);
      if (tm < 0 && cj) { tm = j; idxm = idx; cls = cj; }
    }
    if (tm >= 0) { wbm = wb[tm]; x = rg.lo + (long long)idxm; }
    if (cls == 1) { *x_out = x; return true; }     // the first surviving attempt is a certain accept: no f64 evaluation at all
  } else {
    sz_group_words(seed, coord, idx_lo, tw, g, rg.sh, wa, wb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long xc;
      if (tm < 0 && sz_maybe(wa[j], wb[j], rg, center, inv_s, &xc)) { tm = j; wbm = wb[j]; x = xc; }
    }
  }
  bool accept = false;
  if (tm >= 0) {
    accept = sz_decide(seed, coord, idx_lo, tw, 4 * g + (uint32_t)tm, x, wbm, center, inv_s, rg.sh);
    if (!accept) {
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (!accept && j > tm) accept = sz_attempt(seed, coord, idx_lo, tw, 4 * g + (uint32_t)j, wa[j], wb[j], rg, center, inv_s, &x);
    }
  }
  *x_out = x;
  return accept;
}
// sz_group4 for narrow words with the per-SAMPLE part of the screen (c_rel = fp32(lo - c), fp32(1/s)) supplied by the caller, who computes it once per
// sample instead of once per group of attempts.  Same classes, same exact decisions, same value.
__device__ inline bool sz_group4_narrow(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t g, const SzRange rg, double center,
                                        double inv_s, float c_rel, float inv_s_f, long long* x_out) {
  const U4 w = philox(seed, coord, idx_lo, g, tw);
  const uint32_t word[4] = {w.x, w.y, w.z, w.w};
  // sz_screen16 with its two questions apart (round 5): "not certainly rejected" for the four attempts, "certainly accepted" once, for the first survivor
  uint32_t prod[4];
  float rho[4];
  bool maybe[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    prod[j] = __umul24(word[j] >> 16, rg.N);
    const float a = ((float)(prod[j] >> 16) + c_rel) * inv_s_f;
    rho[j] = __builtin_amdgcn_exp2f(-4.53236014f * (a * a));
    maybe[j] = (prod[j] & 0xffffu) >= rg.thr && (float)(word[j] & 0xffffu) <= (rho[j] * 1.001f + 1e-9f) * 65536.0f;
  }
  if (!(maybe[0] || maybe[1] || maybe[2] || maybe[3])) return false;
  const int tm = maybe[0] ? 0 : maybe[1] ? 1 : maybe[2] ? 2 : 3;
  const float rm = maybe[0] ? rho[0] : maybe[1] ? rho[1] : maybe[2] ? rho[2] : rho[3];
  const uint32_t pm = maybe[0] ? prod[0] : maybe[1] ? prod[1] : maybe[2] ? prod[2] : prod[3];
  const uint32_t wm = maybe[0] ? word[0] : maybe[1] ? word[1] : maybe[2] ? word[2] : word[3];
  long long x = rg.lo + (long long)(pm >> 16);
  *x_out = x;
  const uint32_t wbm = wm & 0xffffu;
  if ((float)wbm + 1.0f <= rm * 0.999f * 65536.0f) return true;      // certainly accepted: no f64 evaluation at all
  bool accept = sz_decide(seed, coord, idx_lo, tw, 4 * g + (uint32_t)tm, x, wbm, center, inv_s, 16);
  if (!accept) {                                    // the later attempts of the group: the screen has already discarded those with maybe[j] == false
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (!accept && j > tm && maybe[j]) {
        const uint32_t wbj = word[j] & 0xffffu;
        const long long xj = rg.lo + (long long)(prod[j] >> 16);
        if ((float)wbj + 1.0f <= rho[j] * 0.999f * 65536.0f || sz_decide(seed, coord, idx_lo, tw, 4 * g + (uint32_t)j, xj, wbj, center, inv_s, 16)) { accept = true; x = xj; }
      }
    *x_out = x;
  }
  return accept;
}
#endif

// ---- table screen (round 5) ---------------------------------------------------------------------------------------------------------------------
// For one s the acceptance threshold floor(rho 2^16) of a narrow attempt depends on the candidate index and on delta = ceil(c) - c only:
// a = (idx - ceil(6 s) + delta) / s.  The host tabulates, per (idx, bin of delta) with F bins, two 16-bit bounds that hold for EVERY delta of the bin --
// certainly accepted below A, certainly rejected above R (one unit of slack on either side of the extreme floors covers the rounding of the exact
// evaluation) -- packed as R << 16 | A.  An attempt is then: 24-bit multiply, Lemire test, one LDS word, two 16-bit compares; what falls between A and R
// (a 1 / F share of the accepted mass: 3e-4 of the attempts at F = 64) is settled by sz_decide as before, so the accepted attempt and its value are
// those of the exact sampler.  The fp32 screen it replaces spends ~14 more vector instructions per attempt on conversions, the exponential and margins.
struct SzTable { const uint32_t* t; uint32_t F; uint32_t rows; };      // t[idx * F + bin], rows = n_int, F a power of two
#if defined(__HIPCC__)
// (the class of the FIRST surviving attempt is all a group needs: "maybe" is evaluated for the four attempts, "sure" once, for the one selected)
__device__ inline bool sz_group4_tab(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t g, const SzRange rg, double center,
                                     double inv_s, const uint32_t* __restrict__ tab_bin /* LDS: table + bin */, uint32_t F, long long* x_out) {
  const U4 w = philox(seed, coord, idx_lo, g, tw);
  const uint32_t word[4] = {w.x, w.y, w.z, w.w};
  uint32_t T[4], prod[4];
  bool maybe[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    prod[j] = __umul24(word[j] >> 16, rg.N);
    T[j] = tab_bin[__umul24(prod[j] >> 16, F)];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) maybe[j] = (prod[j] & 0xffffu) >= rg.thr && (word[j] & 0xffffu) <= (T[j] >> 16);      // Lemire: void below the threshold; not certainly rejected
  if (!(maybe[0] || maybe[1] || maybe[2] || maybe[3])) return false;
  const int tm = maybe[0] ? 0 : maybe[1] ? 1 : maybe[2] ? 2 : 3;
  const uint32_t Tm = maybe[0] ? T[0] : maybe[1] ? T[1] : maybe[2] ? T[2] : T[3];
  const uint32_t pm = maybe[0] ? prod[0] : maybe[1] ? prod[1] : maybe[2] ? prod[2] : prod[3];
  const uint32_t wm = maybe[0] ? word[0] : maybe[1] ? word[1] : maybe[2] ? word[2] : word[3];
  long long x = rg.lo + (long long)(pm >> 16);
  *x_out = x;
  const uint32_t wbm = wm & 0xffffu;
  if (wbm < (Tm & 0xffffu)) return true;            // certainly accepted: no f64 evaluation at all
  bool accept = sz_decide(seed, coord, idx_lo, tw, 4 * g + (uint32_t)tm, x, wbm, center, inv_s, 16);
  if (!accept) {                                    // the later attempts of the group: the screen has already discarded those with maybe[j] == false
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (!accept && j > tm && maybe[j]) {
        const uint32_t wbj = word[j] & 0xffffu;
        const long long xj = rg.lo + (long long)(prod[j] >> 16);
        if (wbj < (T[j] & 0xffffu) || sz_decide(seed, coord, idx_lo, tw, 4 * g + (uint32_t)j, xj, wbj, center, inv_s, 16)) { accept = true; x = xj; }
      }
    *x_out = x;
  }
  return accept;
}
#endif

__host__ __device__ inline long long sample_z(uint64_t seed, uint32_t tag, uint64_t index, uint32_t coord, double center,
                                              const SampleZParams sp, int* fail) {
  const SzRange rg = sz_range(center, sp);
  const uint32_t tw = tag_word(tag, index);
  long long x = 0;
  for (uint32_t g = 0; g < kMaxAttempts / 4; ++g) {
    uint32_t wa[4], wb[4];
    sz_group_words(seed, coord, (uint32_t)index, tw, g, rg.sh, wa, wb);
    for (int j = 0; j < 4; ++j)
      if (sz_attempt(seed, coord, (uint32_t)index, tw, 4 * g + (uint32_t)j, wa[j], wb[j], rg, center, sp.inv_s, &x)) return x;
  }
  *fail = 1;
  return (long long)floor(center + 0.5);
}

// N(0,1) by ratio of uniforms: x = sqrt(2/e) v/u, accepted iff u <= exp(-x^2/4).
__device__ inline double sample_normal(uint64_t seed, uint64_t index, uint32_t coord, int* fail) {
  const uint32_t tw = tag_word(TAG_NORMAL, index);
  for (uint32_t t = 0; t < kMaxAttempts; ++t) {
    const U4 w = philox(seed, coord, (uint32_t)index, t, tw);
    const double u = (double)(((((uint64_t)w.y << 32) | w.x) >> 11) + 1) * 0x1.0p-53;
    const uint64_t vv = (((uint64_t)w.w << 32) | w.z) >> 12;
    const double v = (double)(2 * vv + 1) * 0x1.0p-52 - 1.0;
    const double x = (v * 0.8577638849607068) / u;
    const double rho = det_exp(-0.25 * (x * x));
    if (u <= rho) return x;
  }
  *fail = 1;
  return 0.0;
}

// Uniform on [0, q): multiply-shift with Lemire's rejection of the 2^64 mod q lowest fractions (exactly uniform, like
// MatZq::sample_uniform at mp_perturbation.rs:222); attempt t draws block (c0, c1, t, tag), redraw probability < q / 2^64.
__host__ __device__ inline uint64_t uniform_mod(uint64_t seed, uint32_t tag, uint32_t c0, uint32_t c1, uint64_t q) {
  const uint64_t thr = (0 - q) % q;
  for (uint32_t t = 0;; ++t) {
    const U4 w = philox(seed, c0, c1, t, tag);
    const uint64_t x = ((uint64_t)w.y << 32) | w.x;
    if (x * q >= thr || t == 63) return mulhi64(x, q);
  }
}

}  // namespace psf
