/*
 * psf_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).  See psf_oracle.h for the parity status.
 *
 * Restates, in plain C over flat arrays, the algorithms of qfall/tools (paths relative to the
 * reference root):
 *   src/sample/g_trapdoor/gadget_parameters.rs, gadget_classical.rs, trapdoor_distribution.rs,
 *   short_basis_classical.rs, src/primitive/psf/mp_perturbation.rs.
 * Each function cites the lines it follows.  Floating point: IEEE-754 binary64, explicit fma(),
 * compiled with -ffp-contract=off, every summation order is written out below and is part of
 * the contract the HIP kernels are bit-compared against.
 */
#include "psf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef __int128 i128;

/* ------------------------------------------------------------------------------------------
 * Randomness contract
 * ---------------------------------------------------------------------------------------- */

/* Philox4x32-10 (Salmon et al., SC'11); key = 64-bit seed, counter = (c0,c1,c2,c3). */
void orc_philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline uint32_t tag_word(uint32_t tag, uint64_t index) { return tag | (uint32_t)((index >> 32) << 8); }

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }

/* exp(y) for y <= 0 in pure IEEE arithmetic: k = floor(y*log2e + 1/2), r = y - k*ln2 (two-part),
 * degree-13 Taylor polynomial by Horner/fma, scale by 2^k through the exponent field. */
double orc_det_exp(double y) {
  if (!(y > -708.0)) return 0.0;
  if (y > 0.0) y = 0.0;
  const double LOG2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  double kf = floor(y * LOG2E + 0.5);
  double r = fma(kf, -LN2_HI, y);
  r = fma(kf, -LN2_LO, r);
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
  int64_t k = (int64_t)kf;
  uint64_t bits;
  memcpy(&bits, &p, 8);
  bits += (uint64_t)k << 52;                       /* two's-complement add of k to the exponent field (k may be negative) */
  memcpy(&p, &bits, 8);
  return p;
}

#define ORC_MAX_ATTEMPTS 65536u

/* Z::sample_discrete_gauss / SampleZ of GPV08 as documented at CONTRIBUTING.md:35-45:
 * candidates uniform in [center - ceil(6s), center + floor(6s)], accepted with probability
 * rho_s(x - c) = exp(-pi (x-c)^2 / s^2)  (s = sigma*sqrt(2 pi)).
 * Randomness: an attempt consumes a candidate word wa and an acceptance word wb of sh bits each.
 *   narrow (sh = 16; ceil(6s) + floor(6s) + 1 <= 4096): Philox block g serves attempts 4g + j, j = 0..3, from its word j
 *          (wa = high half, wb = low half);
 *   wide   (sh = 32): block b serves attempts 2b (words x, y) and 2b + 1 (words z, w).
 *   candidate : index = (wa * N) >> sh with Lemire's rejection of the 2^sh mod N lowest fractions (exactly uniform);
 *   acceptance: U = wb * 2^32 + ext is compared with floor(rho * 2^(sh+32)); the low word `ext` is drawn
 *               lazily from block (0x80000000 | t) only when wb equals the high part of the threshold. */
#define ORC_NARROW_MAX_N 4096u

static int sz_attempt(uint64_t seed, uint32_t coord, uint32_t idx_lo, uint32_t tw, uint32_t t, uint32_t wa, uint32_t wb,
                      int64_t lo, uint32_t N, uint32_t thr, uint32_t sh, double center, double inv_s, int64_t* x_out) {
  const double NEG_PI = -3.14159265358979323846;
  uint64_t prod = (uint64_t)wa * N;
  if ((uint32_t)(prod & ((1ull << sh) - 1)) < thr) return 0;
  int64_t x = lo + (int64_t)(prod >> sh);
  double a = ((double)x - center) * inv_s;
  double rs = orc_det_exp(NEG_PI * (a * a)) * (sh == 16 ? 65536.0 : 4294967296.0);      /* rho * 2^sh, exact scaling */
  double rf = floor(rs);
  uint64_t ru = (uint64_t)rf;
  *x_out = x;
  if ((uint64_t)wb < ru) return 1;
  if ((uint64_t)wb > ru) return 0;
  uint32_t w2[4];
  orc_philox4x32(seed, coord, idx_lo, 0x80000000u | t, tw, w2);
  double rfrac = floor((rs - rf) * 4294967296.0);
  return (double)w2[0] < rfrac;
}

static unsigned long g_sample_z_cap_hits = 0;
int64_t orc_sample_z(uint64_t seed, uint32_t tag, uint64_t index, uint32_t coord, double center, double s) {
  /* a centre at or beyond 2^62 does not leave room for ceil(c) - ceil(6 s) and the candidates in 64-bit integers (the conversion below would be undefined
   * behaviour): the draw ends with 0 and is reported like a draw that ended at the attempt cap; the device does the same (psf_np_kernels.hpp) */
  if (!(fabs(center) < 0x1.0p62)) { __atomic_fetch_add(&g_sample_z_cap_hits, 1ul, __ATOMIC_RELAXED); return 0; }
  double inv_s = 1.0 / s;
  int64_t c6 = (int64_t)ceil(6.0 * s), f6 = (int64_t)floor(6.0 * s);
  int64_t lo = (int64_t)ceil(center) - c6;
  int64_t hi = (int64_t)floor(center) + f6;
  uint32_t N = (uint32_t)(hi - lo + 1);
  uint32_t sh = (uint64_t)(c6 + f6 + 1) <= ORC_NARROW_MAX_N ? 16 : 32;   /* by s alone, not by the centre */
  uint32_t thr = (uint32_t)((1ull << sh) % N);
  uint32_t tw = tag_word(tag, index);
  uint32_t w[4], v[4];
  int64_t x;
  for (uint32_t g = 0; g < ORC_MAX_ATTEMPTS / 4; ++g) {
    uint32_t wa[4], wb[4];
    if (sh == 16) {
      orc_philox4x32(seed, coord, (uint32_t)index, g, tw, w);
      for (int j = 0; j < 4; ++j) { wa[j] = w[j] >> 16; wb[j] = w[j] & 0xffffu; }
    } else {
      orc_philox4x32(seed, coord, (uint32_t)index, 2 * g, tw, w);
      orc_philox4x32(seed, coord, (uint32_t)index, 2 * g + 1, tw, v);
      wa[0] = w[0]; wb[0] = w[1]; wa[1] = w[2]; wb[1] = w[3];
      wa[2] = v[0]; wb[2] = v[1]; wa[3] = v[2]; wb[3] = v[3];
    }
    for (uint32_t j = 0; j < 4; ++j)
      if (sz_attempt(seed, coord, (uint32_t)index, tw, 4 * g + j, wa[j], wb[j], lo, N, thr, sh, center, inv_s, &x)) return x;
  }
  /* ORC_MAX_ATTEMPTS attempts without an accept (a width far below 1 with a half-integral centre: acceptance ~e^-25).  The reference would keep
   * drawing; the contract ends the draw with the nearest integer AND reports it: the sampling entry points return ORC_ERR_SAMPLER when this counter
   * moved during their call (the device raises its failure flag in the same place, PSF_ERR_SAMPLER). */
  __atomic_fetch_add(&g_sample_z_cap_hits, 1ul, __ATOMIC_RELAXED);
  return (int64_t)floor(center + 0.5);
}
unsigned long orc_sample_z_cap_hits(void) { return __atomic_load_n(&g_sample_z_cap_hits, __ATOMIC_RELAXED); }
/* status of a sampling entry point: ORC_ERR_SAMPLER when a draw of the call ended at the attempt cap (cap0 = the counter at entry) */
static int cap_status(unsigned long cap0, int rc) { return (rc == ORC_OK && orc_sample_z_cap_hits() != cap0) ? ORC_ERR_SAMPLER : rc; }

/* N(0,1) by Kinderman-Monahan ratio of uniforms: x = sqrt(2/e) v / u, accept iff u <= exp(-x^2/4). */
double orc_sample_normal(uint64_t seed, uint64_t index, uint32_t coord) {
  const double C_RU = 0.8577638849607068; /* sqrt(2/e) */
  uint32_t w[4];
  for (uint32_t t = 0; t < ORC_MAX_ATTEMPTS; ++t) {
    orc_philox4x32(seed, coord, (uint32_t)index, t, tag_word(ORC_TAG_NORMAL, index), w);
    double u = (double)(((((uint64_t)w[1] << 32) | w[0]) >> 11) + 1) * 0x1.0p-53; /* (0,1] */
    uint64_t vv = (((uint64_t)w[3] << 32) | w[2]) >> 12;                          /* 52 bits */
    double v = (double)(2 * vv + 1) * 0x1.0p-52 - 1.0;                            /* (-1,1), exact */
    double x = (v * C_RU) / u;
    double rho = orc_det_exp(-0.25 * (x * x));
    if (u <= rho) return x;
  }
  return 0.0;
}

/* Uniform on [0, q): 64-bit multiply-shift with Lemire's rejection of the (2^64 mod q) lowest fractions, so every
 * residue has exactly floor(2^64 / q) accepted words (MatZq::sample_uniform, mp_perturbation.rs:222, is exactly uniform).
 * Attempt t draws Philox block (c0, c1, t, tag); a redraw happens with probability < q / 2^64. */
uint64_t orc_uniform_mod(uint64_t seed, uint32_t tag, uint32_t c0, uint32_t c1, uint64_t q) {
  uint32_t w[4];
  const uint64_t thr = (0 - q) % q; /* 2^64 mod q */
  for (uint32_t t = 0;; ++t) {
    orc_philox4x32(seed, c0, c1, t, tag, w);
    const uint64_t x = ((uint64_t)w[1] << 32) | w[0];
    if (x * q >= thr || t == 63) return mulhi64(x, q);
  }
}

/* ------------------------------------------------------------------------------------------
 * small integer helpers
 * ---------------------------------------------------------------------------------------- */
static uint64_t log_ceil(uint64_t x, uint64_t base) { /* smallest e with base^e >= x */
  uint64_t e = 0;
  u128 p = 1;
  while (p < x) { p *= base; ++e; }
  return e;
}
static int pow_u128(uint64_t base, uint64_t k, u128* out) {
  u128 p = 1;
  for (uint64_t i = 0; i < k; ++i) {
    if (p > (((u128)1) << 100)) return 1;
    p *= base;
  }
  *out = p;
  return 0;
}
static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t s = a + b; return (s >= q || s < a) ? s - q : s; }
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static inline uint64_t reduce_i128(i128 v, uint64_t q) { i128 r = v % (i128)q; if (r < 0) r += q; return (uint64_t)r; }

/* gadget_parameters.rs:113-133: base 2, k = ceil(log2 q), m_bar = n k + ceil(log2 n)^2 */
int orc_gadget_params_default(uint64_t n, uint64_t q, orc_gadget_params* gp) {
  if (n < 1 || q <= 1) return ORC_ERR_PARAM;
  gp->n = n; gp->base = 2; gp->q = q;
  gp->k = log_ceil(q, 2);
  uint64_t ln = log_ceil(n, 2);
  gp->m_bar = n * gp->k + ln * ln;
  return ORC_OK;
}
/* gadget_parameters.rs:165-185: m_bar = k + 2 */
int orc_gadget_params_ring_default(uint64_t n, uint64_t q, orc_gadget_params* gp) {
  if (n < 1 || q <= 1) return ORC_ERR_PARAM;
  gp->n = n; gp->base = 2; gp->q = q;
  gp->k = log_ceil(q, 2);
  gp->m_bar = gp->k + 2;
  return ORC_OK;
}

/* gadget_classical.rs:128-136 */
int orc_gen_gadget_vec(uint64_t k, uint64_t base, int64_t* out) {
  if (k < 1) return ORC_ERR_PARAM;
  int64_t entry = 1;
  for (uint64_t i = 0; i < k; ++i) { out[i] = entry; entry *= (int64_t)base; }
  return ORC_OK;
}
/* gadget_classical.rs:91-107: I_n (x) g^t */
int orc_gen_gadget_mat(uint64_t n, uint64_t k, uint64_t base, int64_t* out) {
  if (n < 1 || k < 1) return ORC_ERR_PARAM;
  int64_t* g = (int64_t*)malloc(k * sizeof(int64_t));
  orc_gen_gadget_vec(k, base, g);
  memset(out, 0, n * n * k * sizeof(int64_t));
  for (uint64_t j = 0; j < n; ++j)
    for (uint64_t i = 0; i < k; ++i) out[j * (n * k) + j * k + i] = g[i];
  free(g);
  return ORC_OK;
}
/* gadget_classical.rs:169-182: LSB-first base-b digits of the least non-negative residue */
int orc_find_solution_gadget_vec(uint64_t value, uint64_t q, uint64_t k, uint64_t base, int64_t* out) {
  u128 bk;
  if (pow_u128(base, k, &bk) == 0 && bk < q) return ORC_ERR_MODULUS; /* :170-172 */
  value %= q;
  for (uint64_t i = 0; i < k; ++i) {
    uint64_t d = value % base;
    out[i] = (int64_t)d;
    value = (value - d) / base;
  }
  return ORC_OK;
}
/* gadget_classical.rs:219-229: out[k*j + i, col] = digit_i(value[j, col]) */
int orc_find_solution_gadget_mat(const uint64_t* value, size_t rows, size_t cols, uint64_t q, uint64_t k,
                                 uint64_t base, int64_t* out) {
  int64_t* d = (int64_t*)malloc(k * sizeof(int64_t));
  for (size_t i = 0; i < cols; ++i)
    for (size_t j = 0; j < rows; ++j) {
      int rc = orc_find_solution_gadget_vec(value[j * cols + i], q, k, base, d);
      if (rc) { free(d); return rc; }
      for (uint64_t t = 0; t < k; ++t) out[(k * j + t) * cols + i] = d[t];
    }
  free(d);
  return ORC_OK;
}
/* gadget_classical.rs:249-272 */
int orc_short_basis_gadget_block(const orc_gadget_params* gp, int64_t* sk) {
  size_t k = gp->k;
  memset(sk, 0, k * k * sizeof(int64_t));
  for (size_t j = 0; j < k; ++j) sk[j * k + j] = (int64_t)gp->base;       /* :252-254 */
  for (size_t i = 0; i + 1 < k; ++i) sk[(i + 1) * k + i] = -1;            /* :255-257 */
  u128 bk;
  int big = pow_u128(gp->base, gp->k, &bk);
  if (big || bk != gp->q) {                                                /* :258-272 */
    uint64_t q = gp->q;
    for (size_t i = 0; i < k; ++i) {
      uint64_t qi = q % gp->base;
      sk[i * k + (k - 1)] = (int64_t)qi;
      q = (q - qi) / gp->base;
    }
  }
  return ORC_OK;
}
/* gadget_classical.rs:273-286: I_n (x) S_k */
int orc_short_basis_gadget(const orc_gadget_params* gp, int64_t* out) {
  size_t n = gp->n, k = gp->k, w = n * k;
  int64_t* sk = (int64_t*)malloc(k * k * sizeof(int64_t));
  orc_short_basis_gadget_block(gp, sk);
  memset(out, 0, w * w * sizeof(int64_t));
  for (size_t j = 0; j < n; ++j)
    for (size_t a = 0; a < k; ++a)
      for (size_t b = 0; b < k; ++b) out[(j * k + a) * w + (j * k + b)] = sk[a * k + b];
  free(sk);
  return ORC_OK;
}

/* MatQ::gso (mp_perturbation.rs:234): Gram-Schmidt on the COLUMNS, no normalisation.
 * b~_i = b_i - sum_{l<i} (<b_i, b~_l> / <b~_l, b~_l>) b~_l ; dots are ascending fma chains. */
void orc_gso_columns(const int64_t* basis, size_t dim, double* gso) {
  double* norm2 = (double*)malloc(dim * sizeof(double));
  for (size_t i = 0; i < dim; ++i) {
    for (size_t t = 0; t < dim; ++t) gso[t * dim + i] = (double)basis[t * dim + i];
    for (size_t l = 0; l < i; ++l) {
      double num = 0.0;
      for (size_t t = 0; t < dim; ++t) num = fma((double)basis[t * dim + i], gso[t * dim + l], num);
      double mu = num / norm2[l];
      for (size_t t = 0; t < dim; ++t) gso[t * dim + i] = fma(-mu, gso[t * dim + l], gso[t * dim + i]);
    }
    double nn = 0.0;
    for (size_t t = 0; t < dim; ++t) nn = fma(gso[t * dim + i], gso[t * dim + i], nn);
    norm2[i] = nn;
  }
  free(norm2);
}

/* trapdoor_distribution.rs:82-86: difference of two uniform bits; 64 entries per Philox block */
void orc_sample_r(uint64_t seed, size_t m_bar, size_t w, int8_t* R) {
  uint32_t wd[4];
  for (size_t i = 0; i < m_bar; ++i)
    for (size_t j = 0; j < w; ++j) {
      if ((j & 63) == 0 || j == 0) orc_philox4x32(seed, (uint32_t)(j >> 6), (uint32_t)i, 0, ORC_TAG_R, wd);
      uint32_t word = wd[(j & 63) >> 4];
      uint32_t sh = 2 * (j & 15);
      R[i * w + j] = (int8_t)((int)((word >> sh) & 1) - (int)((word >> (sh + 1)) & 1));
    }
}
/* mp_perturbation.rs:222 MatZq::sample_uniform */
void orc_sample_a_bar(uint64_t seed, size_t n, size_t m_bar, uint64_t q, uint64_t* a_bar) {
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < m_bar; ++j) a_bar[i * m_bar + j] = orc_uniform_mod(seed, ORC_TAG_ABAR, (uint32_t)j, (uint32_t)i, q);
}

/* gadget_classical.rs:56-68: A = [A_bar | tag*G - A_bar*R] */
int orc_gen_trapdoor(const orc_gadget_params* gp, const uint64_t* a_bar, const uint64_t* tag, const int8_t* R, uint64_t* A) {
  size_t n = gp->n, k = gp->k, mb = gp->m_bar, w = n * k, m = mb + w;
  uint64_t q = gp->q;
  uint64_t* gvec = (uint64_t*)malloc(k * sizeof(uint64_t));
  uint64_t e = 1 % q;
  for (size_t t = 0; t < k; ++t) { gvec[t] = e; e = mulmod(e, gp->base % q, q); }
  uint64_t* acc = (uint64_t*)malloc(w * sizeof(uint64_t));
  for (size_t i = 0; i < n; ++i) {
    for (size_t j = 0; j < mb; ++j) A[i * m + j] = a_bar[i * mb + j] % q;
    memset(acc, 0, w * sizeof(uint64_t));
    for (size_t t = 0; t < mb; ++t) {
      uint64_t a = a_bar[i * mb + t] % q;
      const int8_t* Rt = R + t * w;
      for (size_t c = 0; c < w; ++c) {
        if (Rt[c] == 1) acc[c] = addmod(acc[c], a, q);
        else if (Rt[c] == -1) acc[c] = submod(acc[c], a, q);
        else if (Rt[c] != 0) acc[c] = reduce_i128((i128)acc[c] + (i128)a * Rt[c], q);
      }
    }
    for (size_t j = 0; j < n; ++j) {
      uint64_t h = tag ? tag[i * n + j] % q : (i == j ? 1 % q : 0);
      for (size_t t = 0; t < k; ++t) {
        uint64_t hg = h ? mulmod(h, gvec[t], q) : 0;
        A[i * m + mb + j * k + t] = submod(hg, acc[j * k + t], q);
      }
    }
  }
  free(acc); free(gvec);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * short_basis_classical.rs
 * ---------------------------------------------------------------------------------------- */
/* :66-74  [ I | R ; 0 | I ] */
int orc_gen_sa_l(const int8_t* R, size_t m_bar, size_t w, int64_t* out) {
  size_t m = m_bar + w;
  memset(out, 0, m * m * sizeof(int64_t));
  for (size_t i = 0; i < m; ++i) out[i * m + i] = 1;
  for (size_t i = 0; i < m_bar; ++i)
    for (size_t j = 0; j < w; ++j) out[i * m + m_bar + j] = R[i * w + j];
  return ORC_OK;
}

/* inverse of an n x n matrix mod q by Gauss-Jordan with unit pivots (tag.inverse(), :106) */
static int mat_inverse_mod(const uint64_t* M, size_t n, uint64_t q, uint64_t* inv);

/* :105-110  G W = -H^{-1} A [I | 0]^t mod q */
int orc_compute_w(const orc_gadget_params* gp, const uint64_t* tag, const uint64_t* A, int64_t* W) {
  size_t n = gp->n, k = gp->k, mb = gp->m_bar, m = mb + n * k;
  uint64_t q = gp->q;
  uint64_t* rhs = (uint64_t*)malloc(n * mb * sizeof(uint64_t));
  uint64_t* tinv = NULL;
  if (tag) {
    tinv = (uint64_t*)malloc(n * n * sizeof(uint64_t));
    if (mat_inverse_mod(tag, n, q, tinv)) { free(rhs); free(tinv); return ORC_ERR_PARAM; }
  }
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < mb; ++j) {
      uint64_t v;
      if (tinv) {
        u128 s = 0;
        for (size_t t = 0; t < n; ++t) s = (s + (u128)tinv[i * n + t] * (A[t * m + j] % q)) % q;
        v = (uint64_t)s;
      } else v = A[i * m + j] % q;
      rhs[i * mb + j] = v ? q - v : 0;
    }
  int rc = orc_find_solution_gadget_mat(rhs, n, mb, q, k, gp->base, W);
  free(rhs); free(tinv);
  return rc;
}

/* :77-102  [ 0 | I ; S' | W ], S' = column-reversed S iff base^k == q (:80-82) */
int orc_gen_sa_r(const orc_gadget_params* gp, const uint64_t* tag, const uint64_t* A, int64_t* out) {
  size_t n = gp->n, k = gp->k, mb = gp->m_bar, w = n * k, m = mb + w;
  int64_t* S = (int64_t*)malloc(w * w * sizeof(int64_t));
  orc_short_basis_gadget(gp, S);
  u128 bk;
  int reversed = (pow_u128(gp->base, gp->k, &bk) == 0 && bk == gp->q);
  int64_t* W = (int64_t*)malloc(w * mb * sizeof(int64_t));
  int rc = orc_compute_w(gp, tag, A, W);
  if (rc) { free(S); free(W); return rc; }
  memset(out, 0, m * m * sizeof(int64_t));
  for (size_t d = 0; d < mb; ++d) out[d * m + (w + d)] = 1;                     /* :90-93 */
  for (size_t i = 0; i < w; ++i) {
    for (size_t j = 0; j < w; ++j) out[(mb + i) * m + j] = S[i * w + (reversed ? (w - 1 - j) : j)];
    for (size_t j = 0; j < mb; ++j) out[(mb + i) * m + w + j] = W[i * mb + j];
  }
  free(S); free(W);
  return ORC_OK;
}

/* :54-63  S_A = sa_l * sa_r  (structured: top = [R S' | I + R W], bottom = [S' | W]) */
int orc_gen_short_basis_for_trapdoor(const orc_gadget_params* gp, const uint64_t* tag, const uint64_t* A,
                                     const int8_t* R, int64_t* out) {
  size_t n = gp->n, k = gp->k, mb = gp->m_bar, w = n * k, m = mb + w;
  int64_t* sar = (int64_t*)malloc(m * m * sizeof(int64_t));
  int rc = orc_gen_sa_r(gp, tag, A, sar);
  if (rc) { free(sar); return rc; }
  /* rows >= m_bar of sa_l are unit rows */
  memcpy(out + mb * m, sar + mb * m, w * m * sizeof(int64_t));
  for (size_t i = 0; i < mb; ++i) {
    int64_t* o = out + i * m;
    memcpy(o, sar + i * m, m * sizeof(int64_t));
    for (size_t t = 0; t < w; ++t) {
      int64_t rv = R[i * w + t];
      if (!rv) continue;
      const int64_t* srow = sar + (mb + t) * m;
      for (size_t j = 0; j < m; ++j) o[j] += rv * srow[j];
    }
  }
  free(sar);
  return ORC_OK;
}

static uint64_t inv_mod(uint64_t a, uint64_t q, int* ok) {
  i128 t = 0, nt = 1, r = q, nr = a % q;
  while (nr != 0) {
    i128 qq = r / nr;
    i128 tmp = t - qq * nt; t = nt; nt = tmp;
    tmp = r - qq * nr; r = nr; nr = tmp;
  }
  if (r != 1) { *ok = 0; return 0; }
  *ok = 1;
  if (t < 0) t += q;
  return (uint64_t)t;
}
static int mat_inverse_mod(const uint64_t* M, size_t n, uint64_t q, uint64_t* inv) {
  uint64_t* a = (uint64_t*)malloc(n * n * sizeof(uint64_t));
  for (size_t i = 0; i < n * n; ++i) a[i] = M[i] % q;
  for (size_t i = 0; i < n; ++i) for (size_t j = 0; j < n; ++j) inv[i * n + j] = (i == j) ? 1 % q : 0;
  for (size_t c = 0; c < n; ++c) {
    size_t p = n; uint64_t pinv = 0;
    for (size_t r = c; r < n; ++r) { int ok; pinv = inv_mod(a[r * n + c], q, &ok); if (ok) { p = r; break; } }
    if (p == n) { free(a); return 1; }
    if (p != c) for (size_t j = 0; j < n; ++j) {
      uint64_t t = a[p * n + j]; a[p * n + j] = a[c * n + j]; a[c * n + j] = t;
      t = inv[p * n + j]; inv[p * n + j] = inv[c * n + j]; inv[c * n + j] = t;
    }
    for (size_t j = 0; j < n; ++j) { a[c * n + j] = mulmod(a[c * n + j], pinv, q); inv[c * n + j] = mulmod(inv[c * n + j], pinv, q); }
    for (size_t r = 0; r < n; ++r) {
      if (r == c) continue;
      uint64_t f = a[r * n + c];
      if (!f) continue;
      for (size_t j = 0; j < n; ++j) {
        a[r * n + j] = submod(a[r * n + j], mulmod(f, a[c * n + j], q), q);
        inv[r * n + j] = submod(inv[r * n + j], mulmod(f, inv[c * n + j], q), q);
      }
    }
  }
  free(a);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * PSFPerturbation (mp_perturbation.rs)
 * ---------------------------------------------------------------------------------------- */
static orc_psfp* psfp_new_impl(const orc_gadget_params* gp, double r, double s, int with_L);
orc_psfp* orc_psfp_new(const orc_gadget_params* gp, double r, double s) { return psfp_new_impl(gp, r, s, 1); }
/* the same object without storage for sqrt(Sigma_2) (h->L == NULL): for keys too large to hold twice in host memory, whose
 * centres x = sqrt(Sigma_2) d are checked through orc_psfp_centres_rows on streamed row blocks */
orc_psfp* orc_psfp_new_nokey(const orc_gadget_params* gp, double r, double s) { return psfp_new_impl(gp, r, s, 0); }
static orc_psfp* psfp_new_impl(const orc_gadget_params* gp, double r, double s, int with_L) {
  if (!gp || gp->n < 1 || gp->k < 1 || gp->q <= 1 || !(r > 0) || !(s > 0)) return NULL;
  orc_psfp* h = (orc_psfp*)calloc(1, sizeof(orc_psfp));
  h->gp = *gp; h->r = r; h->s = s;
  size_t w = gp->n * gp->k;
  h->m = gp->m_bar + w;
  h->A = (uint64_t*)calloc(gp->n * h->m, sizeof(uint64_t));
  h->R = (int8_t*)calloc(gp->m_bar * w, 1);
  h->L = with_L ? (double*)calloc(h->m * (h->m + 1) / 2, sizeof(double)) : NULL;
  h->Sk = (int64_t*)calloc(gp->k * gp->k, sizeof(int64_t));
  h->Sk_gso = (double*)calloc(gp->k * gp->k, sizeof(double));
  /* mp_perturbation.rs:233-234: short_basis_gadget + gso.  I_n (x) S_k is block diagonal, so its GSO is
   * I_n (x) GSO(S_k); only the k x k block is stored. */
  orc_short_basis_gadget_block(gp, h->Sk);
  orc_gso_columns(h->Sk, gp->k, h->Sk_gso);
  return h;
}
void orc_psfp_free(orc_psfp* h) {
  if (!h) return;
  free(h->A); free(h->R); free(h->L); free(h->Sk); free(h->Sk_gso); free(h);
}

/* mp_perturbation.rs:111-139 with Sigma = s_cov^2 I (the form trap_gen passes at :227-231):
 *   Sigma_2 = (1/2pi) r^2 ((Sigma - (b^2+1) T T^t) - I),  T = [R; I_w];  returns its lower Cholesky factor. */
static int sqrt_sigma_2_rows(const orc_psfp* h, const int8_t* R, double s_cov, const double* sigma, size_t m, double* Lp);
int orc_psfp_compute_sqrt_sigma_2(const orc_psfp* h, const int8_t* R, double s_cov, double* Lp) {
  return sqrt_sigma_2_rows(h, R, s_cov, NULL, h->gp.m_bar + h->gp.n * h->gp.k, Lp);
}
/* the general form of mp_perturbation.rs:111: any symmetric mat_sigma, given as its packed lower triangle (row i: i + 1 entries) */
int orc_psfp_compute_sqrt_sigma_2_dense(const orc_psfp* h, const int8_t* R, const double* sigma_packed, double* Lp) {
  return sqrt_sigma_2_rows(h, R, 0.0, sigma_packed, h->gp.m_bar + h->gp.n * h->gp.k, Lp);
}
/* The first m0 rows of the same factor.  Row i of the Cholesky-Banachiewicz recurrence only reads rows <= i, so this is the
 * factor of the leading m0 x m0 block of Sigma_2 and at the same time rows 0..m0-1 of the full factor: a cheap check of a
 * device factor that spans many panels at sizes where the whole recurrence (m^3/3) is out of reach for a scalar CPU loop. */
int orc_psfp_sqrt_sigma_2_leading(const orc_psfp* h, const int8_t* R, double s_cov, size_t m0, double* Lp) {
  if (m0 > h->gp.m_bar + h->gp.n * h->gp.k) return ORC_ERR_PARAM;
  return sqrt_sigma_2_rows(h, R, s_cov, NULL, m0, Lp);
}
static int sqrt_sigma_2_rows(const orc_psfp* h, const int8_t* R, double s_cov, const double* sigma, size_t m, double* Lp) {
  const orc_gadget_params* gp = &h->gp;
  size_t mb = gp->m_bar, w = gp->n * gp->k;
  const double TWO_PI = 6.283185307179586476925;
  double nf_r2 = (1.0 / TWO_PI) * (h->r * h->r);            /* :113, :132-133 */
  double s2 = s_cov * s_cov;
  int64_t b2p1 = (int64_t)(gp->base * gp->base + 1);        /* :126 */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8)
#endif
  for (size_t i = 0; i < m; ++i) {
    double* row = Lp + i * (i + 1) / 2;
    for (size_t j = 0; j <= i; ++j) {
      int64_t tt;
      if (i < mb) { /* both in the R block */
        int32_t acc = 0;
        const int8_t *ri = R + i * w, *rj = R + j * w;
        for (size_t c = 0; c < w; ++c) acc += (int32_t)ri[c] * rj[c];
        tt = acc;
      } else if (j < mb) tt = R[j * w + (i - mb)];
      else tt = (i == j);
      const double sg = sigma ? sigma[i * (i + 1) / 2 + j] : (i == j ? s2 : 0.0);
      double sp = sg - (double)(b2p1 * tt);                     /* Sigma_p entry, :125-126 */
      if (i == j) sp = sp - 1.0;                                /* - I, :134-135 */
      row[j] = nf_r2 * sp;
    }
  }
  /* cholesky_decomposition_flint (:138): row-wise Cholesky-Banachiewicz, ascending fma chains */
  for (size_t i = 0; i < m; ++i) {
    double* li = Lp + i * (i + 1) / 2;
    for (size_t j = 0; j <= i; ++j) {
      const double* lj = Lp + j * (j + 1) / 2;
      double sum = li[j];
      for (size_t t = 0; t < j; ++t) sum = fma(-li[t], lj[t], sum);
      if (i == j) {
        if (!(sum > 0.0)) return ORC_ERR_NOT_PD;            /* :109-110 */
        li[j] = sqrt(sum);
      } else li[j] = sum / lj[j];
    }
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Structured square root of Sigma_2 (the library's labelled opt-in, PSFP_FLAG_STRUCTURED_SQRT; not the reference's Cholesky path):
 *   Sigma_2 = c [[alpha I - kappa R R^t, -kappa R], [-kappa R^t, beta I]]   (c = r^2 / 2 pi, kappa = b^2 + 1, alpha = s^2 - 1, beta = alpha - kappa)
 *           = B B^t,  B = [[L_1, -g R], [0, h I]],  L_1 L_1^t = c (alpha I - kappa (alpha / beta) R R^t),  g = sqrt(c) kappa / sqrt(beta),  h = sqrt(c beta)
 * so x = B d is  x_top = L_1 d_1 - g R d_2,  x_bot = h d_2  -- the same distribution as sqrt(Sigma_2) d of mp_perturbation.rs:315 for ANY square root.
 * Contract: d_1 = normals of coordinates < m_bar; d_2[c] = q_c 2^-32 with q_c = floor(n 2^32 + 1/2) for the normal n of coordinate m_bar + c;
 * y_i = ascending fma chain of L_1 d_1; x_i = fma(-g, (double)(sum_c R[i][c] q_c) 2^-32, y_i) (the integer sum is exact); x_{m_bar+c} = h d_2[c].
 * ---------------------------------------------------------------------------------------- */
void orc_psfp_structured_constants(const orc_psfp* h, double s_cov, double* g, double* hh, double* kab) {
  const double TWO_PI = 6.283185307179586476925;
  const double nf_r2 = (1.0 / TWO_PI) * (h->r * h->r);
  const double kappa = (double)(h->gp.base * h->gp.base + 1), alpha = s_cov * s_cov - 1.0, beta = alpha - kappa;
  *g = (sqrt(nf_r2) * kappa) / sqrt(beta);
  *hh = sqrt(nf_r2 * beta);
  *kab = kappa * (alpha / beta);
}

/* L_1 (m_bar x m_bar lower, packed by rows) */
int orc_psfp_structured_sqrt(const orc_psfp* h, const int8_t* R, double s_cov, double* Lp) {
  const orc_gadget_params* gp = &h->gp;
  const size_t mb = gp->m_bar, w = gp->n * gp->k;
  const double TWO_PI = 6.283185307179586476925;
  const double nf_r2 = (1.0 / TWO_PI) * (h->r * h->r), s2 = s_cov * s_cov;
  double g, hh, kab;
  orc_psfp_structured_constants(h, s_cov, &g, &hh, &kab);
  if (!((s2 - 1.0) - (double)(gp->base * gp->base + 1) > 0.0)) return ORC_ERR_NOT_PD;
  for (size_t i = 0; i < mb; ++i) {
    double* row = Lp + i * (i + 1) / 2;
    for (size_t j = 0; j <= i; ++j) {
      int32_t acc = 0;
      const int8_t *ri = R + i * w, *rj = R + j * w;
      for (size_t c = 0; c < w; ++c) acc += (int32_t)ri[c] * rj[c];
      double sp = (i == j ? s2 : 0.0) - kab * (double)acc;       /* same expression shape as the dense assembly */
      if (i == j) sp = sp - 1.0;
      row[j] = nf_r2 * sp;
    }
  }
  for (size_t i = 0; i < mb; ++i) {
    double* li = Lp + i * (i + 1) / 2;
    for (size_t j = 0; j <= i; ++j) {
      const double* lj = Lp + j * (j + 1) / 2;
      double sum = li[j];
      for (size_t t = 0; t < j; ++t) sum = fma(-li[t], lj[t], sum);
      if (i == j) {
        if (!(sum > 0.0)) return ORC_ERR_NOT_PD;
        li[j] = sqrt(sum);
      } else li[j] = sum / lj[j];
    }
  }
  return ORC_OK;
}

/* one preimage in structured mode, every intermediate exposed; L1p as above, s_cov the Gaussian parameter the factor was built for */
int orc_psfp_samp_p_structured_trace(const orc_psfp* h, const double* L1p, double s_cov, uint64_t seed, uint64_t index, const uint64_t* u,
                                     double* d, double* x, int64_t* p, uint64_t* v, int64_t* z, int64_t* e) {
  const size_t mb = h->gp.m_bar, w = h->gp.n * h->gp.k, m = h->m;
  double g, hh, kab;
  orc_psfp_structured_constants(h, s_cov, &g, &hh, &kab);
  int64_t* q = (int64_t*)malloc(w * sizeof(int64_t));
  for (size_t j = 0; j < mb; ++j) d[j] = orc_sample_normal(seed, index, (uint32_t)j);
  for (size_t c = 0; c < w; ++c) {
    const double nrm = orc_sample_normal(seed, index, (uint32_t)(mb + c));
    const double sc = floor(nrm * 0x1.0p32 + 0.5);
    q[c] = (int64_t)sc;
    d[mb + c] = sc * 0x1.0p-32;
  }
  for (size_t i = 0; i < mb; ++i) {
    const double* li = L1p + i * (i + 1) / 2;
    double acc = 0.0;
    for (size_t j = 0; j <= i; ++j) acc = fma(li[j], d[j], acc);
    const int8_t* ri = h->R + i * w;
    int64_t tot = 0;
    for (size_t c = 0; c < w; ++c) tot += (int64_t)ri[c] * q[c];
    x[i] = fma(-g, (double)tot * 0x1.0p-32, acc);
  }
  for (size_t c = 0; c < w; ++c) x[mb + c] = hh * d[mb + c];
  free(q);
  (void)m;
  return orc_psfp_samp_p_from_x(h, seed, index, u, x, p, v, z, e);
}

/* mp_perturbation.rs:221-244 (tag = identity, :223) */
int orc_psfp_trap_gen(orc_psfp* h, uint64_t seed) {
  const orc_gadget_params* gp = &h->gp;
  size_t n = gp->n, mb = gp->m_bar, w = n * gp->k;
  uint64_t* a_bar = (uint64_t*)malloc(n * mb * sizeof(uint64_t));
  orc_sample_a_bar(seed, n, mb, gp->q, a_bar);                      /* :222 */
  orc_sample_r(seed, mb, w, h->R);                                   /* gadget_classical.rs:62-64 */
  int rc = orc_gen_trapdoor(gp, a_bar, NULL, h->R, h->A);            /* :225 */
  free(a_bar);
  if (rc) return rc;
  return orc_psfp_compute_sqrt_sigma_2(h, h->R, h->s, h->L);          /* :227-231 */
}

int orc_psfp_load_key(orc_psfp* h, const uint64_t* A, const int8_t* R, const double* Lp) {
  size_t n = h->gp.n, w = n * h->gp.k, m = h->m;
  memcpy(h->A, A, n * m * sizeof(uint64_t));
  memcpy(h->R, R, h->gp.m_bar * w);
  if (h->L && Lp) memcpy(h->L, Lp, m * (m + 1) / 2 * sizeof(double));
  return ORC_OK;
}

/* per-gadget-block tables: ||b~_i||^2 and s' = s_G / ||b~_i|| */
static void gadget_tables(const orc_psfp* h, double* norm2, double* s2) {
  size_t k = h->gp.k;
  double sG = h->r * sqrt((double)(h->gp.base * h->gp.base + 1));   /* mp_perturbation.rs:180 */
  for (size_t i = 0; i < k; ++i) {
    double nn = 0.0;
    for (size_t t = 0; t < k; ++t) nn = fma(h->Sk_gso[t * k + i], h->Sk_gso[t * k + i], nn);
    norm2[i] = nn;
    s2[i] = sG / sqrt(nn);
  }
}

/* mp_perturbation.rs:173-191 + MatZ::sample_d_precomputed_gso (GPV08 SampleD) on I_n (x) S_k:
 *   x = G^{-1}(v) (gadget_classical.rs:219-229); c = -x; for i = k-1..0 (per block j):
 *   c' = <c, b~_i>/||b~_i||^2 ; z_i <- D_{Z, s_G/||b~_i||, c'} ; c -= z_i b_i ; result z = x + sum z_i b_i = -c. */
static int gadget_sample_one(const orc_psfp* h, const double* norm2, const double* s2, uint64_t seed,
                             uint64_t index, const uint64_t* v, int64_t* z) {
  size_t n = h->gp.n, k = h->gp.k;
  int64_t* c = (int64_t*)malloc(k * sizeof(int64_t));
  for (size_t j = 0; j < n; ++j) {
    int rc = orc_find_solution_gadget_vec(v[j], h->gp.q, k, h->gp.base, c);
    if (rc) { free(c); return rc; }
    for (size_t t = 0; t < k; ++t) c[t] = -c[t];
    for (size_t ii = k; ii-- > 0;) {
      double dot = 0.0;
      for (size_t t = 0; t < k; ++t) dot = fma((double)c[t], h->Sk_gso[t * k + ii], dot);
      double c2 = dot / norm2[ii];
      int64_t zi = orc_sample_z(seed, ORC_TAG_GADGET, index, (uint32_t)(j * k + ii), c2, s2[ii]);
      for (size_t t = 0; t < k; ++t) c[t] -= zi * h->Sk[t * k + ii];
    }
    for (size_t t = 0; t < k; ++t) z[j * k + t] = -c[t];
  }
  free(c);
  return ORC_OK;
}

int orc_randomized_nearest_plane_gadget(const orc_psfp* h, uint64_t seed, uint64_t index, const uint64_t* v, int64_t* z) {
  size_t k = h->gp.k;
  double* norm2 = (double*)malloc(2 * k * sizeof(double));
  gadget_tables(h, norm2, norm2 + k);
  int rc = gadget_sample_one(h, norm2, norm2 + k, seed, index, v, z);
  free(norm2);
  return rc;
}

/* x[b][i - row0] = sum_{j <= i} L[i][j] d[b][j] for the rows i = row0 .. row0 + nrows - 1 of sqrt(Sigma_2), given as a packed
 * row block (row i holds i + 1 entries, the block starts at row row0): the contract's ascending fma chain from +0
 * (mp_perturbation.rs:315).  d: nb x m, x: nb x nrows.  Lets a test stream a factor that does not fit in host memory twice. */
int orc_psfp_centres_rows(const double* Lrows, size_t row0, size_t nrows, size_t m, const double* d, size_t nb, double* x) {
  if (row0 + nrows > m) return ORC_ERR_PARAM;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (size_t r = 0; r < nrows; ++r) {
    const size_t i = row0 + r;
    const double* li = Lrows + (i * (i + 1) / 2 - row0 * (row0 + 1) / 2);
    for (size_t b = 0; b < nb; ++b) {
      const double* db = d + b * m;
      double acc = 0.0;
      for (size_t j = 0; j <= i; ++j) acc = fma(li[j], db[j], acc);
      x[b * nrows + r] = acc;
    }
  }
  return ORC_OK;
}

/* One preimage, every intermediate exposed (mp_perturbation.rs:304-336). */
int orc_psfp_samp_p_trace(const orc_psfp* h, uint64_t seed, uint64_t index, const uint64_t* u,
                          double* d, double* x, int64_t* p, uint64_t* v, int64_t* z, int64_t* e) {
  size_t m = h->m;
  if (!h->L) return ORC_ERR_PARAM;
  /* :315 sample_d_common_non_spherical(sqrt(Sigma_2), r): d <- N(0,1)^m ; x = sqrt(Sigma_2) d ; p_i <- D_{Z,r,x_i} */
  for (size_t j = 0; j < m; ++j) d[j] = orc_sample_normal(seed, index, (uint32_t)j);
  for (size_t i = 0; i < m; ++i) {
    const double* li = h->L + i * (i + 1) / 2;
    double acc = 0.0;
    for (size_t j = 0; j <= i; ++j) acc = fma(li[j], d[j], acc);
    x[i] = acc;
  }
  return orc_psfp_samp_p_from_x(h, seed, index, u, x, p, v, z, e);
}

/* the stages of samp_p after the centres x are known (mp_perturbation.rs:315 rounding, :318, :321-326, :328-335); needs A and R
 * of the key but not sqrt(Sigma_2) */
int orc_psfp_samp_p_from_x(const orc_psfp* h, uint64_t seed, uint64_t index, const uint64_t* u, const double* x,
                           int64_t* p, uint64_t* v, int64_t* z, int64_t* e) {
  const orc_gadget_params* gp = &h->gp;
  size_t n = gp->n, k = gp->k, mb = gp->m_bar, w = n * k, m = h->m;
  uint64_t q = gp->q;
  const unsigned long cap0 = orc_sample_z_cap_hits();
  (void)k;
  for (size_t i = 0; i < m; ++i) p[i] = orc_sample_z(seed, ORC_TAG_PERTURB, index, (uint32_t)i, x[i], h->r);
  /* :318 v = u - A p */
  for (size_t i = 0; i < n; ++i) {
    i128 acc = 0;
    const uint64_t* ai = h->A + i * m;
    for (size_t j = 0; j < m; ++j) acc += (i128)ai[j] * p[j];
    v[i] = submod(u[i] % q, reduce_i128(acc, q), q);
  }
  /* :321-326 */
  int rc = orc_randomized_nearest_plane_gadget(h, seed, index, v, z);
  if (rc) return rc;
  /* :328-335 e = p + [R; I] z */
  for (size_t i = 0; i < mb; ++i) {
    int64_t acc = 0;
    const int8_t* ri = h->R + i * w;
    for (size_t c = 0; c < w; ++c) acc += (int64_t)ri[c] * z[c];
    e[i] = p[i] + acc;
  }
  for (size_t c = 0; c < w; ++c) e[mb + c] = p[mb + c] + z[c];
  return cap_status(cap0, ORC_OK);
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Batch of B independent samp_p calls; identical results to the trace routine (same summation orders), restructured as a
 * cache-blocked triangular product so that one pass over sqrt(Sigma_2), A and R serves a whole group of preimages:
 *   group   = ORC_GRP preimages (one OpenMP task each)
 *   x = L d : rows in panels of ORC_MC, coordinates in chunks of ORC_KC ascending; the accumulators of a panel (MC x GRP) stay in
 *             L2 while the chunks of D (KC x GRP) stream through; micro-tile 4 rows x 32 preimages = 16 vector accumulators.
 *             Every x[i][b] is still ONE fma chain over j = 0..i in ascending order from +0, so the bits are those of the
 *             scalar loop of orc_psfp_samp_p_trace.  The hot loop is compiled for AVX-512 / AVX2+FMA / baseline and
 *             dispatched at load time (target_clones), because the library is built in one container and runs on another host. */
#define ORC_GRP 256
#define ORC_MC 128
#define ORC_KC 256
#define GRP ORC_GRP

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(ORC_NO_CLONES)
#define ORC_CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))
#else
#define ORC_CLONES
#endif

/* acc[r][b] = fma(L[i0 + r][j], D[j][b], acc[r][b]) for j = j0 .. j1-1 (ascending), r < 4, b in [b0, b0 + 32): all four rows lie
 * entirely below the chunk, i.e. j1 <= i0 + 1 */
ORC_CLONES
static void tile_4x32(const double* const lrow[4], size_t j0, size_t j1, const double* D, double* const arow[4], size_t b0) {
  double a0[32], a1[32], a2[32], a3[32];
  for (int b = 0; b < 32; ++b) { a0[b] = arow[0][b0 + b]; a1[b] = arow[1][b0 + b]; a2[b] = arow[2][b0 + b]; a3[b] = arow[3][b0 + b]; }
  for (size_t j = j0; j < j1; ++j) {
    const double* dj = D + j * GRP + b0;
    const double l0 = lrow[0][j], l1 = lrow[1][j], l2 = lrow[2][j], l3 = lrow[3][j];
#pragma GCC ivdep
    for (int b = 0; b < 32; ++b) {
      const double d = dj[b];
      a0[b] = __builtin_fma(l0, d, a0[b]);
      a1[b] = __builtin_fma(l1, d, a1[b]);
      a2[b] = __builtin_fma(l2, d, a2[b]);
      a3[b] = __builtin_fma(l3, d, a3[b]);
    }
  }
  for (int b = 0; b < 32; ++b) { arow[0][b0 + b] = a0[b]; arow[1][b0 + b] = a1[b]; arow[2][b0 + b] = a2[b]; arow[3][b0 + b] = a3[b]; }
}
/* one row against a chunk that reaches its diagonal: j = j0 .. min(j1, i + 1) - 1 */
ORC_CLONES
static void row_chunk(const double* li, size_t j0, size_t jend, const double* D, double* ai) {
  for (size_t j = j0; j < jend; ++j) {
    const double l = li[j];
    const double* dj = D + j * GRP;
#pragma GCC ivdep
    for (int b = 0; b < GRP; ++b) ai[b] = __builtin_fma(l, dj[b], ai[b]);
  }
}

int orc_psfp_samp_p(const orc_psfp* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u,
                    int64_t* e, int nthreads) {
  const orc_gadget_params* gp = &h->gp;
  const size_t n = gp->n, k = gp->k, mb = gp->m_bar, w = n * k, m = h->m;
  const uint64_t q = gp->q;
  const size_t ngroups = (B + GRP - 1) / GRP, npanels = (m + ORC_MC - 1) / ORC_MC;
  int status = ORC_OK;
  const unsigned long cap0 = orc_sample_z_cap_hits();
  if (!h->L) return ORC_ERR_PARAM;
  if (B == 0) return ORC_OK;
  double* norm2 = (double*)malloc(2 * k * sizeof(double));
  gadget_tables(h, norm2, norm2 + k);
  /* all groups at once, so that every phase has (groups x panels) or (groups x preimages) independent tasks for the threads */
  double* Dall = (double*)calloc(ngroups * m * GRP, sizeof(double));
  int32_t* Pall = (int32_t*)calloc(ngroups * m * GRP, sizeof(int32_t));
  int32_t* Zall = (int32_t*)calloc(ngroups * w * GRP, sizeof(int32_t));
  int64_t* maxp_g = (int64_t*)calloc(ngroups, sizeof(int64_t));
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#endif
  /* ---- d <- N(0,1)^m (:315) */
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(dynamic, 64) num_threads(nthreads)
#endif
  for (size_t g = 0; g < ngroups; ++g)
    for (size_t j = 0; j < m; ++j) {
      const size_t b0 = g * GRP, nb = (B - b0 < GRP) ? B - b0 : GRP;
      double* D = Dall + g * m * GRP;
      for (size_t b = 0; b < nb; ++b) D[j * GRP + b] = orc_sample_normal(seed, first_index + b0 + b, (uint32_t)j);
    }
  /* ---- x = sqrt(Sigma_2) d, p_i <- D_{Z,r,x_i} (:315): one task per (group, panel of ORC_MC rows), heavy panels first */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
  for (size_t task = 0; task < ngroups * npanels; ++task) {
    const size_t g = task % ngroups, pn = npanels - 1 - task / ngroups;
    const size_t b0 = g * GRP, nb = (B - b0 < GRP) ? B - b0 : GRP;
    const double* D = Dall + g * m * GRP;
    int32_t* P = Pall + g * m * GRP;
    double* acc = (double*)malloc(ORC_MC * GRP * sizeof(double));
    const size_t i0 = pn * ORC_MC;
    const size_t rows = m - i0 < ORC_MC ? m - i0 : ORC_MC;
    for (size_t t = 0; t < rows * GRP; ++t) acc[t] = 0.0;
    for (size_t j0 = 0; j0 < i0 + rows; j0 += ORC_KC) {              /* chunks in ascending order: the chain of every element stays ascending */
      const size_t j1 = j0 + ORC_KC;
      size_t first_full = 0;
      while (first_full < rows && i0 + first_full + 1 < j1) ++first_full;   /* row i sees the full chunk iff j1 <= i + 1 */
      for (size_t rr = 0; rr < first_full; ++rr) {                   /* rows that cross or end inside the chunk: one by one */
        const size_t i = i0 + rr;
        if (j0 > i) continue;
        const size_t jend = j1 < i + 1 ? j1 : i + 1;
        row_chunk(h->L + i * (i + 1) / 2, j0, jend, D, acc + rr * GRP);
      }
      size_t rr = first_full;
      for (; rr + 4 <= rows; rr += 4) {                              /* the rest: 4 x 32 tiles */
        const double* lrow[4]; double* arow[4];
        for (int t = 0; t < 4; ++t) { const size_t i = i0 + rr + t; lrow[t] = h->L + i * (i + 1) / 2; arow[t] = acc + (rr + t) * GRP; }
        for (size_t bb = 0; bb < GRP; bb += 32) tile_4x32(lrow, j0, j1, D, arow, bb);
      }
      for (; rr < rows; ++rr) {
        const size_t i = i0 + rr;
        row_chunk(h->L + i * (i + 1) / 2, j0, j1, D, acc + rr * GRP);
      }
    }
    int64_t maxp = 0;
    for (size_t rr = 0; rr < rows; ++rr) {
      const size_t i = i0 + rr;
      for (size_t b = 0; b < nb; ++b) {
        int64_t pv = orc_sample_z(seed, ORC_TAG_PERTURB, first_index + b0 + b, (uint32_t)i, acc[rr * GRP + b], h->r);
        P[i * GRP + b] = (int32_t)pv;
        int64_t ap = pv < 0 ? -pv : pv;
        if (ap > maxp) maxp = ap;
      }
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    { if (maxp > maxp_g[g]) maxp_g[g] = maxp; }
    free(acc);
  }
  /* ---- v = u - A p (:318), z <- gadget sampler (:321-326): one task per preimage */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
  for (size_t bg = 0; bg < B; ++bg) {
    const size_t g = bg / GRP, b = bg % GRP;
    const int32_t* P = Pall + g * m * GRP;
    int32_t* Z = Zall + g * w * GRP;
    uint64_t* v = (uint64_t*)malloc(n * sizeof(uint64_t));
    int64_t* zt = (int64_t*)malloc(w * sizeof(int64_t));
    /* int64 chunks when q * max|p| * chunk fits, else 128-bit */
    const int fast = (q < ((uint64_t)1 << 31)) && (maxp_g[g] < ((int64_t)1 << 24));
    for (size_t i = 0; i < n; ++i) {
      const uint64_t* ai = h->A + i * m;
      uint64_t red;
      if (fast) {
        i128 tot = 0;
        for (size_t j0 = 0; j0 < m; j0 += 128) {
          size_t j1 = j0 + 128 < m ? j0 + 128 : m;
          int64_t acc2 = 0;
          for (size_t j = j0; j < j1; ++j) acc2 += (int64_t)ai[j] * P[j * GRP + b];
          tot += acc2;
        }
        red = reduce_i128(tot, q);
      } else {
        i128 acc2 = 0;
        for (size_t j = 0; j < m; ++j) acc2 += (i128)ai[j] * P[j * GRP + b];
        red = reduce_i128(acc2, q);
      }
      v[i] = submod(u[bg * n + i] % q, red, q);
    }
    int rc = gadget_sample_one(h, norm2, norm2 + k, seed, first_index + bg, v, zt);
    if (rc) { status = rc; }
    for (size_t c = 0; c < w; ++c) Z[c * GRP + b] = (int32_t)zt[c];
    free(v); free(zt);
  }
  /* ---- e = p + [R; I] z (:328-335): one task per (group, 64 rows of R) */
  const size_t nrt = (mb + 63) / 64;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
  for (size_t task = 0; task < ngroups * nrt; ++task) {
    const size_t g = task / nrt, i0 = (task % nrt) * 64, i1 = i0 + 64 < mb ? i0 + 64 : mb;
    const size_t b0 = g * GRP, nb = (B - b0 < GRP) ? B - b0 : GRP;
    const int32_t* P = Pall + g * m * GRP;
    const int32_t* Z = Zall + g * w * GRP;
    int32_t racc[GRP];
    for (size_t i = i0; i < i1; ++i) {
      for (int b = 0; b < GRP; ++b) racc[b] = 0;
      const int8_t* ri = h->R + i * w;
      for (size_t c = 0; c < w; ++c) {
        const int32_t rv = ri[c];
        if (!rv) continue;
        const int32_t* zc = Z + c * GRP;
        for (int b = 0; b < GRP; ++b) racc[b] += rv * zc[b];
      }
      for (size_t b = 0; b < nb; ++b) e[(b0 + b) * m + i] = (int64_t)P[i * GRP + b] + racc[b];
    }
    if (task % nrt == 0)
      for (size_t c = 0; c < w; ++c)
        for (size_t b = 0; b < nb; ++b) e[(b0 + b) * m + mb + c] = (int64_t)P[(mb + c) * GRP + b] + Z[c * GRP + b];
  }
  free(Dall); free(Pall); free(Zall); free(maxp_g); free(norm2);
  return cap_status(cap0, status);
}
#undef GRP

/* mp_perturbation.rs:264-267: D_{Z^m, s*r} centred at 0 */
int orc_psfp_samp_d(const orc_psfp* h, uint64_t seed, uint64_t first_index, size_t B, int64_t* e) {
  size_t m = h->m;
  double sr = h->s * h->r;
  const unsigned long cap0 = orc_sample_z_cap_hits();
  for (size_t b = 0; b < B; ++b)
    for (size_t i = 0; i < m; ++i) e[b * m + i] = orc_sample_z(seed, ORC_TAG_SAMPD, first_index + b, (uint32_t)i, 0.0, sr);
  return cap_status(cap0, ORC_OK);
}

/* mp_perturbation.rs:396-402: column vector of length m with ||sigma||^2 <= s^2 m r^2 */
int orc_psfp_check_domain(const orc_psfp* h, size_t B, const int64_t* e, size_t len, uint8_t* ok) {
  size_t m = h->m;
  double bound = ((h->s * h->s) * (double)m) * (h->r * h->r);
  for (size_t b = 0; b < B; ++b) {
    if (len != m) { ok[b] = 0; continue; }
    u128 nn = 0;
    for (size_t i = 0; i < m; ++i) { i128 v = e[b * len + i]; nn += (u128)(v * v); }
    ok[b] = ((double)nn <= bound) ? 1 : 0;
  }
  return ORC_OK;
}

/* mp_perturbation.rs:366-369 */
int orc_psfp_f_a(const orc_psfp* h, size_t B, const int64_t* e, uint64_t* u) {
  size_t n = h->gp.n, m = h->m;
  uint64_t q = h->gp.q;
  int status = ORC_OK;
  for (size_t b = 0; b < B; ++b) {
    uint8_t ok;
    orc_psfp_check_domain(h, 1, e + b * m, m, &ok);
    if (!ok) status = ORC_ERR_DOMAIN;                               /* assert!, :367 */
    for (size_t i = 0; i < n; ++i) {
      i128 acc = 0;
      const uint64_t* ai = h->A + i * m;
      for (size_t j = 0; j < m; ++j) acc += (i128)ai[j] * e[b * m + j];
      u[b * n + i] = reduce_i128(acc, q);
    }
  }
  return status;
}
