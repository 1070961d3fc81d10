// psf_host.cpp -- host-side mirror of the reference's deterministic gadget helpers (see psf_host.hpp).
#include <climits>
#include "psf_host.hpp"
#include <algorithm>
#include <cstring>

namespace psf {

// GadgetParameters::init_default (gadget_parameters.rs:113-133)
psf_status gadget_params_default(uint64_t n, uint64_t q, psf_gadget_params* out) {
  if (!out || n < 1 || q <= 1 || q >= (1ull << 62)) return PSF_ERR_PARAM;  // assert at :117, Modulus > 1
  const uint64_t k = log_ceil_u64(q, 2), ln = log_ceil_u64(n, 2);
  *out = psf_gadget_params{n, k, n * k + ln * ln, 2, q};
  return PSF_OK;
}

// GadgetParametersRing::init_default (gadget_parameters.rs:165-185)
psf_status gadget_params_ring_default(uint64_t n, uint64_t q, psf_gadget_params* out) {
  if (!out || n < 1 || q <= 1 || q >= (1ull << 62)) return PSF_ERR_PARAM;
  const uint64_t k = log_ceil_u64(q, 2);
  *out = psf_gadget_params{n, k, k + 2, 2, q};
  return PSF_OK;
}

// gadget_classical.rs:128-136
std::vector<int64_t> gen_gadget_vec(uint64_t k, uint64_t base) {
  std::vector<int64_t> g(k);
  int64_t e = 1;
  for (auto& v : g) { v = e; e *= (int64_t)base; }
  return g;
}
std::vector<uint64_t> gen_gadget_vec_mod(uint64_t k, uint64_t base, uint64_t q) {
  std::vector<uint64_t> g(k);
  uint64_t e = 1 % q;
  for (auto& v : g) { v = e; e = mulmod_u64(e, base % q, q); }
  return g;
}

// gadget_classical.rs:91-107
std::vector<int64_t> gen_gadget_mat(uint64_t n, uint64_t k, uint64_t base) {
  const auto g = gen_gadget_vec(k, base);
  std::vector<int64_t> G(n * n * k, 0);
  for (uint64_t row = 0; row < n; ++row) std::copy(g.begin(), g.end(), G.begin() + row * (n * k) + row * k);
  return G;
}

// gadget_classical.rs:249-272
std::vector<int64_t> short_basis_gadget_block(const psf_gadget_params& gp) {
  const size_t k = gp.k;
  std::vector<int64_t> sk(k * k, 0);
  for (size_t d = 0; d < k; ++d) {
    sk[d * k + d] = (int64_t)gp.base;
    if (d + 1 < k) sk[(d + 1) * k + d] = -1;
  }
  if (!is_power_of_base(gp.base, gp.k, gp.q)) {
    uint64_t rest = gp.q;
    for (size_t row = 0; row < k; ++row) {
      sk[row * k + (k - 1)] = (int64_t)(rest % gp.base);
      rest /= gp.base;
    }
  }
  return sk;
}

// gadget_classical.rs:273-286
std::vector<int64_t> short_basis_gadget(const psf_gadget_params& gp) {
  const size_t k = gp.k, w = gp.n * gp.k;
  const auto sk = short_basis_gadget_block(gp);
  std::vector<int64_t> S(w * w, 0);
  for (size_t blk = 0; blk < gp.n; ++blk)
    for (size_t r = 0; r < k; ++r) std::copy(sk.begin() + r * k, sk.begin() + (r + 1) * k, S.begin() + (blk * k + r) * w + blk * k);
  return S;
}

void gso_columns(const std::vector<int64_t>& basis, size_t dim, std::vector<double>& gso, std::vector<double>& norm2) {
  gso.assign(dim * dim, 0.0);
  norm2.assign(dim, 0.0);
  for (size_t col = 0; col < dim; ++col) {
    for (size_t t = 0; t < dim; ++t) gso[t * dim + col] = (double)basis[t * dim + col];
    for (size_t prev = 0; prev < col; ++prev) {
      double num = 0.0;
      for (size_t t = 0; t < dim; ++t) num = std::fma((double)basis[t * dim + col], gso[t * dim + prev], num);
      const double mu = num / norm2[prev];
      for (size_t t = 0; t < dim; ++t) gso[t * dim + col] = std::fma(-mu, gso[t * dim + prev], gso[t * dim + col]);
    }
    double nn = 0.0;
    for (size_t t = 0; t < dim; ++t) nn = std::fma(gso[t * dim + col], gso[t * dim + col], nn);
    norm2[col] = nn;
  }
}

// gadget_classical.rs:174-180
void digits_of(uint64_t value, uint64_t q, uint64_t k, uint64_t base, int64_t* out) {
  uint64_t rest = value % q;
  for (uint64_t pos = 0; pos < k; ++pos) { out[pos] = (int64_t)(rest % base); rest /= base; }
}

static bool inverse_mod(uint64_t a, uint64_t q, uint64_t* inv) {
  i128 r0 = q, r1 = a % q, t0 = 0, t1 = 1;
  while (r1 != 0) {
    const i128 quo = r0 / r1;
    std::swap(r0, r1); r1 -= quo * r0;  // (r0, r1) <- (r1, r0 - quo r1)
    std::swap(t0, t1); t1 -= quo * t0;
  }
  if (r0 != 1) return false;
  if (t0 < 0) t0 += q;
  *inv = (uint64_t)t0;
  return true;
}

bool mat_inverse_mod(const std::vector<uint64_t>& M, size_t n, uint64_t q, std::vector<uint64_t>& inv) {
  std::vector<uint64_t> a(M);
  for (auto& v : a) v %= q;
  inv.assign(n * n, 0);
  for (size_t d = 0; d < n; ++d) inv[d * n + d] = 1 % q;
  for (size_t col = 0; col < n; ++col) {
    size_t piv = n;
    uint64_t pinv = 0;
    for (size_t r = col; r < n && piv == n; ++r)
      if (inverse_mod(a[r * n + col], q, &pinv)) piv = r;
    if (piv == n) return false;
    for (size_t j = 0; j < n; ++j) { std::swap(a[piv * n + j], a[col * n + j]); std::swap(inv[piv * n + j], inv[col * n + j]); }
    for (size_t j = 0; j < n; ++j) { a[col * n + j] = mulmod_u64(a[col * n + j], pinv, q); inv[col * n + j] = mulmod_u64(inv[col * n + j], pinv, q); }
    for (size_t r = 0; r < n; ++r) {
      const uint64_t f = a[r * n + col];
      if (r == col || f == 0) continue;
      for (size_t j = 0; j < n; ++j) {
        a[r * n + j] = submod_u64(a[r * n + j], mulmod_u64(f, a[col * n + j], q), q);
        inv[r * n + j] = submod_u64(inv[r * n + j], mulmod_u64(f, inv[col * n + j], q), q);
      }
    }
  }
  return true;
}

// short_basis_classical.rs:105-110
psf_status compute_w(const psf_gadget_params& gp, const uint64_t* tag, const uint64_t* A, std::vector<int64_t>& W) {
  const size_t n = gp.n, k = gp.k, mb = gp.m_bar, w = n * k, m = mb + w;
  if (gadget_too_short(gp.base, gp.k, gp.q)) return PSF_ERR_MODULUS;
  std::vector<uint64_t> tinv;
  if (tag) {
    std::vector<uint64_t> t(tag, tag + n * n);
    if (!mat_inverse_mod(t, n, gp.q, tinv)) return PSF_ERR_PARAM;
  }
  W.assign(w * mb, 0);
  std::vector<int64_t> dg(k);
  for (size_t col = 0; col < mb; ++col)
    for (size_t row = 0; row < n; ++row) {
      uint64_t v;
      if (tag) {
        u128 acc = 0;
        for (size_t t = 0; t < n; ++t) acc = (acc + (u128)tinv[row * n + t] * (A[t * m + col] % gp.q)) % gp.q;
        v = (uint64_t)acc;
      } else v = A[row * m + col] % gp.q;
      digits_of(v ? gp.q - v : 0, gp.q, k, gp.base, dg.data());
      for (size_t t = 0; t < k; ++t) W[(row * k + t) * mb + col] = dg[t];
    }
  return PSF_OK;
}

// short_basis_classical.rs:54-102.  sa_r = [0 I; S' W], S' = S with columns reversed iff base^k == q;
// sa_l = [I R; 0 I]; the product is formed blockwise: top = [R S' | I + R W], bottom = [S' | W].
psf_status gen_short_basis_for_trapdoor(const psf_gadget_params& gp, const uint64_t* tag, const uint64_t* A,
                                        const int8_t* R, std::vector<int64_t>& out) {
  const size_t n = gp.n, k = gp.k, mb = gp.m_bar, w = n * k, m = mb + w;
  std::vector<int64_t> W;
  const psf_status rc = compute_w(gp, tag, A, W);
  if (rc != PSF_OK) return rc;
  const auto sk = short_basis_gadget_block(gp);
  const bool reversed = is_power_of_base(gp.base, gp.k, gp.q);
  out.assign(m * m, 0);
  // bottom block rows: [S' | W]
  for (size_t r = 0; r < w; ++r) {
    int64_t* row = out.data() + (mb + r) * m;
    const size_t blk = r / k, rr = r % k;
    for (size_t cc = 0; cc < k; ++cc) {
      const size_t scol = blk * k + cc;                       // column of S = I_n (x) S_k
      const size_t dst = reversed ? (w - 1 - scol) : scol;    // reverse_columns (:80-82)
      row[dst] = sk[rr * k + cc];
    }
    std::copy(W.begin() + r * mb, W.begin() + (r + 1) * mb, row + w);
  }
  // top block rows: [0 | I] + R * bottom
  for (size_t r = 0; r < mb; ++r) {
    int64_t* row = out.data() + r * m;
    row[w + r] = 1;
    for (size_t t = 0; t < w; ++t) {
      const int64_t rv = R[r * w + t];
      if (!rv) continue;
      const int64_t* brow = out.data() + (mb + t) * m;
      for (size_t j = 0; j < m; ++j) row[j] += rv * brow[j];
    }
  }
  return PSF_OK;
}

// x mod q for any 64-bit x and q < 2^32 with one high multiply (the elimination below spends its time in reductions: 200 ms -> 25 ms at n = 256)
struct Barrett64 {
  uint64_t q, m;
  explicit Barrett64(uint64_t q_) : q(q_), m(~0ull / q_) {}
  uint64_t red(uint64_t x) const {
    uint64_t r = x - (uint64_t)(((u128)x * m) >> 64) * q;
    while (r >= q) r -= q;
    return r;
  }
};

psf_status solve_precompute(const uint64_t* A, size_t n, size_t m, uint64_t q, std::vector<uint32_t>& piv, std::vector<uint64_t>& T) {
  T.assign(n * n, 0);
  for (size_t d = 0; d < n; ++d) T[d * n + d] = 1 % q;
  piv.clear();
  std::vector<uint64_t> col(n), acol(n);
  const bool narrow = q < (1ull << 32) && n < (1ull << 32);       // every product below 2^64, sums of n of them below 2^96: the same residues, fewer reductions
  const Barrett64 bq(narrow ? q : 3);
  const bool sum64 = narrow && (u128)n * (q - 1) * (q - 1) < ((u128)1 << 64);
  for (size_t c = 0; c < m && piv.size() < n; ++c) {
    const size_t rank = piv.size();
    if (narrow) {
      for (size_t t = 0; t < n; ++t) acol[t] = A[t * m + c] % q;
      for (size_t r = 0; r < n; ++r) {                    // column c of T * A
        const uint64_t* Tr = &T[r * n];
        if (sum64) {                                      // n (q - 1)^2 < 2^64: the whole sum in one word (vectorised by the compiler)
          uint64_t acc = 0;
          for (size_t t = 0; t < n; ++t) acc += Tr[t] * acol[t];
          col[r] = bq.red(acc);
        } else {
          u128 acc = 0;
          for (size_t t = 0; t < n; ++t) acc += (u128)(Tr[t] * acol[t]);
          col[r] = (uint64_t)(acc % q);
        }
      }
    } else {
      for (size_t r = 0; r < n; ++r) {
        u128 acc = 0;
        for (size_t t = 0; t < n; ++t) acc = (acc + (u128)T[r * n + t] * (A[t * m + c] % q)) % q;
        col[r] = (uint64_t)acc;
      }
    }
    size_t p = n;
    uint64_t pinv = 0;
    for (size_t r = rank; r < n && p == n; ++r)
      if (inverse_mod(col[r], q, &pinv)) p = r;
    if (p == n) continue;                                 // no unit in this column: free variable
    if (p != rank) {
      for (size_t j = 0; j < n; ++j) std::swap(T[p * n + j], T[rank * n + j]);
      std::swap(col[p], col[rank]);
    }
    uint64_t* Tp = &T[rank * n];
    if (narrow) for (size_t j = 0; j < n; ++j) Tp[j] = bq.red(Tp[j] * pinv);
    else for (size_t j = 0; j < n; ++j) Tp[j] = mulmod_u64(Tp[j], pinv, q);
    for (size_t r = 0; r < n; ++r) {
      if (r == rank || col[r] == 0) continue;
      const uint64_t f = col[r];
      uint64_t* Tr = &T[r * n];
      if (narrow) {
        const uint64_t g = q - f;                          // T[r] - f T[rank] = T[r] + (q - f) T[rank]: below q + q^2 < 2^64
        for (size_t j = 0; j < n; ++j) Tr[j] = bq.red(Tr[j] + g * Tp[j]);
      } else {
        for (size_t j = 0; j < n; ++j) Tr[j] = submod_u64(Tr[j], mulmod_u64(f, Tp[j], q), q);
      }
    }
    piv.push_back((uint32_t)c);
  }
  return piv.size() == n ? PSF_OK : PSF_ERR_NO_SOLUTION;
}

// rotation_matrix.rs:41-63: column j of rot^-(v) is v multiplied by X^j in Z[X]/(X^n+1)
void rot_minus(const int64_t* vec, size_t n, int64_t* out, size_t ld, size_t col_off) {
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < n; ++j) {
      const size_t pos = i + j;
      if (pos >= n) out[(pos - n) * ld + col_off + j] = -vec[i];
      else out[pos * ld + col_off + j] = vec[i];
    }
}
// rotation_matrix.rs:85-96: [rot^-(col_0) | rot^-(col_1) | ...]
void rot_minus_matrix(const int64_t* mat, size_t rows, size_t cols, int64_t* out) {
  std::vector<int64_t> col(rows);
  for (size_t c = 0; c < cols; ++c) {
    for (size_t r = 0; r < rows; ++r) col[r] = mat[r * cols + c];
    rot_minus(col.data(), rows, out, rows * cols, c * rows);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ring variant
// ---------------------------------------------------------------------------------------------------------------------
void poly_mul_negacyclic(const int64_t* a, const int64_t* b, size_t n, int64_t* out) {
  std::vector<int64_t> t(n, 0);
  for (size_t i = 0; i < n; ++i) {
    if (a[i] == 0) continue;
    for (size_t j = 0; j < n; ++j) {
      const size_t d = i + j;
      if (d >= n) t[d - n] -= a[i] * b[j];
      else t[d] += a[i] * b[j];
    }
  }
  std::copy(t.begin(), t.end(), out);
}

// x * X^shift in Z[X]/(X^n+1)
static void poly_shift_negacyclic(const int64_t* x, size_t n, size_t shift, int64_t* out) {
  for (size_t j = 0; j < n; ++j) {
    const size_t d = j + shift;
    if (d >= n) out[d - n] = -x[j];
    else out[d] = x[j];
  }
}

// gadget_ring.rs:62-81
void ring_assemble_a(const psf_gadget_params& gp, const uint64_t* a_bar, const int64_t* r, const int64_t* e, uint64_t* a) {
  const size_t n = gp.n, k = gp.k;
  const uint64_t q = gp.q;
  std::fill(a, a + (k + 2) * n, 0);
  a[0] = 1 % q;                                                          // :74
  for (size_t c = 0; c < n; ++c) a[n + c] = a_bar[c] % q;                // :75
  uint64_t g = 1 % q;
  std::vector<i128> acc(n);
  for (size_t j = 0; j < k; ++j) {                                       // :76-78  g^t - (a_bar r + e)
    std::fill(acc.begin(), acc.end(), (i128)0);
    for (size_t x = 0; x < n; ++x)
      for (size_t y = 0; y < n; ++y) {
        const i128 p = (i128)(int64_t)(a_bar[x] % q) * r[j * n + y];
        const size_t d = x + y;
        if (d >= n) acc[d - n] -= p;
        else acc[d] += p;
      }
    for (size_t c = 0; c < n; ++c) {
      i128 v = (c == 0 ? (i128)g : (i128)0) - (acc[c] + e[j * n + c]);
      v %= (i128)q;
      if (v < 0) v += q;
      a[(2 + j) * n + c] = (uint64_t)v;
    }
    g = mulmod_u64(g, gp.base % q, q);
  }
}

// short_basis_ring.rs:64-166 in closed form.  With pd = [X^0 .. X^{n-1}], sa_l = [1 0 e; 0 1 r; 0 0 I_k] and
// sa_r = [pd (x) [0; S'] | pd (x) [I_2; W]], column (i, c) of the product is X^i times
//   left  block (c < k)  : ( sum_t S'[t][c] e_t , sum_t S'[t][c] r_t , S'[:, c] )
//   right block (c in 0,1): ( [c=0] + sum_t e_t W[t][c] , [c=1] + sum_t r_t W[t][c] , W[:, c] )
// reduced mod X^n+1 (:72-77).  W[:, c] = digits of -a_c per coefficient (compute_w :128-139, gadget_ring.rs:145-166).
psf_status ring_short_basis_t(const psf_gadget_params& gp, const uint64_t* a, const int64_t* r, const int64_t* e, std::vector<int32_t>& basis_t) {
  const size_t n = gp.n, k = gp.k, K = k + 2, d = n * K;
  const uint64_t q = gp.q;
  if (gadget_too_short(gp.base, gp.k, gp.q)) return PSF_ERR_MODULUS;
  const auto sk = short_basis_gadget_block(gp);                          // compute_s, :142-166
  const bool reversed = is_power_of_base(gp.base, gp.k, gp.q);           // :110-112
  // columns before the X^i shift: K polynomials each
  std::vector<int64_t> W(2 * k * n), dg(k);
  for (size_t c = 0; c < 2; ++c)
    for (size_t j = 0; j < n; ++j) {
      const uint64_t v = a[c * n + j] % q;
      digits_of(v ? q - v : 0, q, k, gp.base, dg.data());
      for (size_t t = 0; t < k; ++t) W[(c * k + t) * n + j] = dg[t];
    }
  basis_t.assign(d * d, 0);
  std::vector<int64_t> col(K * n), prod(n), shifted(n);
  bool overflow = false;
  auto emit = [&](size_t column, size_t shift) {
    for (size_t row = 0; row < K; ++row) {
      poly_shift_negacyclic(col.data() + row * n, n, shift, shifted.data());
      for (size_t c = 0; c < n; ++c) {
        if (shifted[c] > INT32_MAX || shifted[c] < -INT32_MAX) overflow = true;     // caller-supplied (r, e) of unusual size: the basis is held in int32
        basis_t[column * d + row * n + c] = (int32_t)shifted[c];
      }
    }
  };
  for (size_t c = 0; c < k; ++c) {                                       // left block
    std::fill(col.begin(), col.end(), 0);
    const size_t sc = reversed ? (k - 1 - c) : c;
    for (size_t t = 0; t < k; ++t) {
      const int64_t v = sk[t * k + sc];
      if (v == 0) continue;
      for (size_t x = 0; x < n; ++x) { col[x] += v * e[t * n + x]; col[n + x] += v * r[t * n + x]; }
      col[(2 + t) * n] = v;
    }
    for (size_t i = 0; i < n; ++i) emit(i * k + c, i);
  }
  for (size_t c = 0; c < 2; ++c) {                                       // right block
    std::fill(col.begin(), col.end(), 0);
    col[c * n] = 1;
    for (size_t t = 0; t < k; ++t) {
      const int64_t* w = W.data() + (c * k + t) * n;
      poly_mul_negacyclic(e + t * n, w, n, prod.data());
      for (size_t x = 0; x < n; ++x) col[x] += prod[x];
      poly_mul_negacyclic(r + t * n, w, n, prod.data());
      for (size_t x = 0; x < n; ++x) col[n + x] += prod[x];
      std::copy(w, w + n, col.begin() + (2 + t) * n);
    }
    for (size_t i = 0; i < n; ++i) emit(k * n + 2 * i + c, i);
  }
  if (overflow) return PSF_ERR_UNSUPPORTED;
  return PSF_OK;
}

static uint64_t powmod_u64(uint64_t b, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  b %= q;
  while (e) { if (e & 1) r = mulmod_u64(r, b, q); b = mulmod_u64(b, b, q); e >>= 1; }
  return r;
}
static bool is_prime_u64(uint64_t q) {
  if (q < 2) return false;
  for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) { if (q % p == 0) return q == p; }
  uint64_t dd = q - 1; int r = 0;
  while ((dd & 1) == 0) { dd >>= 1; ++r; }
  for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {   // deterministic below 2^64
    uint64_t x = powmod_u64(a, dd, q);
    if (x == 1 || x == q - 1) continue;
    bool comp = true;
    for (int i = 1; i < r && comp; ++i) { x = mulmod_u64(x, x, q); if (x == q - 1) comp = false; }
    if (comp) return false;
  }
  return true;
}

NttPlan make_ntt_plan(uint64_t q, uint32_t n) {
  NttPlan pl;
  pl.q = q; pl.n = n;
  if (n < 2 || (n & (n - 1)) || q >= (1ull << 31) || !is_prime_u64(q)) return pl;
  uint32_t t = 0;
  for (uint64_t x = q - 1; (x & 1) == 0; x >>= 1) ++t;           // t = v2(q - 1): primitive 2^t-th roots of unity exist
  if (t < 2) return pl;
  uint32_t log_n = 0;
  while ((1u << log_n) < n) ++log_n;
  const uint32_t L = (t - 1 < log_n) ? t - 1 : log_n;
  // zeta: element of order exactly 2^(L+1):  x^((q-1)/2^(L+1)) for a quadratic non-residue x
  uint64_t zeta = 0;
  for (uint64_t x = 2; x < q; ++x) {
    if (powmod_u64(x, (q - 1) / 2, q) != q - 1) continue;         // need a non-residue so that the order is full
    zeta = powmod_u64(x, (q - 1) >> (L + 1), q);
    break;
  }
  if (zeta == 0) return pl;
  pl.L = L; pl.d = n >> L;
  const uint32_t cnt = 1u << L;
  pl.zetas.assign(cnt, 0); pl.zetas_inv.assign(cnt, 0);
  for (uint32_t i = 0; i < cnt; ++i) {
    uint32_t br = 0;
    for (uint32_t b = 0; b < L; ++b) if (i & (1u << b)) br |= 1u << (L - 1 - b);
    pl.zetas[i] = powmod_u64(zeta, br, q);
    pl.zetas_inv[i] = powmod_u64(pl.zetas[i], q - 2, q);
  }
  pl.inv_scale = powmod_u64(cnt % q, q - 2, q);                    // 2^-L
  pl.ok = true;
  return pl;
}

static uint32_t ntt_form(uint64_t v, uint64_t q, int qb) {          // centred int32 bits for the 16-bit form, canonical otherwise
  if (qb == 0) return (uint32_t)v;
  const int64_t c = v > q / 2 ? (int64_t)v - (int64_t)q : (int64_t)v;
  return (uint32_t)(int32_t)c;
}
NttTables make_ntt_tables(const NttPlan& pl) {
  NttTables t;
  if (!pl.ok) return t;
  const uint64_t q = pl.q;
  t.q = (uint32_t)q;
  while ((1u << t.logn) < pl.n) ++t.logn;
  while ((1u << t.ld) < pl.d) ++t.ld;
  // the 12-bit bound analysis (fewer uniform reductions) is instantiated for the seven-level shapes of q = 3329 (n = 128, 256, 512) only
  t.wave = t.logn >= 7 && t.logn <= 10 && t.ld <= 2 && t.ld <= t.logn - 6 && pl.L >= 1;
  t.qb = !t.wave ? 0 : q < (1ull << 12) && pl.L == 7 && t.ld == t.logn - 7 ? 12 : q < (1ull << 14) ? 14 : 0;   // the generic LDS kernel computes in the 32-bit form
  uint32_t inv = 1;                                                   // Newton: q^-1 mod 2^32 (q odd)
  for (int i = 0; i < 5; ++i) inv *= 2u - (uint32_t)q * inv;
  t.nqinv32 = 0u - inv;
  t.qinv16 = (int32_t)(int16_t)(uint16_t)inv;
  const uint64_t R = (t.qb ? (1ull << 16) : (1ull << 32)) % q;
  t.r2 = ntt_form(mulmod_u64(R, R, q), q, t.qb);
  const size_t cnt = pl.zetas.size();
  t.zetas.resize(2 * cnt);
  for (size_t i = 0; i < cnt; ++i) {
    t.zetas[i] = ntt_form(mulmod_u64(pl.zetas[i], R, q), q, t.qb);
    t.zetas[cnt + i] = ntt_form(mulmod_u64(pl.zetas_inv[i], R, q), q, t.qb);
  }
  if (t.qb == 12) {                                                   // dot-product form of the forward butterflies: [(z | -q), z q^-1 mod 2^16] per zeta
    t.zetas.resize(4 * cnt);
    for (size_t i = 0; i < cnt; ++i) {
      const uint32_t z = t.zetas[i] & 0xffffu;
      t.zetas[2 * cnt + 2 * i] = z | ((uint32_t)(0x10000u - (uint32_t)q) << 16);
      t.zetas[2 * cnt + 2 * i + 1] = (z * (uint32_t)(uint16_t)t.qinv16) & 0xffffu;
    }
  }
  return t;
}
uint32_t ntt_final_scale(const NttTables& t, const NttPlan& pl, int e) {
  const uint64_t q = pl.q, R = (t.qb ? (1ull << 16) : (1ull << 32)) % q;
  uint64_t f = pl.inv_scale % q;
  for (int i = 0; i <= e; ++i) f = mulmod_u64(f, R, q);
  return ntt_form(f, q, t.qb);
}

// gpv_ring.rs:172-178
void ring_embed_a(const uint64_t* a, size_t n, size_t K, uint64_t q, std::vector<uint64_t>& A_emb) {
  const size_t d = n * K;
  A_emb.assign(n * d, 0);
  for (size_t j = 0; j < K; ++j)
    for (size_t i = 0; i < n; ++i) {
      const uint64_t v = a[j * n + i] % q;
      for (size_t l = 0; l < n; ++l) {
        const size_t row = i + l;
        if (row >= n) A_emb[(row - n) * d + j * n + l] = v ? q - v : 0;
        else A_emb[row * d + j * n + l] = v;
      }
    }
}

}  // namespace psf
