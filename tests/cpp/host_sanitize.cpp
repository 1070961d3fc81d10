// Host-side logic of the library (tools_amd/csrc/psf_host.cpp: parameters, gadget helpers, short bases, elimination, ring
// embedding, NTT plan) exercised under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU, with the invariants the
// reference's tests use (short_basis_classical.rs:128-188, gadget_classical.rs:363-414).  No GPU, no HIP: built with g++.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tools_amd/csrc/psf_host.hpp"

using namespace psf;
typedef unsigned __int128 u128x;

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); std::exit(1); } } while (0)

static void classical(uint64_t n, uint64_t q) {
  psf_gadget_params gp;
  CHECK(gadget_params_default(n, q, &gp) == PSF_OK);
  const size_t k = gp.k, w = n * k, mb = gp.m_bar, m = mb + w;
  const auto g = gen_gadget_vec(k, gp.base);
  const auto G = gen_gadget_mat(n, k, gp.base);
  CHECK(g.size() == k && G.size() == n * w);
  // digits: <g, digits(v)> = v
  for (int t = 0; t < 50; ++t) {
    const uint64_t v = rnd() % q;
    std::vector<int64_t> d(k);
    digits_of(v, q, k, gp.base, d.data());
    u128x acc = 0;
    for (size_t i = 0; i < k; ++i) acc += (u128x)(uint64_t)g[i] * (uint64_t)d[i];
    CHECK((uint64_t)(acc % q) == v % q);
  }
  // S_k is a basis of the q-ary gadget lattice: g^t S_k = 0 mod q
  const auto Sk = short_basis_gadget_block(gp);
  for (size_t c = 0; c < k; ++c) {
    __int128 acc = 0;
    for (size_t r = 0; r < k; ++r) acc += (__int128)g[r] * Sk[r * k + c];
    CHECK((uint64_t)(((acc % (__int128)q) + q) % q) == 0);
  }
  std::vector<double> gso, norm2;
  gso_columns(Sk, k, gso, norm2);
  for (double v : norm2) CHECK(v > 0.0);
  CHECK(short_basis_gadget(gp).size() == w * w);
  // trapdoor A = [A_bar | G - A_bar R], short basis S_A: A S_A = 0 mod q
  std::vector<uint64_t> A(n * m);
  std::vector<int8_t> R(mb * w);
  for (auto& v : R) v = (int8_t)((int)(rnd() % 3) - 1);
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < mb; ++j) A[i * m + j] = rnd() % q;
  for (size_t i = 0; i < n; ++i)
    for (size_t c = 0; c < w; ++c) {
      __int128 acc = G[i * w + c];
      for (size_t t = 0; t < mb; ++t) acc -= (__int128)A[i * m + t] * R[t * w + c];
      A[i * m + mb + c] = (uint64_t)(((acc % (__int128)q) + q) % q);
    }
  std::vector<int64_t> S;
  CHECK(gen_short_basis_for_trapdoor(gp, nullptr, A.data(), R.data(), S) == PSF_OK);
  CHECK(S.size() == m * m);
  for (size_t i = 0; i < n; ++i)
    for (size_t c = 0; c < m; ++c) {
      __int128 acc = 0;
      for (size_t t = 0; t < m; ++t) acc += (__int128)A[i * m + t] * S[t * m + c];
      CHECK((uint64_t)(((acc % (__int128)q) + q) % q) == 0);
    }
  // elimination: A sol = u with sol supported on the pivots
  std::vector<uint32_t> piv;
  std::vector<uint64_t> T;
  const psf_status rc = solve_precompute(A.data(), n, m, q, piv, T);
  if (rc == PSF_OK) {
    std::vector<uint64_t> u(n), sol(m, 0);
    for (auto& v : u) v = rnd() % q;
    for (size_t r = 0; r < n; ++r) {
      u128x acc = 0;
      for (size_t t = 0; t < n; ++t) acc = (acc + (u128x)T[r * n + t] * u[t]) % q;
      sol[piv[r]] = (uint64_t)acc;
    }
    for (size_t i = 0; i < n; ++i) {
      u128x acc = 0;
      for (size_t t = 0; t < m; ++t) acc = (acc + (u128x)A[i * m + t] * sol[t]) % q;
      CHECK((uint64_t)acc == u[i]);
    }
  } else {
    CHECK(rc == PSF_ERR_NO_SOLUTION);
  }
  // tag inverse
  std::vector<uint64_t> M(n * n), inv;
  for (auto& v : M) v = rnd() % q;
  if (mat_inverse_mod(M, n, q, inv)) {
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j) {
        u128x acc = 0;
        for (size_t t = 0; t < n; ++t) acc = (acc + (u128x)M[i * n + t] * inv[t * n + j]) % q;
        CHECK((uint64_t)acc == (i == j ? 1 % q : 0));
      }
  }
}

static void ring(uint64_t n, uint64_t q) {
  psf_gadget_params gp;
  CHECK(gadget_params_ring_default(n, q, &gp) == PSF_OK);
  const size_t k = gp.k, K = k + 2;
  std::vector<uint64_t> a_bar(n), a(K * n);
  std::vector<int64_t> r(k * n), e(k * n);
  for (auto& v : a_bar) v = rnd() % q;
  for (auto& v : r) v = (int64_t)(rnd() % 7) - 3;
  for (auto& v : e) v = (int64_t)(rnd() % 7) - 3;
  ring_assemble_a(gp, a_bar.data(), r.data(), e.data(), a.data());
  std::vector<int32_t> bt;
  CHECK(ring_short_basis_t(gp, a.data(), r.data(), e.data(), bt) == PSF_OK);
  const size_t d = K * n;
  CHECK(bt.size() == d * d);
  std::vector<uint64_t> A_emb;
  ring_embed_a(a.data(), n, K, q, A_emb);
  CHECK(A_emb.size() == n * d);
  for (size_t i = 0; i < n; ++i)                    // rot^-(iota(a)) * basis = 0 mod q (rows of bt are basis vectors)
    for (size_t c = 0; c < d; ++c) {
      __int128 acc = 0;
      for (size_t t = 0; t < d; ++t) acc += (__int128)A_emb[i * d + t] * bt[c * d + t];
      CHECK((uint64_t)(((acc % (__int128)q) + q) % q) == 0);
    }
  std::vector<int64_t> x(n), y(n), z(n), rot(n * n);
  for (auto& v : x) v = (int64_t)(rnd() % 200) - 100;
  for (auto& v : y) v = (int64_t)(rnd() % 200) - 100;
  poly_mul_negacyclic(x.data(), y.data(), n, z.data());
  rot_minus(x.data(), n, rot.data(), n, 0);
  for (size_t i = 0; i < n; ++i) {                  // rot^-(x) y = x * y
    int64_t acc = 0;
    for (size_t t = 0; t < n; ++t) acc += rot[i * n + t] * y[t];
    CHECK(acc == z[i]);
  }
  std::vector<int64_t> mat(2 * 3), out(2 * 2 * 3);
  for (auto& v : mat) v = (int64_t)(rnd() % 9) - 4;
  rot_minus_matrix(mat.data(), 2, 3, out.data());
  const NttPlan plan = make_ntt_plan(q, (uint32_t)n);
  if (plan.ok) { CHECK(plan.n == n && (plan.d << plan.L) == n && plan.zetas.size() >= ((size_t)1 << plan.L)); }
}

int main() {
  const uint64_t cases[][2] = {{2, 8}, {3, 125}, {4, 23}, {5, 256}, {6, 128}, {3, 1073741789ull}, {2, (1ull << 60)}, {2, (1ull << 61) - 1}};
  for (auto& c : cases) classical(c[0], c[1]);
  const uint64_t rcases[][2] = {{4, 16}, {8, 17}, {16, 3329}, {8, 257}, {4, 1073741789ull}};
  for (auto& c : rcases) ring(c[0], c[1]);
  psf_gadget_params gp;
  CHECK(gadget_params_default(0, 8, &gp) != PSF_OK);
  CHECK(gadget_params_default(4, 1, &gp) != PSF_OK);
  std::printf("HOST_SANITIZE_OK\n");
  return 0;
}
