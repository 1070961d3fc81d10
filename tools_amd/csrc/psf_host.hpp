// psf_host.hpp -- host-side (C++) mirror of the reference's deterministic gadget helpers.
// These are setup-time functions of the parameters / key only; the
// per-preimage work is in psf_kernels.hpp.  Row-major flat vectors throughout.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../include/psf_mi355x.h"

// Experiment switches.  The release library (libpsf_mi355x.so) reads NO environment variable for them: psf_exp_env() is a constant nullptr there, the
// branches behind it fold away and the kernels that lost their A/B (k_trmm_f64, k_trmm_f64_reg, k_gadget, k_np_walk2, the per-pair k_np_combine8, the
// copy-kernel transport) are not compiled.  `make exp` builds libpsf_mi355x_exp.so with -DPSF_EXPERIMENTS: the same sources with every PSF_* switch alive --
// what the form-comparison tests and the A/B scripts under tools/ load through PSF_LIB.  A switch a USER needs is a field of the params structs or, for
// the one host-side resource knob (PSF_HOST_WORKERS), read and validated in psfp.hip.
#ifdef PSF_EXPERIMENTS
inline const char* psf_exp_env(const char* name) { return std::getenv(name); }
constexpr bool psf_experiments_build = true;
#else
constexpr const char* psf_exp_env(const char*) { return nullptr; }
constexpr bool psf_experiments_build = false;
#endif

namespace psf {

typedef unsigned __int128 u128;
typedef __int128 i128;

inline uint64_t mulmod_u64(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
inline uint64_t submod_u64(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

// smallest e with base^e >= x  (Z::log_ceil as used at gadget_parameters.rs:121-123)
inline uint64_t log_ceil_u64(uint64_t x, uint64_t base) {
  uint64_t e = 0;
  u128 p = 1;
  while (p < x) { p *= base; ++e; }
  return e;
}
// base^k == q ?  (gadget_classical.rs:258, short_basis_classical.rs:80)
inline bool is_power_of_base(uint64_t base, uint64_t k, uint64_t q) {
  u128 p = 1;
  for (uint64_t i = 0; i < k; ++i) { p *= base; if (p > (u128)q) return false; }
  return p == (u128)q;
}
// base^k < q ?  (gadget_classical.rs:170-172)
inline bool gadget_too_short(uint64_t base, uint64_t k, uint64_t q) {
  u128 p = 1;
  for (uint64_t i = 0; i < k; ++i) { p *= base; if (p >= (u128)q) return false; }
  return p < (u128)q;
}

psf_status gadget_params_default(uint64_t n, uint64_t q, psf_gadget_params* out);
psf_status gadget_params_ring_default(uint64_t n, uint64_t q, psf_gadget_params* out);

std::vector<int64_t> gen_gadget_vec(uint64_t k, uint64_t base);
std::vector<uint64_t> gen_gadget_vec_mod(uint64_t k, uint64_t base, uint64_t q);
std::vector<int64_t> gen_gadget_mat(uint64_t n, uint64_t k, uint64_t base);
// S_k (gadget_classical.rs:249-272), k x k
std::vector<int64_t> short_basis_gadget_block(const psf_gadget_params& gp);
// I_n (x) S_k (gadget_classical.rs:273-286)
std::vector<int64_t> short_basis_gadget(const psf_gadget_params& gp);
// Gram-Schmidt on columns, not normalised (MatQ::gso at mp_perturbation.rs:234); also returns ||b~_i||^2
void gso_columns(const std::vector<int64_t>& basis, size_t dim, std::vector<double>& gso, std::vector<double>& norm2);
// digits of one value (gadget_classical.rs:174-180)
void digits_of(uint64_t value, uint64_t q, uint64_t k, uint64_t base, int64_t* out);
// n x n inverse over Z_q (tag.inverse(), short_basis_classical.rs:106); false if singular
bool mat_inverse_mod(const std::vector<uint64_t>& M, size_t n, uint64_t q, std::vector<uint64_t>& inv);
// W with G W = -H^{-1} A [I|0]^t (short_basis_classical.rs:105-110); w x m_bar
psf_status compute_w(const psf_gadget_params& gp, const uint64_t* tag, const uint64_t* A, std::vector<int64_t>& W);
// S_A = [I R; 0 I] [0 I; S' W] (short_basis_classical.rs:54-102); m x m
psf_status gen_short_basis_for_trapdoor(const psf_gadget_params& gp, const uint64_t* tag, const uint64_t* A,
                                        const int8_t* R, std::vector<int64_t>& out);
// MatZq::solve_gaussian_elimination (gpv.rs:153-156) factored once per key: Gauss-Jordan with unit pivots, columns scanned
// left to right, free variables 0.  piv[r] = r-th pivot column, T (n x n) with sol[piv[r]] = (T u)[r].
psf_status solve_precompute(const uint64_t* A, size_t n, size_t m, uint64_t q, std::vector<uint32_t>& piv, std::vector<uint64_t>& T);
// rotation_matrix.rs:41-63 / :85-96
void rot_minus(const int64_t* vec, size_t n, int64_t* out /*n x n*/, size_t ld, size_t col_off);
void rot_minus_matrix(const int64_t* mat, size_t rows, size_t cols, int64_t* out /*rows x rows*cols*/);

// ---- ring variant (gadget_ring.rs, short_basis_ring.rs, gpv_ring.rs); polynomials = n coefficients, constant term first ----
// out = a * b in Z[X]/(X^n+1)
void poly_mul_negacyclic(const int64_t* a, const int64_t* b, size_t n, int64_t* out);
// gen_trapdoor_ring_lwe (gadget_ring.rs:62-81): a = [1 | a_bar | g_j - (a_bar r_j + e_j)] mod (X^n+1, q); a: (k+2) x n
void ring_assemble_a(const psf_gadget_params& gp, const uint64_t* a_bar, const int64_t* r, const int64_t* e, uint64_t* a);
// coefficient embedding of gen_short_basis_for_trapdoor_ring (short_basis_ring.rs:64-166), transposed: row c = column c
psf_status ring_short_basis_t(const psf_gadget_params& gp, const uint64_t* a, const int64_t* r, const int64_t* e, std::vector<int32_t>& basis_t);
// Plan of an (incomplete) negacyclic NTT of Z_q[X]/(X^n+1): levels L = min(v2(q-1) - 1, log2 n), leaf degree d = n >> L,
// zetas[i] = zeta^{bitrev_L(i)} for a primitive 2^(L+1)-th root of unity zeta (Kyber's table for q = 3329, n = 256: L = 7, d = 2).
// ok = false when q is not a prime with v2(q-1) >= 2, n is not a power of two, or q >= 2^31.
struct NttPlan { bool ok = false; uint32_t n = 0, L = 0, d = 0; uint64_t q = 0, inv_scale = 0; std::vector<uint64_t> zetas, zetas_inv; };
NttPlan make_ntt_plan(uint64_t q, uint32_t n);
// The same plan in the form the wave-level kernels of psf_ntt_core.hpp read: which arithmetic (qb = 12 / 14: signed 16-bit Montgomery form for
// q < 2^12 / 2^14; qb = 0: 32-bit Montgomery form), the constants of the reduction, and the zetas (forward [2^L] | inverse [2^L]) multiplied by
// R = 2^16 (centred, as int32 bits) or 2^32; qb = 12 appends the pairs of the dot-product form (psf_ntt_core.hpp, Mod16D).  wave = false when the shape has no wave kernel (n outside 128 ... 1024, leaf degree above 4 or wider than a lane).
struct NttTables { bool wave = false; int logn = 0, ld = 0, qb = 0; uint32_t q = 0; int32_t qinv16 = 0; uint32_t nqinv32 = 0, r2 = 0; std::vector<uint32_t> zetas; };
NttTables make_ntt_tables(const NttPlan& pl);
// 2^-L R^(e+1) mod q in the tables' form: the last multiplication of a product that carries e factors R^-1
uint32_t ntt_final_scale(const NttTables& t, const NttPlan& pl, int e);
// rot^-(iota(a)) (gpv_ring.rs:172-178, rotation_matrix.rs:85-96): n x n(k+2) over Z_q
void ring_embed_a(const uint64_t* a, size_t n, size_t K, uint64_t q, std::vector<uint64_t>& A_emb);

}  // namespace psf
