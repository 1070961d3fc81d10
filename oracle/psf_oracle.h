/*
 * psf_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into the product).
 *
 * A plain-C restatement of the reference's preimage-sampling algorithms
 * (qfall/tools: src/primitive/psf/ and src/sample/g_trapdoor/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - deterministic helpers (gadget vec/mat, digit decomposition, short bases, rot^-,
 *     default parameters) are PINNED against the reference's own known-answer tests,
 *     transcribed in tests/golden/ref_kats.json.
 *   - everything that consumes randomness is "PARITY UNPINNED": the reference's PSF trait
 *     has no seed argument (psf.rs:48-80) and its RNG lives in the un-vendored qfall-math
 *     crate (Cargo.toml:18, version "0", no lockfile), so no golden samp_p vector exists or
 *     can be generated in this environment.  For those the oracle defines the randomness
 *     contract (Philox4x32-10 streams, rejection SampleZ of GPV08 as documented at
 *     CONTRIBUTING.md:35-45) and the HIP path is bit-exact against THIS file.
 */
#ifndef PSF_ORACLE_H
#define PSF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (mirrors include/psf_mi355x.h) */
enum {
  ORC_OK = 0,
  ORC_ERR_PARAM = 1,
  ORC_ERR_NOT_PD = 2,       /* mp_perturbation.rs:109-110 "panics if Sigma_2 is not positive definite" */
  ORC_ERR_DOMAIN = 3,       /* f_a on sigma outside D_n: mp_perturbation.rs:367 */
  ORC_ERR_MODULUS = 4,      /* gadget_classical.rs:170-172 base^k < q */
  ORC_ERR_NO_SOLUTION = 5,  /* solve_gaussian_elimination -> None -> unwrap panic (gpv.rs:153-155) */
  ORC_ERR_SAMPLER = 6       /* a value left the range the restatement covers (mirrors PSF_ERR_SAMPLER of the product) */
};

/* randomness streams (c3 tag of the Philox counter) */
enum {
  ORC_TAG_ABAR = 1, ORC_TAG_R = 2, ORC_TAG_NORMAL = 3, ORC_TAG_PERTURB = 4,
  ORC_TAG_GADGET = 5, ORC_TAG_SAMPD = 6, ORC_TAG_TARGET = 7, ORC_TAG_GPV = 8,
  ORC_TAG_RING_R = 9, ORC_TAG_RING_E = 10, ORC_TAG_RING_A = 11, ORC_TAG_GPV2 = 12
};

/* ---- primitives of the randomness contract ---- */
void   orc_philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
double orc_det_exp(double y);
/* D_{Z,s,c} by rejection from [ceil(c)-ceil(6s), floor(c)+floor(6s)] (CONTRIBUTING.md:35-45) */
int64_t orc_sample_z(uint64_t seed, uint32_t tag, uint64_t index, uint32_t coord, double center, double s);
/* how many draws have ended at the attempt cap since the library was loaded (see orc_sample_z) */
unsigned long orc_sample_z_cap_hits(void);
double  orc_sample_normal(uint64_t seed, uint64_t index, uint32_t coord);
uint64_t orc_uniform_mod(uint64_t seed, uint32_t tag, uint32_t c0, uint32_t c1, uint64_t q);

/* ---- gadget_parameters.rs:113-133 / :165-185 ---- */
typedef struct { uint64_t n, k, m_bar, base, q; } orc_gadget_params;
int orc_gadget_params_default(uint64_t n, uint64_t q, orc_gadget_params* gp);
int orc_gadget_params_ring_default(uint64_t n, uint64_t q, orc_gadget_params* gp);

/* ---- gadget_classical.rs ---- */
int orc_gen_gadget_vec(uint64_t k, uint64_t base, int64_t* out /*k*/);                  /* :128-136 */
int orc_gen_gadget_mat(uint64_t n, uint64_t k, uint64_t base, int64_t* out /*n x nk*/); /* :91-107  */
int orc_find_solution_gadget_vec(uint64_t value, uint64_t q, uint64_t k, uint64_t base, int64_t* out /*k*/); /* :169-182 */
int orc_find_solution_gadget_mat(const uint64_t* value /*rows x cols*/, size_t rows, size_t cols,
                                 uint64_t q, uint64_t k, uint64_t base, int64_t* out /*k*rows x cols*/); /* :219-229 */
int orc_short_basis_gadget_block(const orc_gadget_params* gp, int64_t* sk /*k x k*/);  /* :249-272 (S_k) */
int orc_short_basis_gadget(const orc_gadget_params* gp, int64_t* out /*nk x nk*/);      /* :248-287 */
/* column-wise Gram-Schmidt in f64 (qfall-math MatQ::gso, used at mp_perturbation.rs:234) */
void orc_gso_columns(const int64_t* basis, size_t dim, double* gso /*dim x dim*/);

/* trapdoor_distribution.rs:82-86 ; gadget_classical.rs:56-68 */
void orc_sample_r(uint64_t seed, size_t m_bar, size_t w, int8_t* R /*m_bar x w*/);
void orc_sample_a_bar(uint64_t seed, size_t n, size_t m_bar, uint64_t q, uint64_t* a_bar /*n x m_bar*/);
int  orc_gen_trapdoor(const orc_gadget_params* gp, const uint64_t* a_bar, const uint64_t* tag /*n x n or NULL=I*/,
                      const int8_t* R, uint64_t* A /*n x (m_bar+nk)*/);

/* short_basis_classical.rs:54-110 */
int orc_gen_sa_l(const int8_t* R, size_t m_bar, size_t w, int64_t* out /*(m_bar+w)^2*/);
int orc_gen_sa_r(const orc_gadget_params* gp, const uint64_t* tag, const uint64_t* A, int64_t* out /*(m_bar+w)^2*/);
int orc_compute_w(const orc_gadget_params* gp, const uint64_t* tag, const uint64_t* A, int64_t* W /*w x m_bar*/);
int orc_gen_short_basis_for_trapdoor(const orc_gadget_params* gp, const uint64_t* tag, const uint64_t* A,
                                     const int8_t* R, int64_t* out /*m x m*/);

/* ---- PSFPerturbation (mp_perturbation.rs) ---- */
typedef struct {
  orc_gadget_params gp;
  double r, s;
  size_t m;              /* m_bar + n*k */
  uint64_t* A;           /* n x m row-major */
  int8_t* R;             /* m_bar x w */
  double* L;             /* sqrt(Sigma_2): lower-triangular, packed by rows, m(m+1)/2 */
  int64_t* Sk;           /* k x k gadget basis block */
  double* Sk_gso;        /* k x k GSO (columns) */
} orc_psfp;

orc_psfp* orc_psfp_new(const orc_gadget_params* gp, double r, double s);
orc_psfp* orc_psfp_new_nokey(const orc_gadget_params* gp, double r, double s);   /* L == NULL: see orc_psfp_centres_rows */
void      orc_psfp_free(orc_psfp*);
/* mp_perturbation.rs:221-244 */
int orc_psfp_trap_gen(orc_psfp*, uint64_t seed);
/* install externally produced key material (copied) */
int orc_psfp_load_key(orc_psfp*, const uint64_t* A, const int8_t* R, const double* L_packed);
/* mp_perturbation.rs:111-139 ; sigma given as scalar*I (the only form trap_gen uses) */
int orc_psfp_compute_sqrt_sigma_2(const orc_psfp*, const int8_t* R, double s_cov, double* L_packed);
int orc_psfp_compute_sqrt_sigma_2_dense(const orc_psfp*, const int8_t* R, const double* sigma_packed, double* L_packed);
/* rows 0..m0-1 of the same factor (= the factor of the leading m0 x m0 block of Sigma_2); Lp: m0(m0+1)/2 */
int orc_psfp_sqrt_sigma_2_leading(const orc_psfp*, const int8_t* R, double s_cov, size_t m0, double* L_packed);
/* x = sqrt(Sigma_2) d restricted to a packed row block of the factor (streamed verification of large keys) */
int orc_psfp_centres_rows(const double* Lrows, size_t row0, size_t nrows, size_t m, const double* d /*nb x m*/, size_t nb,
                          double* x /*nb x nrows*/);
/* samp_p from the centres on: p, v, z, e of one preimage (uses A, R; not L) */
int orc_psfp_samp_p_from_x(const orc_psfp*, uint64_t seed, uint64_t index, const uint64_t* u, const double* x,
                           int64_t* p, uint64_t* v, int64_t* z, int64_t* e);
/* structured square root of Sigma_2 (the product's labelled opt-in; contract in psf_oracle.c) */
int orc_psfp_structured_sqrt(const orc_psfp*, const int8_t* R, double s_cov, double* L1_packed /*m_bar(m_bar+1)/2*/);
int orc_psfp_samp_p_structured_trace(const orc_psfp*, const double* L1_packed, double s_cov, uint64_t seed, uint64_t index, const uint64_t* u,
                                     double* d, double* x, int64_t* p, uint64_t* v, int64_t* z, int64_t* e);
/* mp_perturbation.rs:304-336, B independent calls; u: B x n, e: B x m ; nthreads<=0 -> all cores */
int orc_psfp_samp_p(const orc_psfp*, uint64_t seed, uint64_t first_index, size_t B,
                    const uint64_t* u, int64_t* e, int nthreads);
/* intermediate values of one preimage, for stage-by-stage parity of the HIP kernels */
int orc_psfp_samp_p_trace(const orc_psfp*, uint64_t seed, uint64_t index, const uint64_t* u,
                          double* d /*m*/, double* x /*m*/, int64_t* p /*m*/, uint64_t* v /*n*/,
                          int64_t* z /*w*/, int64_t* e /*m*/);
int orc_psfp_samp_d(const orc_psfp*, uint64_t seed, uint64_t first_index, size_t B, int64_t* e); /* :264-267 */
int orc_psfp_f_a(const orc_psfp*, size_t B, const int64_t* e, uint64_t* u);                      /* :366-369 */
int orc_psfp_check_domain(const orc_psfp*, size_t B, const int64_t* e, size_t len, uint8_t* ok);  /* :396-402 */
/* mp_perturbation.rs:173-191 for one target vector v (n) -> z (nk) */
int orc_randomized_nearest_plane_gadget(const orc_psfp*, uint64_t seed, uint64_t index, const uint64_t* v, int64_t* z);

int orc_num_threads(void);

/* ---- PSFGPV (gpv.rs) and shared pieces: see psf_oracle_gpv.c ---- */
int orc_solve_gaussian_elimination(const uint64_t* A, size_t n, size_t m, uint64_t q, const uint64_t* u, uint64_t* sol);
int orc_solve_precompute(const uint64_t* A, size_t n, size_t m, uint64_t q, size_t* piv, uint64_t* T);
/* G[j][i] = <b_j, b~_i> (j > i) and the blocked randomized nearest plane that uses it: contract in psf_oracle_gpv.c */
void orc_np_gram(const int32_t* basis_t, const double* gso_t, size_t dim, double* G /*dim x dim*/);
void orc_nearest_plane(const int32_t* basis_t, const double* gso_t, const double* G, const double* norm2, size_t dim, double s,
                       uint64_t seed, uint32_t tag, uint64_t index, int64_t* c);
/* Gram-Schmidt on the ROWS of St (MatQ::gso on the columns of the reference's matrix) */
void orc_nearest_plane_trace(const int32_t* basis_t, const double* gso_t, const double* G, const double* norm2, size_t dim, double s,
                             uint64_t seed, uint32_t tag, uint64_t index, int64_t* c, double* centres, int64_t* z_out);
int orc_np_two_pass(uint64_t q, size_t n, double s);      /* q sqrt(n) > 2^13 s: sample in two passes (psf_oracle_gpv.c) */
void orc_gpv_set_two_pass(void*, int mode);               /* -1 auto, 0 / 1 forced */
int orc_gpv_two_pass(const void*);
int orc_gpv_samp_p_trace(const void*, uint64_t seed, uint64_t index, const uint64_t* u, int64_t* e, int64_t* c_start, double* centres, int64_t* z);
void orc_gso_rows(const int32_t* St, size_t m, double* Gt);
void orc_gso_rows_leading(const int32_t* St, size_t nrows, size_t width, double* Gt);
void* orc_gpv_new(const orc_gadget_params* gp, double s);
void orc_gpv_free(void*);
size_t orc_gpv_m(const void*);
uint64_t* orc_gpv_A(void*);
int8_t* orc_gpv_R(void*);
int32_t* orc_gpv_basis_t(void*);       /* m x m, row i = basis vector i (column i of S_A) */
double* orc_gpv_gso_t(void*);          /* m x m, row i = b~_i */
int orc_gpv_trap_gen(void*, uint64_t seed);                                               /* gpv.rs:83-94 */
int orc_gpv_load_key(void*, const uint64_t* A, const int32_t* basis_t, const double* gso_t);
int orc_gpv_samp_p(const void*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e,
                   int percall_elimination, int nthreads);                               /* gpv.rs:152-161 */
int orc_gpv_samp_d(const void*, uint64_t seed, uint64_t first_index, size_t B, int64_t* e);  /* gpv.rs:113-116 */
int orc_gpv_f_a(const void*, size_t B, const int64_t* e, uint64_t* u);                     /* gpv.rs:190-193 */
int orc_gpv_check_domain(const void*, size_t B, const int64_t* e, size_t len, uint8_t* ok); /* gpv.rs:219-224 */
/* ---- ring variant (gadget_ring.rs, short_basis_ring.rs, gpv_ring.rs); polynomials = n int64/uint64 coefficients ---- */
int orc_ring_trap_gen(const orc_gadget_params* gp, double s_td, uint64_t seed, uint64_t* a /*(k+2) x n*/, int64_t* r /*k x n*/, int64_t* e /*k x n*/);
int orc_ring_compute_s(const orc_gadget_params* gp, int64_t* sk /*k x k*/);
int orc_find_solution_gadget_ring(const uint64_t* u, size_t n, uint64_t q, uint64_t k, uint64_t base, int64_t* out /*k x n*/);
void orc_ring_gen_sa_l(const int64_t* first, const int64_t* second, size_t n, size_t k, int64_t* out /*(k+2)^2 x n*/);
int orc_ring_gen_sa_r(const orc_gadget_params* gp, const uint64_t* a, int64_t* out /*(k+2) x n(k+2) x n*/);
int orc_ring_short_basis_t(const orc_gadget_params* gp, const uint64_t* a, const int64_t* r, const int64_t* e, int32_t* basis_t /*d x d*/);
void orc_ring_embed_a(const uint64_t* a, size_t n, size_t K, uint64_t q, uint64_t* A_emb /*n x nK*/);
/* rotation_matrix.rs:41-63 / :85-96 */
void orc_rot_minus(const int64_t* vec, size_t n, int64_t* out, size_t ld, size_t col_off);
void orc_rot_minus_matrix(const int64_t* mat, size_t rows, size_t cols, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif
