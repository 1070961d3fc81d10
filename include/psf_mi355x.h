/*
 * psf_mi355x.h -- C ABI of the MI355X-native preimage-sampling library (libpsf_mi355x.so).
 *
 * Drop-in boundary for ONE hot path of qfall/tools: the `PSF` trait (src/primitive/psf.rs:39-81)
 * as implemented by PSFPerturbation (src/primitive/psf/mp_perturbation.rs), PSFGPV (gpv.rs) and
 * PSFGPVRing (gpv_ring.rs), plus the gadget-lattice helpers under them
 * (src/sample/g_trapdoor/{gadget_classical,gadget_ring,short_basis_classical,short_basis_ring}.rs).
 * The reference has no FFI of its own for this path (it is Rust over qfall-math/FLINT); these entry
 * points are what a Rust `extern "C"` block implementing `PSF` would bind -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns a psf_status; 0 = success.  The reference panics where this ABI returns
 *     a non-zero status (mp_perturbation.rs:190,315,333,367); a shim turns status != 0 into panic!.
 *   - matrices are flat, row-major; batches are "one row per call of the reference":
 *       u : B x n   (Range  = MatZq n x 1 per call,   least non-negative residues)
 *       e : B x m   (Domain = MatZ  m x 1 per call)
 *   - one reference call == one row.  Batching (B > 1) is this library's extension: B independent
 *     samp_p calls sharing (A, trapdoor).  Row b of a batch uses the randomness of global preimage
 *     index `first_index + b`, so results do not depend on how a job is sharded over GPUs.
 *   - randomness: the reference's trait takes no seed (psf.rs:48-80); here every sampling entry point
 *     takes a 64-bit seed keying Philox4x32-10 streams (DESIGN.md "Randomness contract").
 *   - `*_dev` variants take device pointers (HIP) and a hipStream_t (as void*); the plain variants
 *     take host pointers and do the copies themselves.
 *   - a handle is not thread-safe: one handle per host thread / HIP stream, mirroring the reference's
 *     !Send + !Sync PSF values (gadget_parameters.rs:51, trapdoor_distribution.rs:22).
 */
#ifndef PSF_MI355X_H
#define PSF_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int psf_status;
enum {
  PSF_OK = 0,
  PSF_ERR_PARAM = 1,           /* malformed arguments (dimension / NULL / range)                         */
  PSF_ERR_NOT_PD = 2,          /* Sigma_2 not positive definite: mp_perturbation.rs:109-110              */
  PSF_ERR_DOMAIN = 3,          /* f_a on sigma outside D_n: assert! at mp_perturbation.rs:367, gpv.rs:191 */
  PSF_ERR_MODULUS = 4,         /* base^k < q: gadget_classical.rs:170-172                                */
  PSF_ERR_NO_SOLUTION = 5,     /* A x = u has no solution: gpv.rs:153-155 unwrap                         */
  PSF_ERR_NO_KEY = 6,          /* samp_p / f_a before trap_gen / load_key                                */
  PSF_ERR_HIP = 7,             /* HIP runtime error (no device, out of memory, launch failure)           */
  PSF_ERR_UNSUPPORTED = 8,     /* parameter combination outside what the kernels cover                   */
  PSF_ERR_SAMPLER = 9          /* a draw did not accept within 65 536 attempts (Gaussian far below the smoothing parameter), or an
                                  intermediate left its range (nearest-plane centre beyond 2^62 or first-pass representative beyond 2^53,
                                  perturbation beyond 2^23);
                                  the reference would keep computing -- DESIGN.md section 8, "Limits" */
};

const char* psf_status_string(psf_status);
/* "gfx950" etc. of the device the library would run on; PSF_ERR_HIP if none */
psf_status psf_device_info(int device, char* name, size_t name_len, int* compute_units);

/* ------------------------------------------------------------------------------------------------
 * GadgetParameters (gadget_parameters.rs:44-52); distribution is PlusMinusOneZero
 * (trapdoor_distribution.rs:52-53, :82-86) for the classical variants, SampleZ (:58-59) for the ring.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t n;      /* security parameter / ring degree            */
  uint64_t k;      /* gadget length, ceil(log_base q) by default  */
  uint64_t m_bar;  /* n*k + ceil(log2 n)^2 (classical), k+2 (ring) */
  uint64_t base;   /* gadget base (2 by default)                  */
  uint64_t q;      /* modulus, 1 < q < 2^62                       */
} psf_gadget_params;

/* GadgetParameters::init_default (gadget_parameters.rs:113-133) */
psf_status psf_gadget_params_default(uint64_t n, uint64_t q, psf_gadget_params* out);
/* GadgetParametersRing::init_default (gadget_parameters.rs:165-185); modulus polynomial X^n + 1
 * (common_moduli.rs:41-48) */
psf_status psf_gadget_params_ring_default(uint64_t n, uint64_t q, psf_gadget_params* out);

/* ------------------------------------------------------------------------------------------------
 * Deterministic gadget helpers (host entry points; the batched digit decomposition runs on device)
 * ---------------------------------------------------------------------------------------------- */
/* gen_gadget_vec (gadget_classical.rs:128-136): out[k] = base^i */
psf_status psf_gen_gadget_vec(uint64_t k, uint64_t base, int64_t* out);
/* gen_gadget_mat (gadget_classical.rs:91-107): out[n x n*k] = I_n (x) g^t */
psf_status psf_gen_gadget_mat(uint64_t n, uint64_t k, uint64_t base, int64_t* out);
/* find_solution_gadget_mat (gadget_classical.rs:219-229; entry rule :169-182):
 * value[rows x cols] in Z_q -> out[k*rows x cols], out[k*j+i, c] = i-th base-`base` digit of value[j,c].
 * Runs the HIP digit-decomposition kernel.  PSF_ERR_MODULUS if base^k < q. */
psf_status psf_find_solution_gadget_mat(int device, const uint64_t* value, size_t rows, size_t cols,
                                        uint64_t q, uint64_t k, uint64_t base, int64_t* out);
/* short_basis_gadget (gadget_classical.rs:248-287): out[nk x nk] = I_n (x) S_k */
psf_status psf_short_basis_gadget(const psf_gadget_params* gp, int64_t* out);
/* gen_short_basis_for_trapdoor (short_basis_classical.rs:54-63), tag = identity when NULL:
 * out[m x m] = [I R; 0 I] * [0 I; S' W] */
psf_status psf_gen_short_basis_for_trapdoor(const psf_gadget_params* gp, const uint64_t* tag /*n x n*/,
                                            const uint64_t* A /*n x m*/, const int8_t* R /*m_bar x nk*/,
                                            int64_t* out);
/* gen_trapdoor (gadget_classical.rs:56-68) for a caller-supplied A_bar (n x m_bar) and tag H (n x n, NULL = identity, :61-66):
 * R <- PlusMinusOneZero (trapdoor_distribution.rs:82-86) from `seed`, A = [A_bar | H G - A_bar R] computed on the device.
 * PSF_ERR_MODULUS if base^k < q. */
psf_status psf_gen_trapdoor(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const uint64_t* tag, uint64_t seed,
                            uint64_t* A /*n x m*/, int8_t* R /*m_bar x nk*/);
/* the same with the caller's own R (m_bar x nk, row-major) -- the draw of whatever TrapdoorDistribution the caller uses: the reference samples R through
 * the trait object `params.distribution` (gadget_classical.rs:62-64, gadget_parameters.rs:51, trapdoor_distribution.rs:21-48).  |R_ij| <= 127
 * (the trapdoor is an int8 operand of the matrix cores on the device), PSF_ERR_UNSUPPORTED otherwise.  Install the pair with psfp_load_key(A, R, NULL). */
psf_status psf_gen_trapdoor_with_r(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const uint64_t* tag, const int64_t* R,
                                   uint64_t* A /*n x m*/);
/* gen_trapdoor_ring_lwe (gadget_ring.rs:62-81): r, e <- SampleZ(s) from `seed`, a = [1 | a_bar | g^t - (a_bar r + e)] in R_q
 * (gp from psf_gadget_params_ring_default; q < 2^62).  a_bar: n coefficients; a: (k+2) x n; r, e: k x n */
psf_status psf_gen_trapdoor_ring_lwe(int device, const psf_gadget_params* gp, const uint64_t* a_bar, double s, uint64_t seed,
                                     uint64_t* a, int64_t* r, int64_t* e);
/* the same with the caller's own r, e (gadget_ring.rs:69-70 draw them through `params.distribution`); |coefficients| <= 2^30 */
psf_status psf_gen_trapdoor_ring_lwe_with(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const int64_t* r, const int64_t* e, uint64_t* a);
/* gen_gadget_ring (gadget_ring.rs:103-109): the k constant polynomials base^j, out[j] = constant term */
psf_status psf_gen_gadget_ring(uint64_t k, uint64_t base, int64_t* out);
/* find_solution_gadget_ring (gadget_ring.rs:145-166): u (n coefficients of an element of R_q) -> out[k x n], polynomial i = i-th digit
 * of every coefficient (index i + j k of the classical solution, :160) */
psf_status psf_find_solution_gadget_ring(int device, const uint64_t* u, size_t n, uint64_t q, uint64_t k, uint64_t base, int64_t* out);
/* gen_short_basis_for_trapdoor_ring (short_basis_ring.rs:64-79): out[(row * n(k+2) + col) * n + coeff], (k+2) x n(k+2) polynomials */
psf_status psf_gen_short_basis_for_trapdoor_ring(const psf_gadget_params* gp, const uint64_t* a, const int64_t* r, const int64_t* e, int64_t* out);
/* Gram-Schmidt orthogonalisation of the ROWS of an integer matrix on the device -- the `MatQ::gso` step of PSFGPV::trap_gen (gpv.rs:88-91) and of
 * MatPolyOverZ::sample_d (gpv_ring.rs:205), as a free function: basis_t[rows x width] (row i = basis vector i), out[rows x width] = b~_i.
 * Blocked, re-orthogonalised, FP64 matrix cores (psf_gemm_kernels.hpp).  PSF_ERR_PARAM if the rows are linearly dependent. */
psf_status psf_gso_rows(int device, const int32_t* basis_t, size_t rows, size_t width, double* out);
/* contiguous shares of `total` rows over `world` workers; the first total % world workers get one row more */
psf_status psf_shard_range(size_t total, int world, int rank, size_t* first, size_t* count);
/* PolynomialRingZq product in R_q = Z_q[X]/(X^n + 1) (common_moduli.rs:41-48), the arithmetic under the MatPolynomialRingZq
 * products at gadget_ring.rs:78 and gpv_ring.rs:245-246: out[c] = a[c] * b[c] mod (X^n + 1, q) for `count` pairs of n
 * coefficients (constant term first); a as residues, b as signed integers (a MatPolyOverZ entry).  Runs on the device. */
psf_status psf_poly_mul_negacyclic(int device, uint64_t q, size_t n, size_t count, const uint64_t* a, const int64_t* b, uint64_t* out);
/* the same product with the method chosen explicitly: method 0 = schoolbook kernel, 1 = (incomplete) negacyclic NTT kernel,
 * available when q < 2^31 is a prime with 4 | q-1 and n is a power of two (q = 3329, n = 256: seven levels and degree-1
 * leaves, as in ML-KEM); PSF_ERR_UNSUPPORTED otherwise.  psf_poly_mul_negacyclic uses the NTT whenever it is available. */
psf_status psf_poly_mul_negacyclic_method(int device, uint64_t q, size_t n, size_t count, const uint64_t* a, const int64_t* b, uint64_t* out, int method);
/* The same product on DEVICE buffers in the caller's stream, nothing allocated per call (the tables of a (device, q, n) are built at first use and
 * kept): `count` products d_out[c] = d_a[c] * d_b[c] (PolynomialRingZq multiplication, gadget_ring.rs:78, gpv_ring.rs:245-246).
 *   io_bits = 64: the layout above (a uint64 of any value, b int64 of any value, out uint64 in [0, q)); every modulus below 2^62 (NTT when q is an
 *                 NTT-friendly prime below 2^31, the schoolbook kernel otherwise).
 *   io_bits = 16: a uint16 in [0, q), b int16 in (-q, q), out uint16 in [0, q) -- a quarter of the bytes; NTT-friendly primes q < 2^14
 *                 (3329, 7681, 12289: common_moduli.rs:41-48) with n = 128 ... 1024; PSF_ERR_UNSUPPORTED otherwise.
 * One 128 ... 1024-point transform per WAVEFRONT: every butterfly in registers, lane bits exchanged by DPP / permlane swaps, Montgomery
 * arithmetic (R = 2^16 on 24-bit multiplies for q < 2^14, R = 2^32 above), no division and no barrier (tools_amd/csrc/psf_ntt_core.hpp). */
psf_status psf_poly_mul_negacyclic_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, const void* d_b, void* d_out, int io_bits, void* stream);
/* A polynomial that takes part in many products (a key: a_bar of gadget_ring.rs:78, a of gpv_ring.rs:245) is transformed ONCE:
 * psf_ntt_forward_dev writes its image (count * n 32-bit words, opaque), psf_poly_mul_hat_dev multiplies images by polynomials: d_out[c] =
 * image[c * hat_stride ...] * d_b[c]; hat_stride in words, 0 = one image for every product.  Shapes with a wave kernel only (q an NTT-friendly
 * prime < 2^31, n = 128 ... 1024, leaf degree <= 4); PSF_ERR_UNSUPPORTED otherwise. */
psf_status psf_ntt_forward_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, int io_bits, uint32_t* d_hat, void* stream);
psf_status psf_poly_mul_hat_dev(int device, uint64_t q, size_t n, size_t count, const uint32_t* d_hat, size_t hat_stride, const void* d_b, void* d_out, int io_bits,
                                void* stream);
/* rot_minus_matrix (rotation_matrix.rs:85-96): mat[rows x cols] -> out[rows x rows*cols] */
psf_status psf_rot_minus_matrix(const int64_t* mat, size_t rows, size_t cols, int64_t* out);

/* ------------------------------------------------------------------------------------------------
 * PSFPerturbation (mp_perturbation.rs:57-62, impl PSF :193-403)
 *   A        = MatZq  n x m            -> uint64_t[n*m]
 *   Trapdoor = (R, sqrt(Sigma_2), (S, S~)) (mp_perturbation.rs:195)
 *                R            : int8_t[m_bar * n*k]           entries in {-1,0,1}
 *                sqrt(Sigma_2): double, lower triangular, packed by rows: row i holds i+1 entries,
 *                               m(m+1)/2 doubles (the Cholesky factor of :138)
 *                (S, S~)      : I_n (x) S_k and its GSO; a function of the parameters only, so it is
 *                               rebuilt inside the handle instead of being passed around
 *   Domain   = MatZ m x 1, Range = MatZq n x 1
 * ---------------------------------------------------------------------------------------------- */
typedef struct psfp_handle psfp_handle;

typedef struct {
  psf_gadget_params gp;
  double r;        /* rounding parameter (mp_perturbation.rs:60) */
  double s;        /* Gaussian parameter (mp_perturbation.rs:61) */
  int32_t device;  /* HIP device ordinal */
  uint32_t flags;  /* 0, or PSFP_FLAG_STRUCTURED_SQRT */
} psfp_params;

/* Opt-in: a STRUCTURED square root of Sigma_2 instead of the dense Cholesky factor of mp_perturbation.rs:138.  With Sigma = s^2 I,
 *   Sigma_2 = c [[alpha I - kappa R R^t, -kappa R], [-kappa R^t, beta I]],  c = r^2 / 2 pi, kappa = base^2 + 1, alpha = s^2 - 1, beta = alpha - kappa,
 * factors as B B^t with B = [[L_1, -g R], [0, h I]], L_1 = chol(c (alpha I - kappa (alpha / beta) R R^t)) (m_bar x m_bar), g = sqrt(c) kappa / sqrt(beta),
 * h = sqrt(c beta).  The perturbation is sampled as x_top = L_1 d_1 - g R d_2, x_bot = h d_2: a quarter of the FP64 work, a key of m_bar(m_bar+1)/2 doubles
 * instead of m(m+1)/2, trap_gen eight times cheaper.  Any square root of Sigma_2 gives the same distribution (the reference's sampler only needs
 * B B^t = Sigma_2); individual outputs differ from the dense path, so this is a different -- labelled -- algorithm, not the parity path.
 * d_2 is taken in fixed point (multiples of 2^-32) so that R d_2 is an exact integer sum.  In this mode the `sqrt_sigma2_packed` arguments of
 * psfp_load_key / psfp_export_key / psfp_export_sqrt_sigma2_rows hold L_1 (m_bar(m_bar+1)/2 doubles); psfp_load_key expects a factor produced with the
 * handle's (r, s). */
#define PSFP_FLAG_STRUCTURED_SQRT 2u

/* Limits: k <= 64, 1 < q < 2^62, and s * r * sqrt(m) < 2^23 (every coordinate of an in-domain vector then fits the three int8
 * digit planes of the Z_q products); PSF_ERR_UNSUPPORTED otherwise.  BASELINE's largest set (n=1024, q=2^60, s=1024, r=10)
 * reaches 3.6e6 of the 8.39e6 allowed. */
psf_status psfp_create(const psfp_params* params, psfp_handle** out);
void       psfp_destroy(psfp_handle*);
/* m = m_bar + n*k */
size_t     psfp_m(const psfp_handle*);

/* PSF::trap_gen (mp_perturbation.rs:221-244): samples A_bar, R on device, builds A = [A_bar | G - A_bar R]
 * (gadget_classical.rs:56-68), Sigma_2 and its Cholesky factor (mp_perturbation.rs:111-139).
 * PSF_ERR_NOT_PD if Sigma_2 is not positive definite (s too small). */
psf_status psfp_trap_gen(psfp_handle*, uint64_t seed);
/* PSFPerturbation::compute_sqrt_sigma_2 (mp_perturbation.rs:111-139) for Sigma = s_cov^2 * I using the
 * handle's R; replaces the handle's sqrt(Sigma_2) (the doctest at :89-107).  With PSFP_FLAG_STRUCTURED_SQRT only s_cov == s is
 * accepted (PSF_ERR_UNSUPPORTED otherwise): the structured factor's constants are rebuilt from s when a key is loaded. */
psf_status psfp_compute_sqrt_sigma_2(psfp_handle*, double s_cov);
/* The general form: `mat_sigma: &MatQ` of mp_perturbation.rs:111 is any symmetric m x m matrix (used as a full matrix at :125-126).
 * sigma_lower_packed: its lower triangle, row i holding i + 1 entries (m(m+1)/2 doubles).  PSF_ERR_NOT_PD if Sigma_2 is not positive
 * definite; PSF_ERR_UNSUPPORTED on a PSFP_FLAG_STRUCTURED_SQRT handle (that factor exists for Sigma = s^2 I only). */
psf_status psfp_compute_sqrt_sigma_2_dense(psfp_handle*, const double* sigma_lower_packed);
/* install / read back key material (host buffers).  Any of the out pointers may be NULL.
 * psfp_load_key(A, NULL, NULL): the PUBLIC key only -- what a verifier holds; f_a, check_domain and samp_d work (PSF::f_a takes `a` alone,
 *   mp_perturbation.rs:366), samp_p returns PSF_ERR_NO_KEY.
 * psfp_load_key(A, R, NULL): sqrt(Sigma_2) is recomputed from R with the handle's s, as trap_gen does (mp_perturbation.rs:227-231). */
psf_status psfp_load_key(psfp_handle*, const uint64_t* A, const int8_t* R, const double* sqrt_sigma2_packed);
/* (A, R) WITHOUT a factor and without computing one: the state PSFPerturbation::compute_sqrt_sigma_2 (mp_perturbation.rs:111-139, a pure function of
 * mat_r and mat_sigma) starts from.  A may be NULL (the handle's public matrix, if any, stays installed).  samp_p returns PSF_ERR_NO_KEY until
 * psfp_compute_sqrt_sigma_2 / _dense has produced the factor.  (psfp_load_key(A, R, NULL) instead runs the whole Cholesky with the handle's s.) */
psf_status psfp_load_trapdoor(psfp_handle*, const uint64_t* A, const int8_t* R);
psf_status psfp_export_key(const psfp_handle*, uint64_t* A, int8_t* R, double* sqrt_sigma2_packed);
/* rows [row0, row0 + nrows) of sqrt(Sigma_2) in the same packed form (row i holds i + 1 entries): the factor of BASELINE's
 * largest set is 60.5 GB, so a caller that inspects or ships it does so in row blocks */
psf_status psfp_export_sqrt_sigma2_rows(const psfp_handle*, size_t row0, size_t nrows, double* out);
/* the gadget part of the trapdoor tuple: S_k (k x k) and its Gram-Schmidt vectors (columns, k x k) */
psf_status psfp_export_gadget_basis(const psfp_handle*, int64_t* Sk, double* Sk_gso);

/* PSF::samp_d (mp_perturbation.rs:264-267): e[b] <- D_{Z^m, s*r} */
psf_status psfp_samp_d(psfp_handle*, uint64_t seed, uint64_t first_index, size_t B, int64_t* e);
/* PSF::samp_p (mp_perturbation.rs:304-336), B independent calls */
psf_status psfp_samp_p(psfp_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e);
/* The same, asynchronous: returns once the work is enqueued (u has been staged and may be reused); e[] is complete when psfp_wait returns.  At most
 * two calls are in flight per handle (a third waits for the first).  The rows of call i cross PCIe (narrowed to int32 on the device, widened into e by
 * worker threads) while call i + 1 computes, so a loop of asynchronous calls runs at the device-resident rate; psfp_samp_p = psfp_samp_p_async +
 * psfp_wait.  psfp_wait returns the first non-OK status of the outstanding calls, oldest first (PSF_ERR_SAMPLER as psfp_samp_p would).
 * RULE: every other entry point that rewrites the key (psfp_trap_gen, psfp_load_key, psfp_load_trapdoor, psfp_compute_sqrt_sigma_2(_dense)) or uses the
 * handle's per-batch buffers (psfp_samp_p_dev, psfp_samp_d(_dev), psfp_f_a(_dev), psfp_samp_p_stages) first waits for the asynchronous calls in flight --
 * they never see a half-replaced key or share buffers with the new call -- and returns their status if one of them failed.  A handle is driven by one
 * thread at a time. */
psf_status psfp_samp_p_async(psfp_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e);
psf_status psfp_wait(psfp_handle*);
/* Per-call status for a caller that keeps several batches (the shim's PendingBatch).  Every asynchronous call of a handle carries a ticket 0, 1, 2, ...:
 * psfp_async_next_ticket says which one the NEXT call will get; psfp_wait_ticket waits for that call (and the older one in flight, nothing newer) and returns ITS
 * status, however many other waits have joined it in the meantime (psfp_wait reports the first failure of everything outstanding and so consumes statuses of
 * calls the caller may not be asking about).  PSF_ERR_PARAM for a ticket never issued or older than the handle's last 8 joined calls.  (The synchronous
 * psfp_samp_p of a large batch is an asynchronous call + wait inside the library and takes a ticket too.) */
uint64_t   psfp_async_next_ticket(const psfp_handle*);
psf_status psfp_wait_ticket(psfp_handle*, uint64_t ticket);
/* PSF::f_a (mp_perturbation.rs:366-369): u[b] = A e[b] mod q; PSF_ERR_DOMAIN (u still written) if any
 * row fails check_domain */
psf_status psfp_f_a(psfp_handle*, size_t B, const int64_t* e, uint64_t* u);
/* PSF::check_domain (mp_perturbation.rs:396-402) for rows of length `len`; ok[b] = 0/1 */
psf_status psfp_check_domain(psfp_handle*, size_t B, const int64_t* e, size_t len, uint8_t* ok);

/* One job over `count` handles, one per GPU of the node, each holding the same key (psfp_trap_gen with the same seed, or psfp_load_key):
 * the B rows are cut into contiguous shares (psf_shard_range), share i is computed by handles[i] on its own device, all devices at once.
 * Row b uses global index first_index + b, so the result equals psfp_samp_p on one handle bit for bit.  Host buffers (pageable is fine):
 * one worker thread per handle drives its device for the call (upload, samp_p, download), because HIP copies from / to pageable memory
 * block the issuing thread.  A handle may appear once in `handles`.  Every worker finishes and synchronises its stream before the call
 * returns, also after an error on another device; the first non-OK status in handle order is returned. */
psf_status psfp_samp_p_multi(psfp_handle* const* handles, int count, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e);
/* this handle's window inside the last psfp_samp_p_multi call, in ms since that call began (host clock): its first samp_p launch
 * sequence enqueued / its last row landed in e; -1 if it had no rows.  Overlapping windows = the devices worked at the same time. */
psf_status psfp_get_multi_timing(const psfp_handle*, double* launched_ms, double* done_ms);

/* device-resident variants: d_u, d_e are HIP device pointers, stream is a hipStream_t (NULL = default).
 * Asynchronous with respect to the host; errors detected on device are reported by psfp_last_status. */
psf_status psfp_samp_p_dev(psfp_handle*, uint64_t seed, uint64_t first_index, size_t B,
                           const uint64_t* d_u, int64_t* d_e, void* stream);
psf_status psfp_samp_d_dev(psfp_handle*, uint64_t seed, uint64_t first_index, size_t B, int64_t* d_e, void* stream);
psf_status psfp_f_a_dev(psfp_handle*, size_t B, const int64_t* d_e, uint64_t* d_u, uint8_t* d_ok, void* stream);
/* synchronises the stream of the last *_dev call and returns the device-side status of that call */
psf_status psfp_last_status(psfp_handle*);
/* synthetic uniform targets u <- Z_q^{B x n} (benches/psf.rs:35,60,87) written to device memory */
psf_status psfp_uniform_targets_dev(psfp_handle*, uint64_t seed, uint64_t first_index, size_t B, uint64_t* d_u, void* stream);
/* result rows for the inter-GPU gather: narrows count int64 preimage coordinates in device memory to
 * int32 (|e_i| <= 6 s r sqrt(m) for every parameter set of mp_perturbation.rs / gpv.rs); *d_overflow (device int) is
 * OR-ed with 1 if a value does not fit, so the caller can refuse to ship truncated rows */
psf_status psf_narrow_rows_dev(const int64_t* d_src, int32_t* d_dst, size_t count, int* d_overflow, int device, void* stream);

/* stage-level access for parity tests and profiling (host buffers; NULL = skip):
 * runs samp_p for B rows and copies out the intermediates of the reference's call stack
 *   d : B x m  standard normals fed to sqrt(Sigma_2)        (mp_perturbation.rs:315)
 *   x : B x m  centres  x = sqrt(Sigma_2) d
 *   p : B x m  perturbation p_i <- D_{Z,r,x_i}
 *   v : B x n  v = u - A p                                  (:318)
 *   z : B x nk gadget preimage                               (:321-326)
 */
psf_status psfp_samp_p_stages(psfp_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u,
                              double* d, double* x, int64_t* p, uint64_t* v, int64_t* z, int64_t* e);
/* per-kernel average duration (ms) of the last samp_p*_dev/samp_p call, measured with HIP events on the
 * launch stream when timing is enabled; names are ';'-separated in `names`. */
psf_status psfp_enable_timing(psfp_handle*, int on);
psf_status psfp_get_timing(psfp_handle*, char* names, size_t names_len, double* ms, size_t* count);

/* ------------------------------------------------------------------------------------------------
 * PSFGPV (gpv.rs:53-57, impl PSF :59-225)
 *   A        = MatZq n x m
 *   Trapdoor = (short_base, short_base_gso) (gpv.rs:61): both m x m.  They cross this ABI TRANSPOSED: row i of
 *              `basis_t` / `gso_t` is basis vector i, i.e. column i of the reference's MatZ / MatQ
 *              (gen_short_basis_for_trapdoor, short_basis_classical.rs:54-63; MatQ::gso, gpv.rs:91).
 *   samp_p (gpv.rs:152-161): sol = A.solve_gaussian_elimination(u); e = sol + SampleD(basis, gso, -sol, s), SampleD in the
 *   batched blocked form of tools_amd/csrc/psf_np_kernels.hpp (any lattice dimension that fits the device memory).
 *   The elimination is factored once per key (pivot columns + n x n operator); it returns the same particular
 *   solution as eliminating [A | u] per call with unit pivots and free variables 0.
 *   Precision of the centres.  The reference holds the centre in exact rationals (MatQ, gpv.rs:158-160); the walk here keeps its running
 *   projections in doubles.  With the centre -sol (entries up to q on n coordinates) the error of a centre, in units of that draw's width
 *   s / |b~_i|, is about 2^-53 q sqrt(n) / s whatever the basis.  Keys with q sqrt(n) <= 2^13 s (C2, C4: 2^-47) are sampled in one pass at
 *   a relative centre error <= 2^-40; for larger moduli -- every q < 2^62 -- samp_p runs TWO passes: the first finds a short element e1 of
 *   the coset A e = u, the second samples e1 + D_{Lambda, s, -e1}, whose centres are of ordinary size (|e1| ~ s sqrt(m)): the output
 *   distribution is that of gpv.rs:160 for any e1 (GPV08), so the first pass's imprecision cannot reach it, and the relative centre error
 *   of the pass that matters is <= 2^-35 (asserted against 100-digit arithmetic in tests/test_oracle_centre_precision.py; the floor is
 *   the double-precision Gram-Schmidt data, 1e-13 relative).  psfgpv_two_pass() says which form a handle uses.
 * ---------------------------------------------------------------------------------------------- */
typedef struct psfgpv_handle psfgpv_handle;
typedef struct {
  psf_gadget_params gp;
  double s;        /* Gaussian parameter (gpv.rs:56) */
  int32_t device;
  uint32_t flags;  /* reserved, 0 */
} psfgpv_params;

psf_status psfgpv_create(const psfgpv_params* params, psfgpv_handle** out);
void       psfgpv_destroy(psfgpv_handle*);
size_t     psfgpv_m(const psfgpv_handle*);
/* PSF::trap_gen (gpv.rs:83-94): A, R as for PSFPerturbation; short basis and its GSO built on device.
 * PSF_ERR_NO_SOLUTION if A has fewer than n unit pivots mod q. */
psf_status psfgpv_trap_gen(psfgpv_handle*, uint64_t seed);
psf_status psfgpv_load_key(psfgpv_handle*, const uint64_t* A, const int32_t* basis_t, const double* gso_t);
psf_status psfgpv_export_key(const psfgpv_handle*, uint64_t* A, int8_t* R, int32_t* basis_t, double* gso_t);
psf_status psfgpv_samp_d(psfgpv_handle*, uint64_t seed, uint64_t first_index, size_t B, int64_t* e);         /* gpv.rs:113-116 */
psf_status psfgpv_samp_p(psfgpv_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e);
psf_status psfgpv_samp_p_dev(psfgpv_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_e, void* stream);
/* gpv.rs:152-161 on host buffers without waiting (the transport of psfp_samp_p_async: rows narrowed to int32 on the device, chunk transfers by the DMA engines into
 * per-call pinned rings, widened into e by worker threads): returns once the work is enqueued (u may be reused), e[] is complete when psfgpv_wait returns.  At most two
 * calls in flight per handle; the rows of call i cross PCIe while call i + 1 walks.  psfgpv_wait returns the first non-OK status of the outstanding calls, oldest first:
 * PSF_ERR_SAMPLER as psfgpv_samp_p would; PSF_ERR_UNSUPPORTED if a row entry did not fit 32 bits (the synchronous call copies 64-bit rows in that case). */
psf_status psfgpv_samp_p_async(psfgpv_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e);
psf_status psfgpv_wait(psfgpv_handle*);
uint64_t   psfgpv_async_next_ticket(const psfgpv_handle*);                    /* as psfp_async_next_ticket / psfp_wait_ticket */
psf_status psfgpv_wait_ticket(psfgpv_handle*, uint64_t ticket);
psf_status psfgpv_f_a(psfgpv_handle*, size_t B, const int64_t* e, uint64_t* u);                               /* gpv.rs:190-193 */
psf_status psfgpv_f_a_dev(psfgpv_handle*, size_t B, const int64_t* d_e, uint64_t* d_u, uint8_t* d_ok, void* stream);
psf_status psfgpv_check_domain(psfgpv_handle*, size_t B, const int64_t* e, size_t len, uint8_t* ok);         /* gpv.rs:219-224 */
psf_status psfgpv_uniform_targets_dev(psfgpv_handle*, uint64_t seed, uint64_t first_index, size_t B, uint64_t* d_u, void* stream);
psf_status psfgpv_last_status(psfgpv_handle*);
/* HIP-event durations (ms) of the last samp_p call (0 if timing was off): the solve kernel, and the whole nearest plane
 * (initial projection, per block one sampling and one update launch, recombination) */
psf_status psfgpv_enable_timing(psfgpv_handle*, int on);
psf_status psfgpv_get_timing(psfgpv_handle*, double* solve_ms, double* nearest_plane_ms);
/* diagnostic of the last samp_p call: the number of 64-row blocks the nearest-plane walk (gpv.rs:160) was cut into, and whether
 * e = sum z_i b_i was recombined by the 64-bit integer kernel (1) instead of the int8 matrix-core planes (0) -- the former when a
 * basis entry or a drawn z_i does not fit two balanced base-256 digits (|.| > 32639); the result is the same either way */
psf_status psfgpv_get_nearest_plane_stats(psfgpv_handle*, size_t* blocks, size_t* generic_recombination);
/* The walk of gpv.rs:160 has two launch forms with identical results: ONE launch (k_np_walk<G>: sampler workgroups and updater workgroups that hand blocks to each
 * other through device memory; chosen when the device's occupancy figures say every workgroup is resident at once) and one launch per 64-row block (k_np_step<G>).
 * A large batch of the second form walks as two column ranges side by side on two streams of the handle (the matrix-core update tiles of one range run beside the
 * samplers of the other), joined on the caller's stream before the call's last kernel: form 2.
 * form: 1 / 0 / 2 as launched by the last call; preimages_per_wave: G; reruns: walks of this handle since its creation in which a workgroup of the one-launch form gave
 * up waiting for another (a GPU shared with other work) and the call was walked again, inside the same call, by a form without waits between workgroups
 * (k_np_walk_solo).  Contention costs time, never the call: PSF_ERR_SAMPLER is reserved for SampleZ itself.  One-launch walks of one process take turns per device. */
psf_status psfgpv_get_nearest_plane_form(psfgpv_handle*, int* form, int* preimages_per_wave, size_t* blocks, uint64_t* reruns);
/* 1 if samp_p of this handle draws in two passes (large moduli, see "Precision of the centres" above), else 0 */
int psfgpv_two_pass(const psfgpv_handle*);

/* ------------------------------------------------------------------------------------------------
 * PSFGPVRing (gpv_ring.rs:62-67, impl PSF :69-284) over R_q = Z_q[X]/(X^n + 1)
 *   A        = MatPolynomialRingZq 1 x (k+2)  -> uint64_t[(k+2) * n], polynomial j at a + j*n, constant term first
 *   Trapdoor = (r, e), two 1 x k MatPolyOverZ  -> int64_t[k * n] each (gadget_ring.rs:62-81)
 *   Domain   = MatPolyOverZ (k+2) x 1          -> int64_t[(k+2) * n] per call
 *   Range    = one element of R_q               -> uint64_t[n] per call
 * Any modulus 1 < q < 2^62 (GadgetParametersRing carries an arbitrary ModulusPolynomialRingZq, gadget_parameters.rs:73-81): R_q products by the NTT
 * kernel where q allows and by the exact schoolbook kernel elsewhere, the walk in two passes where q sqrt(n) > 2^13 s (see PSFGPV above).
 * samp_p (gpv_ring.rs:160-212) works on the coefficient embedding: the short basis
 * (gen_short_basis_for_trapdoor_ring, short_basis_ring.rs:64-79), rot^-(iota(a)) (rotation_matrix.rs:85-96), the
 * elimination and the Gram-Schmidt vectors are built ONCE per key here, where the reference rebuilds them per call.
 * f_a (gpv_ring.rs:243-247) is the R_q product a * sigma, evaluated as rot^-(iota(a)) iota(sigma) on the int8 matrix cores.
 * ---------------------------------------------------------------------------------------------- */
typedef struct psfring_handle psfring_handle;
typedef struct {
  psf_gadget_params gp;   /* from psf_gadget_params_ring_default: m_bar = k + 2 */
  double s;               /* gpv_ring.rs:65 */
  double s_td;            /* gpv_ring.rs:66 */
  int32_t device;
  uint32_t flags;
} psfring_params;

psf_status psfring_create(const psfring_params* params, psfring_handle** out);
void       psfring_destroy(psfring_handle*);
/* PSF::trap_gen (gpv_ring.rs:91-98) */
psf_status psfring_trap_gen(psfring_handle*, uint64_t seed);
psf_status psfring_load_key(psfring_handle*, const uint64_t* a, const int64_t* r, const int64_t* e);
/* a, r, e as above; basis_t / gso_t: d x d with d = n(k+2), row c = embedded basis vector c (any may be NULL) */
psf_status psfring_export_key(const psfring_handle*, uint64_t* a, int64_t* r, int64_t* e, int32_t* basis_t, double* gso_t);
psf_status psfring_samp_d(psfring_handle*, uint64_t seed, uint64_t first_index, size_t B, int64_t* sigma);            /* :118-122 */
psf_status psfring_samp_p(psfring_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* sigma);
psf_status psfring_samp_p_dev(psfring_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_sigma, void* stream);
/* gpv_ring.rs:160-212 without waiting: as psfgpv_samp_p_async / psfgpv_wait */
psf_status psfring_samp_p_async(psfring_handle*, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* sigma);
psf_status psfring_wait(psfring_handle*);
uint64_t   psfring_async_next_ticket(const psfring_handle*);
psf_status psfring_wait_ticket(psfring_handle*, uint64_t ticket);
psf_status psfring_f_a(psfring_handle*, size_t B, const int64_t* sigma, uint64_t* u);                                  /* :243-247 */
psf_status psfring_f_a_dev(psfring_handle*, size_t B, const int64_t* d_sigma, uint64_t* d_u, uint8_t* d_ok, void* stream);
psf_status psfring_check_domain(psfring_handle*, size_t B, const int64_t* sigma, size_t len, uint8_t* ok);            /* :274-283 */
psf_status psfring_uniform_targets_dev(psfring_handle*, uint64_t seed, uint64_t first_index, size_t B, uint64_t* d_u, void* stream);
psf_status psfring_last_status(psfring_handle*);
psf_status psfring_enable_timing(psfring_handle*, int on);
psf_status psfring_get_timing(psfring_handle*, double* solve_ms, double* nearest_plane_ms);
psf_status psfring_get_nearest_plane_form(psfring_handle*, int* form, int* preimages_per_wave, size_t* blocks, uint64_t* reruns);

#ifdef __cplusplus
}
#endif
#endif
