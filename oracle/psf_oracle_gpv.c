/*
 * psf_oracle_gpv.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): PSFGPV (src/primitive/psf/gpv.rs) and the ring
 * helpers of PSFGPVRing (gpv_ring.rs, gadget_ring.rs, short_basis_ring.rs, utils/rotation_matrix.rs).
 * Parity status as in psf_oracle.h: deterministic helpers are pinned by the reference's known-answer tests,
 * sampled values and the particular solution picked by solve_gaussian_elimination are "parity unpinned".
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

static inline uint64_t g_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static inline uint64_t g_submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
static int g_inv_mod(uint64_t a, uint64_t q, uint64_t* inv) {
  i128 t = 0, nt = 1, r = q, nr = a % q;
  while (nr != 0) {
    i128 qq = r / nr, tmp = t - qq * nt;
    t = nt; nt = tmp;
    tmp = r - qq * nr; r = nr; nr = tmp;
  }
  if (r != 1) return 0;
  if (t < 0) t += q;
  *inv = (uint64_t)t;
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * MatZq::solve_gaussian_elimination (gpv.rs:153-156, gpv_ring.rs:180-185), restated as Gauss-Jordan with UNIT
 * pivots: columns are scanned left to right, the first row (at or below the current rank) whose entry is
 * invertible mod q becomes the pivot, free variables are 0.  [UNVERIFIED against qfall-math: which particular
 * solution it returns is not pinned by any reference test; the distribution of samp_p does not depend on it.]
 * A: n x m, u: n.  Returns ORC_ERR_NO_SOLUTION if fewer than n unit pivots exist and the system is inconsistent.
 * ---------------------------------------------------------------------------------------- */
int orc_solve_gaussian_elimination(const uint64_t* A, size_t n, size_t m, uint64_t q, const uint64_t* u, uint64_t* sol) {
  uint64_t* M = (uint64_t*)malloc(n * (m + 1) * sizeof(uint64_t));
  size_t* piv = (size_t*)malloc(n * sizeof(size_t));
  for (size_t i = 0; i < n; ++i) {
    for (size_t j = 0; j < m; ++j) M[i * (m + 1) + j] = A[i * m + j] % q;
    M[i * (m + 1) + m] = u[i] % q;
  }
  size_t rank = 0;
  for (size_t c = 0; c < m && rank < n; ++c) {
    size_t p = n; uint64_t pinv = 0;
    for (size_t r = rank; r < n; ++r)
      if (g_inv_mod(M[r * (m + 1) + c], q, &pinv)) { p = r; break; }
    if (p == n) continue;
    if (p != rank)
      for (size_t j = 0; j <= m; ++j) { uint64_t t = M[p * (m + 1) + j]; M[p * (m + 1) + j] = M[rank * (m + 1) + j]; M[rank * (m + 1) + j] = t; }
    for (size_t j = 0; j <= m; ++j) M[rank * (m + 1) + j] = g_mulmod(M[rank * (m + 1) + j], pinv, q);
    for (size_t r = 0; r < n; ++r) {
      if (r == rank) continue;
      uint64_t f = M[r * (m + 1) + c];
      if (!f) continue;
      for (size_t j = 0; j <= m; ++j) M[r * (m + 1) + j] = g_submod(M[r * (m + 1) + j], g_mulmod(f, M[rank * (m + 1) + j], q), q);
    }
    piv[rank++] = c;
  }
  int rc = ORC_OK;
  for (size_t r = rank; r < n; ++r)
    if (M[r * (m + 1) + m] != 0) rc = ORC_ERR_NO_SOLUTION;
  memset(sol, 0, m * sizeof(uint64_t));
  for (size_t r = 0; r < rank; ++r) sol[piv[r]] = M[r * (m + 1) + m];
  free(M); free(piv);
  return rc;
}

/* The same elimination factored out of the per-call path: pivot columns and T (n x n) with sol[piv[r]] = (T u)[r].
 * Identical result to orc_solve_gaussian_elimination whenever n unit pivots exist (pivoting never looks at u). */
int orc_solve_precompute(const uint64_t* A, size_t n, size_t m, uint64_t q, size_t* piv, uint64_t* T) {
  uint64_t* cur = (uint64_t*)malloc(n * sizeof(uint64_t));
  for (size_t i = 0; i < n * n; ++i) T[i] = 0;
  for (size_t i = 0; i < n; ++i) T[i * n + i] = 1 % q;
  size_t rank = 0;
  for (size_t c = 0; c < m && rank < n; ++c) {
    for (size_t r = 0; r < n; ++r) {           /* column c of T*A */
      u128 acc = 0;
      for (size_t t = 0; t < n; ++t) acc = (acc + (u128)T[r * n + t] * (A[t * m + c] % q)) % q;
      cur[r] = (uint64_t)acc;
    }
    size_t p = n; uint64_t pinv = 0;
    for (size_t r = rank; r < n; ++r)
      if (g_inv_mod(cur[r], q, &pinv)) { p = r; break; }
    if (p == n) continue;
    if (p != rank) {
      for (size_t j = 0; j < n; ++j) { uint64_t t = T[p * n + j]; T[p * n + j] = T[rank * n + j]; T[rank * n + j] = t; }
      uint64_t t = cur[p]; cur[p] = cur[rank]; cur[rank] = t;
    }
    for (size_t j = 0; j < n; ++j) T[rank * n + j] = g_mulmod(T[rank * n + j], pinv, q);
    for (size_t r = 0; r < n; ++r) {
      if (r == rank || !cur[r]) continue;
      uint64_t f = cur[r];
      for (size_t j = 0; j < n; ++j) T[r * n + j] = g_submod(T[r * n + j], g_mulmod(f, T[rank * n + j], q), q);
    }
    piv[rank++] = c;
  }
  free(cur);
  return rank == n ? ORC_OK : ORC_ERR_NO_SOLUTION;
}

/* ------------------------------------------------------------------------------------------
 * MatZ::sample_d_precomputed_gso (gpv.rs:160; GPV08 SampleD): for i = dim-1..0:
 *   c' = <c, b~_i> / ||b~_i||^2 ; z_i <- D_{Z, s/||b~_i||, c'} ; c -= z_i b_i.      Returns c in place (the sample is
 *   (initial c) - (final c)).
 *
 * Evaluation contract (the batched, blocked form of the same walk; the device follows it bit for bit).  With
 *   g[j][i] = <b_j, b~_i>  (ascending fma chain over the coordinates, from +0; used for j > i only)
 * the projection at step i is  <c - sum_{j>i} z_j b_j, b~_i> = <c0, b~_i> - sum_{j>i} z_j g[j][i].  Rows are cut into blocks
 * of ORC_NP_BLOCK = 64 consecutive indices [64 J, 64 J + 64) (the top block may be short):
 *   t_i  = chain_{j ascending} fma((double) c0[j], b~_i[j], .) from +0                      (all i, once)
 *   for J descending:  for i descending inside J:
 *        c'  = t_i * (1 / ||b~_i||^2)            (the reciprocal rounded once per key)
 *        z_i <- D_{Z, s/||b~_i||, c'}             (the SampleZ contract of psf_oracle.c, stream (tag, index, coordinate i))
 *        t_i' = fma(-(double) z_i, g[i][i'], t_i')            for the rows i' < i of the same block
 *     then for every row i' below the block, for j in J ascending:  t_i' = fma(-(double) z_j, g[j][i'], t_i')
 *   c_final = c0 - sum_i z_i b_i   (integers, exact).
 * basis_t / gso_t are TRANSPOSED: row i holds basis vector i (column i of the reference's matrices).
 * ---------------------------------------------------------------------------------------- */
#define ORC_NP_BLOCK 64

void orc_np_gram(const int32_t* basis_t, const double* gso_t, size_t dim, double* G) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (size_t j = 0; j < dim; ++j) {
    const int32_t* bj = basis_t + j * dim;
    for (size_t i = 0; i < j; ++i) {
      const double* gi = gso_t + i * dim;
      double acc = 0.0;
      for (size_t t = 0; t < dim; ++t) acc = fma((double)bj[t], gi[t], acc);
      G[j * dim + i] = acc;
    }
    for (size_t i = j; i < dim; ++i) G[j * dim + i] = 0.0;
  }
}

void orc_nearest_plane_trace(const int32_t* basis_t, const double* gso_t, const double* G, const double* norm2, size_t dim, double s,
                             uint64_t seed, uint32_t tag, uint64_t index, int64_t* c, double* centres, int64_t* z_out);
void orc_nearest_plane(const int32_t* basis_t, const double* gso_t, const double* G, const double* norm2, size_t dim, double s,
                       uint64_t seed, uint32_t tag, uint64_t index, int64_t* c) {
  orc_nearest_plane_trace(basis_t, gso_t, G, norm2, dim, s, seed, tag, index, c, NULL, NULL);
}
/* the same walk; `centres` (dim doubles) receives c'_i as the sampler saw it, `z_out` the drawn coefficients (either may be NULL) */
void orc_nearest_plane_trace(const int32_t* basis_t, const double* gso_t, const double* G, const double* norm2, size_t dim, double s,
                             uint64_t seed, uint32_t tag, uint64_t index, int64_t* c, double* centres, int64_t* z_out) {
  double* t = (double*)malloc(dim * sizeof(double));
  int64_t* z = (int64_t*)calloc(dim, sizeof(int64_t));
  for (size_t i = 0; i < dim; ++i) {
    const double* gi = gso_t + i * dim;
    double acc = 0.0;
    for (size_t j = 0; j < dim; ++j)
      if (c[j]) acc = fma((double)c[j], gi[j], acc);        /* a zero term leaves the chain unchanged (it never holds -0) */
    t[i] = acc;
  }
  const size_t nblk = (dim + ORC_NP_BLOCK - 1) / ORC_NP_BLOCK;
  for (size_t jb = nblk; jb-- > 0;) {
    const size_t j0 = jb * ORC_NP_BLOCK, j1 = j0 + ORC_NP_BLOCK < dim ? j0 + ORC_NP_BLOCK : dim;
    for (size_t i = j1; i-- > j0;) {
      const double inv = 1.0 / norm2[i];
      const double cen = t[i] * inv;
      if (centres) centres[i] = cen;
      z[i] = orc_sample_z(seed, tag, index, (uint32_t)i, cen, s / sqrt(norm2[i]));
      const double nz = -(double)z[i];
      const double* gi = G + i * dim;
      for (size_t i2 = j0; i2 < i; ++i2) t[i2] = fma(nz, gi[i2], t[i2]);
    }
    for (size_t i2 = 0; i2 < j0; ++i2) {
      double acc = t[i2];
      for (size_t j = j0; j < j1; ++j) acc = fma(-(double)z[j], G[j * dim + i2], acc);
      t[i2] = acc;
    }
  }
  /* c -= sum z_i b_i.  The coefficients take part as the doubles the walk itself used: exact below 2^53, rounded above -- which only happens in the first pass
   * of a two-pass walk at moduli near 2^60, where the result need only be SOME representative of the coset (a rounded z_i shifts it by the lattice vector
   * (z_i - fl(z_i)) b_i), and it is what the device's 64-bit recombination reads (k_np_combine_generic).  Sums of products beyond 2^63 cancel back into range:
   * the arithmetic is modulo 2^64 by definition here (unsigned), not by accident of the compiler.  (Found by tools/fuzz_configs.py: the exact integers used
   * here before differed from the device at q = 2^57 .. 2^59.) */
  for (size_t i = 0; i < dim; ++i) {
    if (z_out) z_out[i] = z[i];
    if (!z[i]) continue;
    const uint64_t zz = (uint64_t)(int64_t)(double)z[i];
    const int32_t* bi = basis_t + i * dim;
    for (size_t j = 0; j < dim; ++j) c[j] = (int64_t)((uint64_t)c[j] - zz * (uint64_t)(int64_t)bi[j]);
  }
  free(t); free(z);
}

/* ------------------------------------------------------------------------------------------
 * PSFGPV (gpv.rs:53-57, impl PSF :59-225)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  orc_gadget_params gp;
  double s;
  size_t m;
  uint64_t* A;        /* n x m */
  int8_t* R;          /* m_bar x w (kept for inspection) */
  int32_t* St;        /* m x m, row i = basis vector i of the short basis S_A */
  double* Gt;         /* m x m, row i = b~_i */
  double* norm2;      /* m */
  double* G;          /* m x m, G[j][i] = <b_j, b~_i> for j > i (orc_np_gram) */
  size_t* piv;        /* n pivot columns */
  uint64_t* T;        /* n x n solve operator */
  int has_solver;
  int two_pass_mode;  /* -1: by orc_np_two_pass(q, n, s); 0 / 1: forced (tests) */
} orc_gpv;

static orc_gpv* gpv_new_impl(const orc_gadget_params* gp, double s) {
  if (!gp || gp->n < 1 || gp->q <= 1 || !(s > 0)) return NULL;
  orc_gpv* h = (orc_gpv*)calloc(1, sizeof(orc_gpv));
  h->gp = *gp; h->s = s; h->two_pass_mode = -1;
  size_t w = gp->n * gp->k;
  h->m = gp->m_bar + w;
  h->A = (uint64_t*)calloc(gp->n * h->m, sizeof(uint64_t));
  h->R = (int8_t*)calloc(gp->m_bar * w, 1);
  h->St = (int32_t*)calloc(h->m * h->m, sizeof(int32_t));
  h->Gt = (double*)calloc(h->m * h->m, sizeof(double));
  h->norm2 = (double*)calloc(h->m, sizeof(double));
  h->G = (double*)calloc(h->m * h->m, sizeof(double));
  h->piv = (size_t*)calloc(gp->n, sizeof(size_t));
  h->T = (uint64_t*)calloc(gp->n * gp->n, sizeof(uint64_t));
  return h;
}
void* orc_gpv_new(const orc_gadget_params* gp, double s) { return gpv_new_impl(gp, s); }
void orc_gpv_free(void* hv) {
  orc_gpv* h = (orc_gpv*)hv;
  if (!h) return;
  free(h->A); free(h->R); free(h->St); free(h->Gt); free(h->norm2); free(h->G); free(h->piv); free(h->T); free(h);
}
size_t orc_gpv_m(const void* h) { return ((const orc_gpv*)h)->m; }
uint64_t* orc_gpv_A(void* h) { return ((orc_gpv*)h)->A; }
int8_t* orc_gpv_R(void* h) { return ((orc_gpv*)h)->R; }
int32_t* orc_gpv_basis_t(void* h) { return ((orc_gpv*)h)->St; }
double* orc_gpv_gso_t(void* h) { return ((orc_gpv*)h)->Gt; }

static int gpv_finish_key(orc_gpv* h) {
  size_t m = h->m;
  for (size_t i = 0; i < m; ++i) {
    double nn = 0.0;
    for (size_t j = 0; j < m; ++j) nn = fma(h->Gt[i * m + j], h->Gt[i * m + j], nn);
    h->norm2[i] = nn;
  }
  orc_np_gram(h->St, h->Gt, m, h->G);
  int rc = orc_solve_precompute(h->A, h->gp.n, m, h->gp.q, h->piv, h->T);
  h->has_solver = (rc == ORC_OK);
  return rc;
}

/* Gram-Schmidt on the rows of St (= columns of S_A), MatQ::gso at gpv.rs:88 -- which is EXACT (rationals).  The restatement works in
 * floating point, so it is written to stay close to the exact result at every size: modified Gram-Schmidt (the coefficient of b~_l is taken
 * from the CURRENT remainder of the vector, not from the original b_i), every vector orthogonalised twice ("twice is enough"), dot products
 * accumulated in long double.  A single classical pass in double -- round 2's form -- loses orthogonality as (|b_i| / |b~_i|)^2 eps, which at
 * C2 (d = 6208, shortest b~ ~ 0.026 from vectors of norm ~ 100) reached 1.6e-6.  The device's blocked form (psf_gemm_kernels.hpp) is compared
 * with this chain within a tolerance; the leading `nrows` vectors of a `width`-dimensional basis only need the rows above them. */
static void gso_chain(const int32_t* St, size_t nrows, size_t width, double* Gt) {
  long double* norm2 = (long double*)malloc(nrows * sizeof(long double));
  for (size_t i = 0; i < nrows; ++i) {
    double* gi = Gt + i * width;
    const int32_t* bi = St + i * width;
    for (size_t j = 0; j < width; ++j) gi[j] = (double)bi[j];
    for (int pass = 0; pass < 2; ++pass)
      for (size_t l = 0; l < i; ++l) {
        const double* gl = Gt + l * width;
        long double num = 0.0L;
        for (size_t j = 0; j < width; ++j) num += (long double)gi[j] * (long double)gl[j];
        const double mu = (double)(num / norm2[l]);
        for (size_t j = 0; j < width; ++j) gi[j] = fma(-mu, gl[j], gi[j]);
      }
    long double nn = 0.0L;
    for (size_t j = 0; j < width; ++j) nn += (long double)gi[j] * (long double)gi[j];
    norm2[i] = nn;
  }
  free(norm2);
}
static void gso_rows(const int32_t* St, size_t m, double* Gt) { gso_chain(St, m, m, Gt); }

void orc_gso_rows(const int32_t* St, size_t m, double* Gt) { gso_rows(St, m, Gt); }

/* The same chain for the LEADING nrows vectors of a basis of dimension `width` (row i of St = basis vector i, ld = width): test infrastructure
 * for tests/test_gpu_gpv_scale.py (the leading-rows trick of orc_psfp_sqrt_sigma_2_leading). */
void orc_gso_rows_leading(const int32_t* St, size_t nrows, size_t width, double* Gt) { gso_chain(St, nrows, width, Gt); }

/* gpv.rs:83-94 */
int orc_gpv_trap_gen(void* hv, uint64_t seed) {
  orc_gpv* h = (orc_gpv*)hv;
  const orc_gadget_params* gp = &h->gp;
  size_t n = gp->n, mb = gp->m_bar, w = n * gp->k, m = h->m;
  uint64_t* a_bar = (uint64_t*)malloc(n * mb * sizeof(uint64_t));
  orc_sample_a_bar(seed, n, mb, gp->q, a_bar);                       /* :84 */
  orc_sample_r(seed, mb, w, h->R);
  int rc = orc_gen_trapdoor(gp, a_bar, NULL, h->R, h->A);            /* :88 */
  free(a_bar);
  if (rc) return rc;
  int64_t* S = (int64_t*)malloc(m * m * sizeof(int64_t));
  rc = orc_gen_short_basis_for_trapdoor(gp, NULL, h->A, h->R, S);    /* :90 */
  if (rc) { free(S); return rc; }
  for (size_t i = 0; i < m; ++i)
    for (size_t j = 0; j < m; ++j) h->St[i * m + j] = (int32_t)S[j * m + i];
  free(S);
  gso_rows(h->St, m, h->Gt);                                          /* :91 */
  return gpv_finish_key(h);
}

int orc_gpv_load_key(void* hv, const uint64_t* A, const int32_t* basis_t, const double* gso_t) {
  orc_gpv* h = (orc_gpv*)hv;
  size_t m = h->m;
  memcpy(h->A, A, h->gp.n * m * sizeof(uint64_t));
  memcpy(h->St, basis_t, m * m * sizeof(int32_t));
  memcpy(h->Gt, gso_t, m * m * sizeof(double));
  return gpv_finish_key(h);
}

/* Large moduli.  The walk keeps its running projections in doubles; their first values <c0, b~_i> are as large as q sqrt(n) |b~_i| for the
 * centre c0 = -sol of gpv.rs:158 (sol has entries up to q on the n pivot columns), so the centre c' = t / |b~|^2 carries an absolute error of
 * about 2^-53 q sqrt(n) / |b~_i| -- in units of the sampler's width s / |b~_i|: 2^-53 q sqrt(n) / s, whatever the basis.  The reference keeps
 * the centre in exact rationals (MatQ, gpv.rs:158-160).  When that relative error would exceed 2^-40, i.e. q sqrt(n) > 2^13 s, the sample is
 * drawn in TWO passes: the first walk (stream ORC_TAG_GPV, centre -sol) only serves to find a SHORT element e1 of the coset (A e1 = u, |e1| ~ s
 * sqrt(m)); the second walk (stream ORC_TAG_GPV2) samples v ~ D_{Lambda, s, -e1} with centres of ordinary size and the result is e = e1 + v ~
 * D_{Lambda_u^perp, s}: exactly the distribution of gpv.rs:160, for ANY coset representative e1 (GPV08 SampleD is correct for every centre), so
 * the imprecision of the first pass cannot reach the output distribution.  C2 / C4 (q = 3329) are single-pass. */
int orc_np_two_pass(uint64_t q, size_t n, double s) { return (double)q * sqrt((double)n) > s * 8192.0; }
void orc_gpv_set_two_pass(void* hv, int mode) { ((orc_gpv*)hv)->two_pass_mode = mode; }
int orc_gpv_two_pass(const void* hv) {
  const orc_gpv* h = (const orc_gpv*)hv;
  return h->two_pass_mode >= 0 ? h->two_pass_mode : orc_np_two_pass(h->gp.q, h->gp.n, h->s);
}

/* gpv.rs:152-161.  percall != 0: run the elimination on [A | u] for every call exactly as the reference does;
 * otherwise use the factored solver (same solution, see orc_solve_precompute). */
int orc_gpv_samp_p(const void* hv, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e,
                   int percall, int nthreads) {
  const orc_gpv* h = (const orc_gpv*)hv;
  size_t n = h->gp.n, m = h->m;
  uint64_t q = h->gp.q;
  int status = ORC_OK;
  const unsigned long cap0 = orc_sample_z_cap_hits();
  if (!percall && !h->has_solver) return ORC_ERR_NO_SOLUTION;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
  for (size_t b = 0; b < B; ++b) {
    uint64_t* sol = (uint64_t*)calloc(m, sizeof(uint64_t));
    int64_t* c = (int64_t*)malloc(m * sizeof(int64_t));
    if (percall) {
      int rc = orc_solve_gaussian_elimination(h->A, n, m, q, u + b * n, sol);     /* :153-156 */
      if (rc) status = rc;
    } else {
      for (size_t r = 0; r < n; ++r) {
        u128 acc = 0;
        for (size_t t = 0; t < n; ++t) acc = (acc + (u128)h->T[r * n + t] * (u[b * n + t] % q)) % q;
        sol[h->piv[r]] = (uint64_t)acc;
      }
    }
    for (size_t j = 0; j < m; ++j) c[j] = -(int64_t)sol[j];                         /* :158 center = -sol */
    orc_nearest_plane(h->St, h->Gt, h->G, h->norm2, m, h->s, seed, ORC_TAG_GPV, first_index + b, c);
    /* :160  sol + sample, sample = (-sol) - c_final */
    for (size_t j = 0; j < m; ++j) e[b * m + j] = -c[j];
    if (orc_gpv_two_pass(h)) {                                                      /* second pass from the short representative e1 = -c */
      for (size_t j = 0; j < m; ++j)
        if (c[j] >= (1ll << 53) || c[j] <= -(1ll << 53)) status = ORC_ERR_SAMPLER;
      orc_nearest_plane(h->St, h->Gt, h->G, h->norm2, m, h->s, seed, ORC_TAG_GPV2, first_index + b, c);
      for (size_t j = 0; j < m; ++j) e[b * m + j] = -c[j];
    }
    free(sol); free(c);
  }
  return (status == ORC_OK && orc_sample_z_cap_hits() != cap0) ? ORC_ERR_SAMPLER : status;      /* a draw that ended at the attempt cap (orc_sample_z) */
}

/* one preimage with the walk's own view of it (tests/test_oracle_centre_precision.py): the integer centre vector the FINAL pass started from
 * (-sol, or -e1 in two-pass mode), the centres c'_i it computed in doubles and the coefficients z_i it drew */
int orc_gpv_samp_p_trace(const void* hv, uint64_t seed, uint64_t index, const uint64_t* u, int64_t* e, int64_t* c_start, double* centres, int64_t* z) {
  const orc_gpv* h = (const orc_gpv*)hv;
  size_t n = h->gp.n, m = h->m;
  uint64_t q = h->gp.q;
  if (!h->has_solver) return ORC_ERR_NO_SOLUTION;
  const unsigned long cap0 = orc_sample_z_cap_hits();
  int64_t* c = (int64_t*)calloc(m, sizeof(int64_t));
  for (size_t r = 0; r < n; ++r) {
    u128 acc = 0;
    for (size_t t = 0; t < n; ++t) acc = (acc + (u128)h->T[r * n + t] * (u[t] % q)) % q;
    c[h->piv[r]] = -(int64_t)(uint64_t)acc;
  }
  if (orc_gpv_two_pass(h)) {
    orc_nearest_plane(h->St, h->Gt, h->G, h->norm2, m, h->s, seed, ORC_TAG_GPV, index, c);
    memcpy(c_start, c, m * sizeof(int64_t));
    orc_nearest_plane_trace(h->St, h->Gt, h->G, h->norm2, m, h->s, seed, ORC_TAG_GPV2, index, c, centres, z);
  } else {
    memcpy(c_start, c, m * sizeof(int64_t));
    orc_nearest_plane_trace(h->St, h->Gt, h->G, h->norm2, m, h->s, seed, ORC_TAG_GPV, index, c, centres, z);
  }
  for (size_t j = 0; j < m; ++j) e[j] = -c[j];
  free(c);
  return orc_sample_z_cap_hits() != cap0 ? ORC_ERR_SAMPLER : ORC_OK;
}

/* gpv.rs:113-116: D_{Z^m, s} centred at 0 */
int orc_gpv_samp_d(const void* hv, uint64_t seed, uint64_t first_index, size_t B, int64_t* e) {
  const orc_gpv* h = (const orc_gpv*)hv;
  const unsigned long cap0 = orc_sample_z_cap_hits();
  for (size_t b = 0; b < B; ++b)
    for (size_t i = 0; i < h->m; ++i) e[b * h->m + i] = orc_sample_z(seed, ORC_TAG_SAMPD, first_index + b, (uint32_t)i, 0.0, h->s);
  return orc_sample_z_cap_hits() != cap0 ? ORC_ERR_SAMPLER : ORC_OK;
}

/* gpv.rs:219-224 */
int orc_gpv_check_domain(const void* hv, size_t B, const int64_t* e, size_t len, uint8_t* ok) {
  const orc_gpv* h = (const orc_gpv*)hv;
  const double bound = (h->s * h->s) * (double)h->m;
  for (size_t b = 0; b < B; ++b) {
    if (len != h->m) { ok[b] = 0; continue; }
    u128 nn = 0;
    for (size_t i = 0; i < len; ++i) { i128 v = e[b * len + i]; nn += (u128)(v * v); }
    ok[b] = ((double)nn <= bound) ? 1 : 0;
  }
  return ORC_OK;
}

/* gpv.rs:190-193 */
int orc_gpv_f_a(const void* hv, size_t B, const int64_t* e, uint64_t* u) {
  const orc_gpv* h = (const orc_gpv*)hv;
  size_t n = h->gp.n, m = h->m;
  int status = ORC_OK;
  for (size_t b = 0; b < B; ++b) {
    uint8_t ok;
    orc_gpv_check_domain(h, 1, e + b * m, m, &ok);
    if (!ok) status = ORC_ERR_DOMAIN;
    for (size_t i = 0; i < n; ++i) {
      i128 acc = 0;
      for (size_t j = 0; j < m; ++j) acc += (i128)h->A[i * m + j] * e[b * m + j];
      i128 r = acc % (i128)h->gp.q;
      if (r < 0) r += h->gp.q;
      u[b * n + i] = (uint64_t)r;
    }
  }
  return status;
}

/* ------------------------------------------------------------------------------------------
 * rotation_matrix.rs:41-63 / :85-96
 * ---------------------------------------------------------------------------------------- */
void orc_rot_minus(const int64_t* vec, size_t n, int64_t* out, size_t ld, size_t col_off) {
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < n; ++j) {
      size_t k = i + j;
      if (k >= n) out[(k % n) * ld + col_off + j] = -vec[i];
      else out[k * ld + col_off + j] = vec[i];
    }
}
void orc_rot_minus_matrix(const int64_t* mat, size_t rows, size_t cols, int64_t* out) {
  int64_t* col = (int64_t*)malloc(rows * sizeof(int64_t));
  for (size_t c = 0; c < cols; ++c) {
    for (size_t r = 0; r < rows; ++r) col[r] = mat[r * cols + c];
    orc_rot_minus(col, rows, out, rows * cols, c * rows);
  }
  free(col);
}

/* ------------------------------------------------------------------------------------------
 * Ring variant: gadget_ring.rs, short_basis_ring.rs, gpv_ring.rs.  Polynomials are int64 coefficient arrays of
 * length n (degree < n), arithmetic in Z[X]/(X^n + 1) (common_moduli.rs:41-48) unless stated otherwise.
 * ---------------------------------------------------------------------------------------- */
/* out = a * b in Z[X]/(X^n+1) (PolyOverZ product followed by reduce_by_poly, short_basis_ring.rs:72-77) */
static void poly_mul_negacyclic(const int64_t* a, const int64_t* b, size_t n, int64_t* out) {
  int64_t* t = (int64_t*)calloc(n, sizeof(int64_t));
  for (size_t i = 0; i < n; ++i) {
    if (!a[i]) continue;
    for (size_t j = 0; j < n; ++j) {
      size_t d = i + j;
      if (d >= n) t[d - n] -= a[i] * b[j];
      else t[d] += a[i] * b[j];
    }
  }
  memcpy(out, t, n * sizeof(int64_t));
  free(t);
}

/* gen_trapdoor_ring_lwe (gadget_ring.rs:62-81) with a_bar, r, e drawn from the Philox streams:
 *   a_bar uniform in Z_q^n (gpv_ring.rs:92-94), r_j, e_j <- D_{Z,s_td}^n (SampleZ, trapdoor_distribution.rs:112-122)
 *   A = [1 | a_bar | g_j - (a_bar r_j + e_j)] mod (X^n+1, q);  a: (k+2) x n, r, e: k x n */
int orc_ring_trap_gen(const orc_gadget_params* gp, double s_td, uint64_t seed, uint64_t* a, int64_t* r, int64_t* e) {
  size_t n = gp->n, k = gp->k;
  uint64_t q = gp->q;
  for (size_t j = 0; j < k; ++j)
    for (size_t c = 0; c < n; ++c) {
      r[j * n + c] = orc_sample_z(seed, ORC_TAG_RING_R, 0, (uint32_t)(j * n + c), 0.0, s_td);
      e[j * n + c] = orc_sample_z(seed, ORC_TAG_RING_E, 0, (uint32_t)(j * n + c), 0.0, s_td);
    }
  memset(a, 0, (k + 2) * n * sizeof(uint64_t));
  a[0] = 1 % q;
  int64_t* abar = (int64_t*)malloc(n * sizeof(int64_t));
  for (size_t c = 0; c < n; ++c) { a[n + c] = orc_uniform_mod(seed, ORC_TAG_RING_A, (uint32_t)c, 0, q); abar[c] = (int64_t)a[n + c]; }
  int64_t* prod = (int64_t*)malloc(n * sizeof(int64_t));
  uint64_t g = 1 % q;
  for (size_t j = 0; j < k; ++j) {
    /* a_bar * r_j in Z[X]/(X^n+1): entries up to n q |r| < 2^63 for the supported q < 2^31 */
    i128* acc = (i128*)calloc(n, sizeof(i128));
    for (size_t x = 0; x < n; ++x)
      for (size_t y = 0; y < n; ++y) {
        size_t d = x + y;
        i128 p = (i128)abar[x] * r[j * n + y];
        if (d >= n) acc[d - n] -= p; else acc[d] += p;
      }
    for (size_t c = 0; c < n; ++c) {
      i128 v = (c == 0 ? (i128)g : 0) - (acc[c] + e[j * n + c]);
      i128 m = v % (i128)q;
      if (m < 0) m += q;
      a[(2 + j) * n + c] = (uint64_t)m;
    }
    free(acc);
    g = g_mulmod(g, gp->base % q, q);
  }
  free(abar); free(prod);
  return ORC_OK;
}

/* compute_s (short_basis_ring.rs:142-166): k x k integers (constant polynomials) */
int orc_ring_compute_s(const orc_gadget_params* gp, int64_t* sk) { return orc_short_basis_gadget_block(gp, sk); }

/* find_solution_gadget_ring (gadget_ring.rs:145-166): digits of every coefficient; out: k x n (poly i = digit i) */
int orc_find_solution_gadget_ring(const uint64_t* u, size_t n, uint64_t q, uint64_t k, uint64_t base, int64_t* out) {
  int64_t* d = (int64_t*)malloc(k * sizeof(int64_t));
  for (size_t j = 0; j < n; ++j) {
    int rc = orc_find_solution_gadget_vec(u[j], q, k, base, d);
    if (rc) { free(d); return rc; }
    for (size_t i = 0; i < k; ++i) out[i * n + j] = d[i];           /* index i + j*k of the classical solution (:160) */
  }
  free(d);
  return ORC_OK;
}

/* gen_sa_l (short_basis_ring.rs:82-91): (k+2) x (k+2) matrix of polynomials [1 0 first ; 0 1 second ; 0 0 I_k];
 * the caller passes (e, r) as gen_short_basis_for_trapdoor_ring does (:70).  out[(row*(k+2)+col)*n + coeff]. */
void orc_ring_gen_sa_l(const int64_t* first, const int64_t* second, size_t n, size_t k, int64_t* out) {
  size_t K = k + 2;
  memset(out, 0, K * K * n * sizeof(int64_t));
  for (size_t d = 0; d < K; ++d) out[(d * K + d) * n] = 1;
  for (size_t j = 0; j < k; ++j)
    for (size_t c = 0; c < n; ++c) {
      out[(0 * K + 2 + j) * n + c] = first[j * n + c];
      out[(1 * K + 2 + j) * n + c] = second[j * n + c];
    }
}

/* gen_sa_r (short_basis_ring.rs:96-124): (k+2) x n(k+2) polynomials, [pd (x) [0; S'] | pd (x) [I_2; W]], pd = [X^0..X^{n-1}].
 * a: (k+2) x n residues.  out[(row*cols + col)*n + coeff], cols = n(k+2); entries already reduced mod X^n+1. */
int orc_ring_gen_sa_r(const orc_gadget_params* gp, const uint64_t* a, int64_t* out) {
  size_t n = gp->n, k = gp->k, K = k + 2, cols = n * K;
  uint64_t q = gp->q;
  int64_t* sk = (int64_t*)malloc(k * k * sizeof(int64_t));
  orc_ring_compute_s(gp, sk);
  u128 bk = 1; int pw = 1;
  for (uint64_t i = 0; i < k; ++i) { bk *= gp->base; if (bk > q) { pw = 0; break; } }
  int reversed = pw && bk == q;                                                    /* :110-112 */
  /* W = [w_0 | w_1], w_c = digits of -a_c (compute_w, :128-139) */
  int64_t* W = (int64_t*)malloc(2 * k * n * sizeof(int64_t));
  uint64_t* neg = (uint64_t*)malloc(n * sizeof(uint64_t));
  for (int c = 0; c < 2; ++c) {
    for (size_t j = 0; j < n; ++j) { uint64_t v = a[c * n + j] % q; neg[j] = v ? q - v : 0; }
    int rc = orc_find_solution_gadget_ring(neg, n, q, k, gp->base, W + (size_t)c * k * n);
    if (rc) { free(sk); free(W); free(neg); return rc; }
  }
  memset(out, 0, K * cols * n * sizeof(int64_t));
  int64_t* tmp = (int64_t*)malloc(n * sizeof(int64_t));
  for (size_t i = 0; i < n; ++i) {
    /* X^i * S' block: columns i*k + c, rows 2..k+1 */
    for (size_t c = 0; c < k; ++c)
      for (size_t t = 0; t < k; ++t) {
        int64_t v = sk[t * k + (reversed ? (k - 1 - c) : c)];
        if (v) out[((2 + t) * cols + i * k + c) * n + i] = v;                     /* v * X^i, i < n: no wrap */
      }
    /* X^i * [I_2; W]: columns kn + 2i + c */
    for (int c = 0; c < 2; ++c) {
      size_t col = k * n + 2 * i + c;
      out[((size_t)c * cols + col) * n + i] = 1;
      for (size_t t = 0; t < k; ++t) {
        const int64_t* wp = W + ((size_t)c * k + t) * n;
        memset(tmp, 0, n * sizeof(int64_t));
        for (size_t j = 0; j < n; ++j) {                                          /* X^i * w mod X^n+1 */
          size_t d = i + j;
          if (d >= n) tmp[d - n] -= wp[j]; else tmp[d] += wp[j];
        }
        memcpy(out + ((2 + t) * cols + col) * n, tmp, n * sizeof(int64_t));
      }
    }
  }
  free(sk); free(W); free(neg); free(tmp);
  return ORC_OK;
}

/* gen_short_basis_for_trapdoor_ring (short_basis_ring.rs:64-79) followed by the coefficient embedding used by
 * MatPolyOverZ::sample_d (gpv_ring.rs:204-210): basis_t is d x d (d = n(k+2)), row c = embedding of column c of
 * sa_l * sa_r mod X^n+1, entry index polyrow*n + coeff. */
int orc_ring_short_basis_t(const orc_gadget_params* gp, const uint64_t* a, const int64_t* r, const int64_t* e, int32_t* basis_t) {
  size_t n = gp->n, k = gp->k, K = k + 2, d = n * K;
  int64_t* sal = (int64_t*)malloc(K * K * n * sizeof(int64_t));
  int64_t* sar = (int64_t*)malloc(K * d * n * sizeof(int64_t));
  orc_ring_gen_sa_l(e, r, n, k, sal);                                              /* :70 gen_sa_l(e, r) */
  int rc = orc_ring_gen_sa_r(gp, a, sar);                                          /* :71 */
  if (rc) { free(sal); free(sar); return rc; }
  int64_t* acc = (int64_t*)malloc(n * sizeof(int64_t));
  int64_t* prod = (int64_t*)malloc(n * sizeof(int64_t));
  for (size_t col = 0; col < d; ++col)
    for (size_t row = 0; row < K; ++row) {
      memset(acc, 0, n * sizeof(int64_t));
      for (size_t t = 0; t < K; ++t) {
        const int64_t* x = sal + (row * K + t) * n;
        const int64_t* y = sar + (t * d + col) * n;
        int zero = 1;
        for (size_t c = 0; c < n && zero; ++c) if (x[c]) zero = 0;
        if (zero) continue;
        poly_mul_negacyclic(x, y, n, prod);                                        /* :72-77 product then reduce_by_poly */
        for (size_t c = 0; c < n; ++c) acc[c] += prod[c];
      }
      for (size_t c = 0; c < n; ++c) basis_t[col * d + row * n + c] = (int32_t)acc[c];
    }
  free(sal); free(sar); free(acc); free(prod);
  return ORC_OK;
}

/* rot^-(iota(a)) (gpv_ring.rs:172-178): n x n(k+2), block j = rot_minus(coefficients of a_j) */
void orc_ring_embed_a(const uint64_t* a, size_t n, size_t K, uint64_t q, uint64_t* A_emb) {
  size_t d = n * K;
  for (size_t j = 0; j < K; ++j)
    for (size_t i = 0; i < n; ++i) {
      uint64_t v = a[j * n + i] % q;
      for (size_t l = 0; l < n; ++l) {
        size_t row = i + l;
        if (row >= n) A_emb[(row - n) * d + j * n + l] = v ? q - v : 0;
        else A_emb[row * d + j * n + l] = v;
      }
    }
}
