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
 *   c' = <c, b~_i> / ||b~_i||^2 ; z <- D_{Z, s/||b~_i||, c'} ; c -= z b_i.      Returns c in place (the sample is
 *   (initial c) - (final c)).
 * Contract for the projection: 256 partial fma chains over j = t, t+256, ... (ascending), combined per group of
 * 64 by the xor butterfly 32,16,8,4,2,1 and then ((w0+w1)+(w2+w3)).
 * basis_t / gso_t are TRANSPOSED: row i holds basis vector i (column i of the reference's matrices).
 * ---------------------------------------------------------------------------------------- */
double orc_dot256(const int64_t* c, const double* g, size_t dim) {
  double p[256];
  for (int t = 0; t < 256; ++t) {
    double acc = 0.0;
    for (size_t j = (size_t)t; j < dim; j += 256) acc = fma((double)c[j], g[j], acc);
    p[t] = acc;
  }
  double w[4];
  for (int wv = 0; wv < 4; ++wv) {
    double a[64], b2[64];
    memcpy(a, p + 64 * wv, sizeof(a));
    for (int off = 32; off >= 1; off >>= 1) {
      for (int l = 0; l < 64; ++l) b2[l] = a[l] + a[l ^ off];
      memcpy(a, b2, sizeof(a));
    }
    w[wv] = a[0];
  }
  return (w[0] + w[1]) + (w[2] + w[3]);
}

void orc_nearest_plane(const int32_t* basis_t, const double* gso_t, const double* norm2, size_t dim, double s,
                       uint64_t seed, uint32_t tag, uint64_t index, int64_t* c) {
  for (size_t ii = dim; ii-- > 0;) {
    const double dot = orc_dot256(c, gso_t + ii * dim, dim);
    const double c2 = dot / norm2[ii];
    const double s2 = s / sqrt(norm2[ii]);
    const int64_t z = orc_sample_z(seed, tag, index, (uint32_t)ii, c2, s2);
    if (z) {
      const int32_t* bi = basis_t + ii * dim;
      for (size_t j = 0; j < dim; ++j) c[j] -= z * (int64_t)bi[j];
    }
  }
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
  size_t* piv;        /* n pivot columns */
  uint64_t* T;        /* n x n solve operator */
  int has_solver;
} orc_gpv;

static orc_gpv* gpv_new_impl(const orc_gadget_params* gp, double s) {
  if (!gp || gp->n < 1 || gp->q <= 1 || !(s > 0)) return NULL;
  orc_gpv* h = (orc_gpv*)calloc(1, sizeof(orc_gpv));
  h->gp = *gp; h->s = s;
  size_t w = gp->n * gp->k;
  h->m = gp->m_bar + w;
  h->A = (uint64_t*)calloc(gp->n * h->m, sizeof(uint64_t));
  h->R = (int8_t*)calloc(gp->m_bar * w, 1);
  h->St = (int32_t*)calloc(h->m * h->m, sizeof(int32_t));
  h->Gt = (double*)calloc(h->m * h->m, sizeof(double));
  h->norm2 = (double*)calloc(h->m, sizeof(double));
  h->piv = (size_t*)calloc(gp->n, sizeof(size_t));
  h->T = (uint64_t*)calloc(gp->n * gp->n, sizeof(uint64_t));
  return h;
}
void* orc_gpv_new(const orc_gadget_params* gp, double s) { return gpv_new_impl(gp, s); }
void orc_gpv_free(void* hv) {
  orc_gpv* h = (orc_gpv*)hv;
  if (!h) return;
  free(h->A); free(h->R); free(h->St); free(h->Gt); free(h->norm2); free(h->piv); free(h->T); free(h);
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
  int rc = orc_solve_precompute(h->A, h->gp.n, m, h->gp.q, h->piv, h->T);
  h->has_solver = (rc == ORC_OK);
  return rc;
}

/* Gram-Schmidt on the rows of St (= columns of S_A), MatQ::gso at gpv.rs:88 */
static void gso_rows(const int32_t* St, size_t m, double* Gt) {
  double* norm2 = (double*)malloc(m * sizeof(double));
  for (size_t i = 0; i < m; ++i) {
    double* gi = Gt + i * m;
    const int32_t* bi = St + i * m;
    for (size_t j = 0; j < m; ++j) gi[j] = (double)bi[j];
    for (size_t l = 0; l < i; ++l) {
      const double* gl = Gt + l * m;
      double num = 0.0;
      for (size_t j = 0; j < m; ++j) num = fma((double)bi[j], gl[j], num);
      const double mu = num / norm2[l];
      for (size_t j = 0; j < m; ++j) gi[j] = fma(-mu, gl[j], gi[j]);
    }
    double nn = 0.0;
    for (size_t j = 0; j < m; ++j) nn = fma(gi[j], gi[j], nn);
    norm2[i] = nn;
  }
  free(norm2);
}

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

/* gpv.rs:152-161.  percall != 0: run the elimination on [A | u] for every call exactly as the reference does;
 * otherwise use the factored solver (same solution, see orc_solve_precompute). */
int orc_gpv_samp_p(const void* hv, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e,
                   int percall, int nthreads) {
  const orc_gpv* h = (const orc_gpv*)hv;
  size_t n = h->gp.n, m = h->m;
  uint64_t q = h->gp.q;
  int status = ORC_OK;
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
    orc_nearest_plane(h->St, h->Gt, h->norm2, m, h->s, seed, ORC_TAG_GPV, first_index + b, c);
    /* :160  sol + sample, sample = (-sol) - c_final */
    for (size_t j = 0; j < m; ++j) e[b * m + j] = -c[j];
    free(sol); free(c);
  }
  return status;
}

/* gpv.rs:113-116: D_{Z^m, s} centred at 0 */
int orc_gpv_samp_d(const void* hv, uint64_t seed, uint64_t first_index, size_t B, int64_t* e) {
  const orc_gpv* h = (const orc_gpv*)hv;
  for (size_t b = 0; b < B; ++b)
    for (size_t i = 0; i < h->m; ++i) e[b * h->m + i] = orc_sample_z(seed, ORC_TAG_SAMPD, first_index + b, (uint32_t)i, 0.0, h->s);
  return ORC_OK;
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
