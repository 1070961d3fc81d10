/*
 * psf_faithful_gmp.c -- CPU ORACLE, "faithful mode" (TEST / MEASUREMENT INFRASTRUCTURE ONLY; never linked into the product).
 *
 * The reference computes samp_p on FLINT-backed exact types: MatQ (fmpq) for sqrt(Sigma_2) and the DENSE nk x nk Gram-Schmidt matrix of
 * I_n (x) S_k (mp_perturbation.rs:233-234), MatZ / MatZq (fmpz) for everything else, and it rebuilds [R; I] on every call
 * (:328-333).  This file restates one samp_p call in that style with GMP (mpq / mpz) so that the cost gap between reference-style
 * arithmetic and the flat-array port (psf_oracle.c) can be measured at the sizes the reference's own benches use (benches/psf.rs:27,52,79)
 * -- BASELINE.md section 3.  Random decisions are the port's (same Philox streams, same SampleZ): a rational centre is rounded to a double
 * once, where the port rounds after every fma, so individual samples may differ in rare borderline cases; every output is checked through
 * the invariants (A e = u, check_domain), not bitwise.
 */
#include <gmp.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "psf_oracle.h"

static void mpq_set_double(mpq_t q, double d) { mpq_set_d(q, d); }

/* mp_perturbation.rs:304-336, one call */
int orc_faithful_psfp_samp_p(const orc_psfp* h, uint64_t seed, uint64_t index, const uint64_t* u, int64_t* e) {
  const orc_gadget_params* gp = &h->gp;
  const size_t n = gp->n, k = gp->k, mb = gp->m_bar, w = n * k, m = h->m;
  if (!h->L) return ORC_ERR_PARAM;
  mpq_t acc, t1, t2;
  mpq_inits(acc, t1, t2, NULL);
  /* :315 sample_d_common_non_spherical: d <- N(0,1)^m, x = sqrt(Sigma_2) d over Q, p_i <- D_{Z,r,x_i} */
  mpq_t* d = (mpq_t*)malloc(m * sizeof(mpq_t));
  for (size_t j = 0; j < m; ++j) { mpq_init(d[j]); mpq_set_double(d[j], orc_sample_normal(seed, index, (uint32_t)j)); }
  int64_t* p = (int64_t*)malloc(m * sizeof(int64_t));
  for (size_t i = 0; i < m; ++i) {
    const double* li = h->L + i * (i + 1) / 2;
    mpq_set_ui(acc, 0, 1);
    for (size_t j = 0; j <= i; ++j) {            /* MatQ * MatQ: every entry an fmpq product and sum (dense row: zeros above the diagonal included) */
      mpq_set_double(t1, li[j]);
      mpq_mul(t2, t1, d[j]);
      mpq_add(acc, acc, t2);
    }
    for (size_t j = i + 1; j < m; ++j) { mpq_set_ui(t1, 0, 1); mpq_mul(t2, t1, d[j]); mpq_add(acc, acc, t2); }
    p[i] = orc_sample_z(seed, ORC_TAG_PERTURB, index, (uint32_t)i, mpq_get_d(acc), h->r);
  }
  /* :318 v = u - A p over Z_q (fmpz_mod) */
  mpz_t za, zp, zq, zt;
  mpz_inits(za, zp, zq, zt, NULL);
  mpz_set_ui(zq, 0); mpz_import(zq, 1, -1, sizeof(uint64_t), 0, 0, &gp->q);
  uint64_t* v = (uint64_t*)malloc(n * sizeof(uint64_t));
  for (size_t i = 0; i < n; ++i) {
    mpz_set_ui(za, 0);
    for (size_t j = 0; j < m; ++j) {
      mpz_import(zt, 1, -1, sizeof(uint64_t), 0, 0, &h->A[i * m + j]);
      mpz_set_si(zp, (long)p[j]);
      mpz_addmul(za, zt, zp);
    }
    mpz_import(zt, 1, -1, sizeof(uint64_t), 0, 0, &u[i]);
    mpz_sub(za, zt, za);
    mpz_mod(za, za, zq);
    uint64_t out = 0; size_t cnt = 0;
    mpz_export(&out, &cnt, -1, sizeof(uint64_t), 0, 0, za);
    v[i] = cnt ? out : 0;
  }
  /* :321-326 randomized_nearest_plane_gadget on the DENSE nk x nk basis and Gram-Schmidt matrix (:233-234, :183-190) */
  int64_t* x = (int64_t*)malloc(w * sizeof(int64_t));
  int rc = orc_find_solution_gadget_mat(v, n, 1, gp->q, k, gp->base, x);
  if (rc) return rc;
  mpq_t* S = (mpq_t*)malloc(w * w * sizeof(mpq_t));      /* basis, columns = vectors */
  mpq_t* G = (mpq_t*)malloc(w * w * sizeof(mpq_t));      /* GSO */
  for (size_t a = 0; a < w * w; ++a) { mpq_init(S[a]); mpq_init(G[a]); }
  for (size_t blk = 0; blk < n; ++blk)
    for (size_t i = 0; i < k; ++i)
      for (size_t j = 0; j < k; ++j) {
        mpq_set_si(S[(blk * k + i) * w + blk * k + j], (long)h->Sk[i * k + j], 1);
        mpq_set_double(G[(blk * k + i) * w + blk * k + j], h->Sk_gso[i * k + j]);
      }
  mpq_t* c = (mpq_t*)malloc(w * sizeof(mpq_t));
  for (size_t t = 0; t < w; ++t) { mpq_init(c[t]); mpq_set_si(c[t], (long)-x[t], 1); }
  const double sG = h->r * sqrt((double)(gp->base * gp->base + 1));
  mpq_t nn;
  mpq_init(nn);
  for (size_t ii = w; ii-- > 0;) {
    mpq_set_ui(acc, 0, 1); mpq_set_ui(nn, 0, 1);
    for (size_t t = 0; t < w; ++t) {
      mpq_mul(t2, c[t], G[t * w + ii]); mpq_add(acc, acc, t2);
      mpq_mul(t2, G[t * w + ii], G[t * w + ii]); mpq_add(nn, nn, t2);
    }
    mpq_div(acc, acc, nn);
    const int64_t zi = orc_sample_z(seed, ORC_TAG_GADGET, index, (uint32_t)ii, mpq_get_d(acc), sG / sqrt(mpq_get_d(nn)));
    mpq_set_si(t1, (long)zi, 1);
    for (size_t t = 0; t < w; ++t) { mpq_mul(t2, t1, S[t * w + ii]); mpq_sub(c[t], c[t], t2); }
  }
  int64_t* z = (int64_t*)malloc(w * sizeof(int64_t));
  for (size_t t = 0; t < w; ++t) z[t] = -(int64_t)mpq_get_d(c[t]);
  /* :328-335 T = [R; I] rebuilt per call, e = p + T z over Z */
  mpz_t* T = (mpz_t*)malloc(m * w * sizeof(mpz_t));
  for (size_t a = 0; a < m * w; ++a) mpz_init(T[a]);
  for (size_t i = 0; i < mb; ++i)
    for (size_t cc = 0; cc < w; ++cc) mpz_set_si(T[i * w + cc], (long)h->R[i * w + cc]);
  for (size_t cc = 0; cc < w; ++cc) mpz_set_ui(T[(mb + cc) * w + cc], 1);
  for (size_t i = 0; i < m; ++i) {
    mpz_set_si(za, (long)p[i]);
    for (size_t cc = 0; cc < w; ++cc) { mpz_set_si(zp, (long)z[cc]); mpz_addmul(za, T[i * w + cc], zp); }
    e[i] = (int64_t)mpz_get_si(za);
  }
  for (size_t a = 0; a < m * w; ++a) mpz_clear(T[a]);
  for (size_t a = 0; a < w * w; ++a) { mpq_clear(S[a]); mpq_clear(G[a]); }
  for (size_t t = 0; t < w; ++t) mpq_clear(c[t]);
  for (size_t j = 0; j < m; ++j) mpq_clear(d[j]);
  free(T); free(S); free(G); free(c); free(d); free(p); free(v); free(x); free(z);
  mpq_clears(acc, t1, t2, nn, NULL);
  mpz_clears(za, zp, zq, zt, NULL);
  return ORC_OK;
}

/* gpv.rs:152-161, one call: per-call Gaussian elimination (the port's routine: word arithmetic mod q), then sample_d_precomputed_gso with the dense
 * m x m rational Gram-Schmidt matrix */
int orc_faithful_gpv_samp_p(const uint64_t* A, size_t n, size_t m, uint64_t q, const int32_t* basis_t, const double* gso_t, double s,
                            uint64_t seed, uint64_t index, const uint64_t* u, int64_t* e) {
  uint64_t* sol = (uint64_t*)calloc(m, sizeof(uint64_t));
  int rc = orc_solve_gaussian_elimination(A, n, m, q, u, sol);
  if (rc) { free(sol); return rc; }
  mpq_t* G = (mpq_t*)malloc(m * m * sizeof(mpq_t));
  mpq_t* c = (mpq_t*)malloc(m * sizeof(mpq_t));
  for (size_t a = 0; a < m * m; ++a) { mpq_init(G[a]); mpq_set_d(G[a], gso_t[a]); }
  for (size_t j = 0; j < m; ++j) { mpq_init(c[j]); mpq_set_si(c[j], -(long)sol[j], 1); }
  mpq_t acc, nn, t1, t2;
  mpq_inits(acc, nn, t1, t2, NULL);
  for (size_t ii = m; ii-- > 0;) {
    mpq_set_ui(acc, 0, 1); mpq_set_ui(nn, 0, 1);
    for (size_t j = 0; j < m; ++j) {
      mpq_mul(t2, c[j], G[ii * m + j]); mpq_add(acc, acc, t2);
      mpq_mul(t2, G[ii * m + j], G[ii * m + j]); mpq_add(nn, nn, t2);
    }
    mpq_div(acc, acc, nn);
    const int64_t z = orc_sample_z(seed, ORC_TAG_GPV, index, (uint32_t)ii, mpq_get_d(acc), s / sqrt(mpq_get_d(nn)));
    mpq_set_si(t1, (long)z, 1);
    for (size_t j = 0; j < m; ++j) { mpq_set_si(t2, (long)basis_t[ii * m + j], 1); mpq_mul(t2, t2, t1); mpq_sub(c[j], c[j], t2); }
  }
  for (size_t j = 0; j < m; ++j) e[j] = -(int64_t)mpq_get_d(c[j]);
  for (size_t a = 0; a < m * m; ++a) mpq_clear(G[a]);
  for (size_t j = 0; j < m; ++j) mpq_clear(c[j]);
  mpq_clears(acc, nn, t1, t2, NULL);
  free(G); free(c); free(sol);
  return ORC_OK;
}
