/* The CPU oracle (test infrastructure) itself under AddressSanitizer + UndefinedBehaviorSanitizer: one small flow per scheme
 * with the reference's invariants (A e = u, check_domain).  Built from oracle/*.c with gcc -fsanitize=address,undefined. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../oracle/psf_oracle.h"

#define CHECK(c) do { if (!(c)) { printf("CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); exit(1); } } while (0)

static void perturbation(uint64_t n, uint64_t q, double r, double s) {
  orc_gadget_params gp;
  CHECK(orc_gadget_params_default(n, q, &gp) == 0);
  orc_psfp* h = orc_psfp_new(&gp, r, s);
  CHECK(h && orc_psfp_trap_gen(h, 5) == 0);
  const size_t m = gp.m_bar + gp.n * gp.k, B = 3;
  uint64_t* u = (uint64_t*)malloc(B * n * sizeof(uint64_t));
  uint64_t* u2 = (uint64_t*)malloc(B * n * sizeof(uint64_t));
  int64_t* e = (int64_t*)malloc(B * m * sizeof(int64_t));
  uint8_t ok[3];
  for (size_t i = 0; i < B * n; ++i) u[i] = orc_uniform_mod(9, 7, (uint32_t)i, 0, q);
  CHECK(orc_psfp_samp_p(h, 11, 4, B, u, e, 1) == 0);
  CHECK(orc_psfp_f_a(h, B, e, u2) == 0);
  for (size_t i = 0; i < B * n; ++i) CHECK(u2[i] == u[i]);
  CHECK(orc_psfp_check_domain(h, B, e, m, ok) == 0 && ok[0] && ok[1] && ok[2]);
  CHECK(orc_psfp_samp_d(h, 3, 0, B, e) == 0);
  CHECK(orc_psfp_check_domain(h, B, e, m, ok) == 0 && ok[0] && ok[1] && ok[2]);
  free(u); free(u2); free(e);
  orc_psfp_free(h);
}

static void gpv(uint64_t n, uint64_t q, double s) {
  orc_gadget_params gp;
  CHECK(orc_gadget_params_default(n, q, &gp) == 0);
  void* h = orc_gpv_new(&gp, s);
  CHECK(h && orc_gpv_trap_gen(h, 6) == 0);
  const size_t m = orc_gpv_m(h), B = 2;
  uint64_t* u = (uint64_t*)malloc(B * n * sizeof(uint64_t));
  uint64_t* u2 = (uint64_t*)malloc(B * n * sizeof(uint64_t));
  int64_t* e = (int64_t*)malloc(B * m * sizeof(int64_t));
  for (size_t i = 0; i < B * n; ++i) u[i] = orc_uniform_mod(2, 7, (uint32_t)i, 0, q);
  CHECK(orc_gpv_samp_p(h, 13, 0, B, u, e, 0, 1) == 0);
  CHECK(orc_gpv_f_a(h, B, e, u2) == 0);
  for (size_t i = 0; i < B * n; ++i) CHECK(u2[i] == u[i]);
  free(u); free(u2); free(e);
  orc_gpv_free(h);
}

static void ring(uint64_t n, uint64_t q) {
  orc_gadget_params gp;
  CHECK(orc_gadget_params_ring_default(n, q, &gp) == 0);
  const size_t k = gp.k, K = k + 2, d = K * n;
  uint64_t* a = (uint64_t*)malloc(K * n * sizeof(uint64_t));
  int64_t* r = (int64_t*)malloc(k * n * sizeof(int64_t));
  int64_t* e = (int64_t*)malloc(k * n * sizeof(int64_t));
  int32_t* bt = (int32_t*)malloc(d * d * sizeof(int32_t));
  uint64_t* A = (uint64_t*)malloc(n * d * sizeof(uint64_t));
  CHECK(orc_ring_trap_gen(&gp, 1.005, 3, a, r, e) == 0);
  CHECK(orc_ring_short_basis_t(&gp, a, r, e, bt) == 0);
  orc_ring_embed_a(a, n, K, q, A);
  for (size_t i = 0; i < n; ++i)
    for (size_t c = 0; c < d; ++c) {
      __int128 acc = 0;
      for (size_t t = 0; t < d; ++t) acc += (__int128)A[i * d + t] * bt[c * d + t];
      CHECK((uint64_t)(((acc % (__int128)q) + q) % q) == 0);
    }
  free(a); free(r); free(e); free(bt); free(A);
}

int main(void) {
  perturbation(5, 32, 2.5, 25.0);
  perturbation(3, 125, 2.0, 40.0);
  perturbation(2, 1ull << 60, 2.0, 70.0);
  gpv(4, 23, 12.0);
  gpv(5, 256, 10.0);
  ring(8, 17);
  ring(4, 16);
  for (int i = 0; i < 2000; ++i) {                      /* narrow and wide SampleZ words, far-off centres */
    (void)orc_sample_z(1, 4, (uint64_t)i, 3, -1e9 + i * 0.37, 4.5);
    (void)orc_sample_z(1, 4, (uint64_t)i, 3, 0.5 * i, 700.0);
  }
  printf("ORACLE_SANITIZE_OK\n");
  return 0;
}
