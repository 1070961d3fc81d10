// psf_mi355x.hpp -- C++ host-side mirror of the reference's `PSF` trait (src/primitive/psf.rs:39-81) over the C ABI of
// psf_mi355x.h.  Header-only; link with -lpsf_mi355x.  The reference is Rust; no Rust toolchain exists in the build image,
// so this is the compiled-language face of the drop-in (INTEGRATION.md shows the equivalent Rust shim).
//
//   trait PSF { type A; type Trapdoor; type Domain; type Range;
//               fn trap_gen(&self) -> (A, Trapdoor);  fn samp_d(&self) -> Domain;
//               fn samp_p(&self, a, r, u) -> Domain;   fn f_a(&self, a, sigma) -> Range;  fn check_domain(&self, sigma) -> bool; }
//
// Differences that the ABI imposes and this mirror keeps explicit:
//   * randomness is seeded (the trait has no seed, psf.rs:48-80): every sampling method takes (seed, first_index);
//   * a call may carry B rows (B independent reference calls); vectors are flat row-major std::vector;
//   * the key lives in the handle (device memory): samp_p / f_a use the key of the last trap_gen / load_key, so the
//     `a` and `r` arguments of the trait are implicit;
//   * where the reference panics, PsfError is thrown.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "psf_mi355x.h"

namespace psf_mi355x {

struct PsfError : std::runtime_error {
  psf_status status;
  PsfError(psf_status s, const char* where) : std::runtime_error(std::string(where) + ": " + psf_status_string(s)), status(s) {}
};
inline void check(psf_status s, const char* where) { if (s != PSF_OK) throw PsfError(s, where); }

// GadgetParameters::init_default (gadget_parameters.rs:113-133) / GadgetParametersRing::init_default (:165-185)
inline psf_gadget_params gadget_parameters_default(uint64_t n, uint64_t q) {
  psf_gadget_params gp;
  check(psf_gadget_params_default(n, q, &gp), "GadgetParameters::init_default");
  return gp;
}
inline psf_gadget_params gadget_parameters_ring_default(uint64_t n, uint64_t q) {
  psf_gadget_params gp;
  check(psf_gadget_params_ring_default(n, q, &gp), "GadgetParametersRing::init_default");
  return gp;
}

using MatZq = std::vector<uint64_t>;   // least non-negative residues, row-major
using MatZ = std::vector<int64_t>;

// ---- PSFPerturbation (mp_perturbation.rs:57-62, :193-403) -------------------------------------------------------------
class PSFPerturbation {
 public:
  struct Trapdoor { std::vector<int8_t> R; std::vector<double> sqrt_sigma_2; MatZ Sk; std::vector<double> Sk_gso; };   // :195

  PSFPerturbation(const psf_gadget_params& gp, double r, double s, int device = 0) : gp_(gp) {
    psfp_params p{gp, r, s, device, 0};
    check(psfp_create(&p, &h_), "PSFPerturbation");
  }
  ~PSFPerturbation() { psfp_destroy(h_); }
  PSFPerturbation(const PSFPerturbation&) = delete;
  PSFPerturbation& operator=(const PSFPerturbation&) = delete;

  size_t n() const { return gp_.n; }
  size_t m() const { return psfp_m(h_); }

  std::pair<MatZq, Trapdoor> trap_gen(uint64_t seed) {                                          // :221-244
    check(psfp_trap_gen(h_, seed), "trap_gen");
    const size_t mm = m(), w = gp_.n * gp_.k;
    MatZq A(gp_.n * mm);
    Trapdoor td;
    td.R.resize(gp_.m_bar * w); td.sqrt_sigma_2.resize(mm * (mm + 1) / 2); td.Sk.resize(gp_.k * gp_.k); td.Sk_gso.resize(gp_.k * gp_.k);
    check(psfp_export_key(h_, A.data(), td.R.data(), td.sqrt_sigma_2.data()), "export_key");
    check(psfp_export_gadget_basis(h_, td.Sk.data(), td.Sk_gso.data()), "export_gadget_basis");
    return {std::move(A), std::move(td)};
  }
  void load_key(const MatZq& A, const Trapdoor& td) { check(psfp_load_key(h_, A.data(), td.R.data(), td.sqrt_sigma_2.data()), "load_key"); }
  void load_public_key(const MatZq& A) { check(psfp_load_key(h_, A.data(), nullptr, nullptr), "load_key"); }      // a verifier's handle: f_a / check_domain only
  // (A, R) without a factor and without computing one: what compute_sqrt_sigma_2 (a pure function of mat_r and mat_sigma, :111) starts from; A may be null
  void load_trapdoor(const std::vector<int8_t>& R, const MatZq* A = nullptr) { check(psfp_load_trapdoor(h_, A ? A->data() : nullptr, R.data()), "load_trapdoor"); }
  // compute_sqrt_sigma_2 (:111-139): Sigma = s_cov^2 I, or any symmetric covariance as its packed lower triangle (row i: i + 1 entries)
  void compute_sqrt_sigma_2(double s_cov) { check(psfp_compute_sqrt_sigma_2(h_, s_cov), "compute_sqrt_sigma_2"); }
  void compute_sqrt_sigma_2(const std::vector<double>& sigma_lower_packed) { check(psfp_compute_sqrt_sigma_2_dense(h_, sigma_lower_packed.data()), "compute_sqrt_sigma_2"); }
  MatZ samp_d(uint64_t seed, size_t B = 1, uint64_t first_index = 0) {                           // :264-267
    MatZ e(B * m());
    check(psfp_samp_d(h_, seed, first_index, B, e.data()), "samp_d");
    return e;
  }
  MatZ samp_p(const MatZq& u, uint64_t seed, uint64_t first_index = 0) {                         // :304-336
    const size_t B = u.size() / n();
    MatZ e(B * m());
    check(psfp_samp_p(h_, seed, first_index, B, u.data(), e.data()), "samp_p");
    return e;
  }
  // the same, asynchronous: e (B * m entries, caller-owned) is complete after wait(); at most two calls in flight per handle
  void samp_p_async(const MatZq& u, MatZ& e, uint64_t seed, uint64_t first_index = 0) {
    const size_t B = u.size() / n();
    e.resize(B * m());
    check(psfp_samp_p_async(h_, seed, first_index, B, u.data(), e.data()), "samp_p_async");
  }
  void wait() { check(psfp_wait(h_), "wait"); }
  MatZq f_a(const MatZ& sigma) {                                                                  // :366-369
    if (sigma.empty() || sigma.size() % m() != 0) throw PsfError(PSF_ERR_DOMAIN, "f_a");
    const size_t B = sigma.size() / m();
    MatZq u(B * n());
    check(psfp_f_a(h_, B, sigma.data(), u.data()), "f_a");
    return u;
  }
  bool check_domain(const MatZ& sigma) {                                                          // :396-402 (one vector)
    uint8_t ok = 0;
    check(psfp_check_domain(h_, 1, sigma.data(), sigma.size(), &ok), "check_domain");
    return ok != 0;
  }
  psfp_handle* raw() { return h_; }

 private:
  psf_gadget_params gp_;
  psfp_handle* h_ = nullptr;
};

// ---- PSFGPV (gpv.rs:53-57, :59-225) ------------------------------------------------------------------------------------------
class PSFGPV {
 public:
  struct Trapdoor { std::vector<int32_t> basis_t; std::vector<double> gso_t; };   // (short_base, short_base_gso) transposed, gpv.rs:61

  PSFGPV(const psf_gadget_params& gp, double s, int device = 0) : gp_(gp) {
    psfgpv_params p{gp, s, device, 0};
    check(psfgpv_create(&p, &h_), "PSFGPV");
  }
  ~PSFGPV() { psfgpv_destroy(h_); }
  PSFGPV(const PSFGPV&) = delete;
  PSFGPV& operator=(const PSFGPV&) = delete;
  size_t n() const { return gp_.n; }
  size_t m() const { return psfgpv_m(h_); }

  std::pair<MatZq, Trapdoor> trap_gen(uint64_t seed) {                                            // :83-94
    check(psfgpv_trap_gen(h_, seed), "trap_gen");
    const size_t mm = m();
    MatZq A(gp_.n * mm);
    Trapdoor td;
    td.basis_t.resize(mm * mm); td.gso_t.resize(mm * mm);
    check(psfgpv_export_key(h_, A.data(), nullptr, td.basis_t.data(), td.gso_t.data()), "export_key");
    return {std::move(A), std::move(td)};
  }
  void load_key(const MatZq& A, const Trapdoor& td) { check(psfgpv_load_key(h_, A.data(), td.basis_t.data(), td.gso_t.data()), "load_key"); }
  MatZ samp_d(uint64_t seed, size_t B = 1, uint64_t first_index = 0) {                            // :113-116
    MatZ e(B * m());
    check(psfgpv_samp_d(h_, seed, first_index, B, e.data()), "samp_d");
    return e;
  }
  MatZ samp_p(const MatZq& u, uint64_t seed, uint64_t first_index = 0) {                          // :152-161
    const size_t B = u.size() / n();
    MatZ e(B * m());
    check(psfgpv_samp_p(h_, seed, first_index, B, u.data(), e.data()), "samp_p");
    return e;
  }
  // the same, asynchronous: e (B * m entries, caller-owned) is complete after wait(); at most two calls in flight per handle
  void samp_p_async(const MatZq& u, MatZ& e, uint64_t seed, uint64_t first_index = 0) {
    const size_t B = u.size() / n();
    e.resize(B * m());
    check(psfgpv_samp_p_async(h_, seed, first_index, B, u.data(), e.data()), "samp_p_async");
  }
  void wait() { check(psfgpv_wait(h_), "wait"); }
  MatZq f_a(const MatZ& sigma) {                                                                   // :190-193
    if (sigma.empty() || sigma.size() % m() != 0) throw PsfError(PSF_ERR_DOMAIN, "f_a");
    const size_t B = sigma.size() / m();
    MatZq u(B * n());
    check(psfgpv_f_a(h_, B, sigma.data(), u.data()), "f_a");
    return u;
  }
  bool check_domain(const MatZ& sigma) {                                                           // :219-224
    uint8_t ok = 0;
    check(psfgpv_check_domain(h_, 1, sigma.data(), sigma.size(), &ok), "check_domain");
    return ok != 0;
  }

 private:
  psf_gadget_params gp_;
  psfgpv_handle* h_ = nullptr;
};

// ---- PSFGPVRing (gpv_ring.rs:62-67, :69-284); polynomials are n coefficients, constant term first ----------------------------
class PSFGPVRing {
 public:
  struct Trapdoor { MatZ r, e; };   // two 1 x k MatPolyOverZ, gpv_ring.rs:72

  PSFGPVRing(const psf_gadget_params& ring_gp, double s, double s_td, int device = 0) : gp_(ring_gp) {
    psfring_params p{ring_gp, s, s_td, device, 0};
    check(psfring_create(&p, &h_), "PSFGPVRing");
  }
  ~PSFGPVRing() { psfring_destroy(h_); }
  PSFGPVRing(const PSFGPVRing&) = delete;
  PSFGPVRing& operator=(const PSFGPVRing&) = delete;
  size_t n() const { return gp_.n; }
  size_t polys() const { return gp_.k + 2; }

  std::pair<MatZq, Trapdoor> trap_gen(uint64_t seed) {                                             // :91-98
    check(psfring_trap_gen(h_, seed), "trap_gen");
    MatZq a(polys() * n());
    Trapdoor td;
    td.r.resize(gp_.k * n()); td.e.resize(gp_.k * n());
    check(psfring_export_key(h_, a.data(), td.r.data(), td.e.data(), nullptr, nullptr), "export_key");
    return {std::move(a), std::move(td)};
  }
  void load_key(const MatZq& a, const Trapdoor& td) { check(psfring_load_key(h_, a.data(), td.r.data(), td.e.data()), "load_key"); }
  MatZ samp_d(uint64_t seed, size_t B = 1, uint64_t first_index = 0) {                             // :118-122
    MatZ sg(B * polys() * n());
    check(psfring_samp_d(h_, seed, first_index, B, sg.data()), "samp_d");
    return sg;
  }
  MatZ samp_p(const MatZq& u, uint64_t seed, uint64_t first_index = 0) {                           // :160-212
    const size_t B = u.size() / n();
    MatZ sg(B * polys() * n());
    check(psfring_samp_p(h_, seed, first_index, B, u.data(), sg.data()), "samp_p");
    return sg;
  }
  void samp_p_async(const MatZq& u, MatZ& sg, uint64_t seed, uint64_t first_index = 0) {           // sg is complete after wait()
    const size_t B = u.size() / n();
    sg.resize(B * polys() * n());
    check(psfring_samp_p_async(h_, seed, first_index, B, u.data(), sg.data()), "samp_p_async");
  }
  void wait() { check(psfring_wait(h_), "wait"); }
  MatZq f_a(const MatZ& sigma) {                                                                    // :243-247
    const size_t d = polys() * n();
    if (sigma.empty() || sigma.size() % d != 0) throw PsfError(PSF_ERR_DOMAIN, "f_a");
    MatZq u(sigma.size() / d * n());
    check(psfring_f_a(h_, sigma.size() / d, sigma.data(), u.data()), "f_a");
    return u;
  }
  bool check_domain(const MatZ& sigma) {                                                            // :274-283
    uint8_t ok = 0;
    check(psfring_check_domain(h_, 1, sigma.data(), sigma.size(), &ok), "check_domain");
    return ok != 0;
  }

 private:
  psf_gadget_params gp_;
  psfring_handle* h_ = nullptr;
};

}  // namespace psf_mi355x
