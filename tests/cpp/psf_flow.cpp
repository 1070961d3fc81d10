// The README / doctest flow of the reference (README.md:62-77, mp_perturbation.rs:43-56, gpv.rs:40-51, gpv_ring.rs:44-60)
// written against the C++ mirror of the PSF trait.  Exit code 0 = all invariants hold, 3 = no usable GPU.
#include <cstdio>
#include "../../include/psf_mi355x.hpp"

using namespace psf_mi355x;

template <class P>
static bool flow(P& psf, const char* name) {
  auto key = psf.trap_gen(1);
  auto domain_sample = psf.samp_d(2);
  if (!psf.check_domain(domain_sample)) { std::printf("%s: samp_d outside D_n\n", name); return false; }
  auto range_fa = psf.f_a(domain_sample);
  auto preimage = psf.samp_p(range_fa, 3);
  if (!psf.check_domain(preimage)) { std::printf("%s: preimage outside D_n\n", name); return false; }
  if (psf.f_a(preimage) != range_fa) { std::printf("%s: f_a(preimage) != u\n", name); return false; }
  std::printf("%s ok\n", name);
  return true;
}

int main() {
  try {
    PSFPerturbation p(gadget_parameters_default(8, 64), 3.0, 25.0);
    PSFGPV g(gadget_parameters_default(8, 64), 12.0);
    PSFGPVRing r(gadget_parameters_ring_default(8, 512), 100.0, 1.005);
    bool ok = flow(p, "PSFPerturbation") && flow(g, "PSFGPV") && flow(r, "PSFGPVRing");
    // f_a on a vector outside the domain must fail like the reference's assert! (mp_perturbation.rs:367)
    MatZ big(p.m(), 0);
    big[0] = 25 * (int64_t)p.m();
    bool threw = false;
    try { p.f_a(big); } catch (const PsfError& e) { threw = e.status == PSF_ERR_DOMAIN; }
    if (!threw) { std::printf("f_a accepted a vector outside D_n\n"); ok = false; }
    {  // psfp_samp_p_async / psfp_wait: two calls in flight return the rows of two synchronous calls
      auto key = p.trap_gen(7);
      auto u1 = p.f_a(p.samp_d(8, 3)), u2 = p.f_a(p.samp_d(9, 3));
      const MatZ s1 = p.samp_p(u1, 10), s2 = p.samp_p(u2, 11);
      MatZ a1, a2;
      p.samp_p_async(u1, a1, 10);
      p.samp_p_async(u2, a2, 11);
      p.wait();
      if (a1 != s1 || a2 != s2) { std::printf("asynchronous calls differ from synchronous ones\n"); ok = false; }
      // a trapdoor installed without a factor cannot sample until compute_sqrt_sigma_2 has run
      PSFPerturbation p2(gadget_parameters_default(8, 64), 3.0, 25.0);
      p2.load_trapdoor(key.second.R, &key.first);
      bool nokey = false;
      try { p2.samp_p(u1, 10); } catch (const PsfError& e) { nokey = e.status == PSF_ERR_NO_KEY; }
      p2.compute_sqrt_sigma_2(25.0);
      if (!nokey || p2.samp_p(u1, 10) != s1) { std::printf("load_trapdoor / compute_sqrt_sigma_2 flow failed\n"); ok = false; }
    }
    return ok ? 0 : 1;
  } catch (const PsfError& e) {
    std::printf("PsfError: %s\n", e.what());
    return e.status == PSF_ERR_HIP ? 3 : 2;
  }
}
