// The nearest-plane configurations (C2: PSFGPV n = 256 q = 3329 B = 1024; C4: PSFGPVRing degree 256 q = 3329 B = 4096) as a plain C++ process over
// the C ABI: the profiled program for `rocprofv3 --pmc FETCH_SIZE -- tools/bin/np_harness c2` (counter collection on `python3 bench.py` crashed at
// C2 and hung at C4 in round 2).  One key, `reps` samp_p calls on device-resident targets, invariants checked on the last.
// g++ -O2 -std=c++17 -I include tools/np_harness.cpp -L tools_amd/lib -lpsf_mi355x -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../../tools_amd/lib' -o tools/bin/np_harness
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "psf_mi355x.h"

#define CK(x) do { psf_status s_ = (x); if (s_ != PSF_OK) { std::printf("%s -> %s\n", #x, psf_status_string(s_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const char* cfg = argc > 1 ? argv[1] : "c2";
  const int reps = argc > 2 ? std::atoi(argv[2]) : 3;
  const bool ring = !std::strcmp(cfg, "c4");
  const uint64_t n = 256, q = 3329;
  const size_t B = ring ? 4096 : 1024;
  size_t d = 0;
  psfgpv_handle* g = nullptr;
  psfring_handle* r = nullptr;
  if (!ring) {
    psf_gadget_params gp;
    CK(psf_gadget_params_default(n, q, &gp));
    psfgpv_params p{gp, 1024.0, 0, 0};
    CK(psfgpv_create(&p, &g));
    CK(psfgpv_trap_gen(g, 3));
    d = psfgpv_m(g);
  } else {
    psf_gadget_params gp;
    CK(psf_gadget_params_ring_default(n, q, &gp));
    const double s = ((2 * 2 * 1.005 * std::sqrt((double)n) + 1) * 2) * 4;          // compute_s, gpv_ring.rs:296-298
    psfring_params p{gp, s, 1.005, 0, 0};
    CK(psfring_create(&p, &r));
    CK(psfring_trap_gen(r, 3));
    d = n * (gp.k + 2);
  }
  uint64_t *du = nullptr, *du2 = nullptr; int64_t* de = nullptr; uint8_t* dok = nullptr;
  if (hipMalloc(&du, B * n * 8) != hipSuccess || hipMalloc(&du2, B * n * 8) != hipSuccess || hipMalloc(&de, B * d * 8) != hipSuccess || hipMalloc(&dok, B) != hipSuccess) return 3;
  if (!ring) CK(psfgpv_uniform_targets_dev(g, 7, 0, B, du, nullptr)); else CK(psfring_uniform_targets_dev(r, 7, 0, B, du, nullptr));
  double best = 1e30;
  for (int i = 0; i < reps; ++i) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    if (!ring) CK(psfgpv_samp_p_dev(g, 1000 + i, 0, B, du, de, nullptr)); else CK(psfring_samp_p_dev(r, 1000 + i, 0, B, du, de, nullptr));
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    best = ms < best ? ms : best;
  }
  if (!ring) { CK(psfgpv_last_status(g)); CK(psfgpv_f_a_dev(g, B, de, du2, dok, nullptr)); } else { CK(psfring_last_status(r)); CK(psfring_f_a_dev(r, B, de, du2, dok, nullptr)); }
  hipDeviceSynchronize();
  std::vector<uint64_t> u(B * n), u2(B * n); std::vector<uint8_t> ok(B);
  hipMemcpy(u.data(), du, B * n * 8, hipMemcpyDeviceToHost); hipMemcpy(u2.data(), du2, B * n * 8, hipMemcpyDeviceToHost); hipMemcpy(ok.data(), dok, B, hipMemcpyDeviceToHost);
  bool valid = u == u2;
  for (uint8_t o : ok) valid = valid && o;
  std::printf("{\"config\": \"%s\", \"d\": %zu, \"batch\": %zu, \"reps\": %d, \"best_ms_per_call\": %.3f, \"valid\": %s}\n", cfg, d, B, reps, best, valid ? "true" : "false");
  if (g) psfgpv_destroy(g);
  if (r) psfring_destroy(r);
  return valid ? 0 : 1;
}
