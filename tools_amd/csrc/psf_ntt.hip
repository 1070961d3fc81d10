// psf_ntt.hip -- negacyclic NTT products over R_q = Z_q[X]/(X^n + 1): plan cache, shape dispatch and launches (psf_ntt_api.hpp).
// PolynomialRingZq multiplication under gadget_ring.rs:78 and gpv_ring.rs:243-247; the kernels are in psf_ntt_kernels.hpp / psf_ntt_core.hpp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <mutex>
#include <tuple>
#include <type_traits>
#include "psf_host.hpp"
#include "psf_ntt_api.hpp"
#include "psf_ntt_kernels.hpp"

using namespace psf;
using namespace psf::ntt;

#define NTT_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      std::fprintf(stderr, "[psf_mi355x] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return PSF_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

namespace {

// ---- generic form: any power-of-two n <= 8192, any plan (leaf degree d = n >> L of any size), 32-bit Montgomery arithmetic, data in LDS ------------
// One product per workgroup at a time; the n/2 butterflies of a and of b of one level run side by side (n threads busy), leaves by schoolbook.
__global__ __launch_bounds__(256) void k_ntt_polymul_lds(NttDev p, uint32_t n, uint32_t L, uint32_t d, const uint64_t* __restrict__ A, const int64_t* __restrict__ Bp,
                                                         uint64_t* __restrict__ out, size_t count) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_smem[];   // zetas fwd [2^L] | inv [2^L] | a[n] | b[n] | c[n]
  const Mod32<DevWave> md = make_policy<0>(p, 0);
  uint32_t* zf = lds_smem;
  uint32_t* zi = zf + (1u << L);
  uint32_t* sa = zi + (1u << L);
  uint32_t* sb = sa + n;
  uint32_t* sc = sb + n;
  for (uint32_t i = threadIdx.x; i < (2u << L); i += 256) zf[i] = p.zetas[i];
  for (size_t pr = blockIdx.x; pr < count; pr += gridDim.x) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
      sa[i] = (uint32_t)(A[pr * n + i] % p.q);
      const int64_t v = Bp[pr * n + i] % (int64_t)p.q;
      sb[i] = (uint32_t)(v < 0 ? v + (int64_t)p.q : v);
    }
    __syncthreads();
    uint32_t len = n >> 1;
    for (uint32_t l = 0; l < L; ++l, len >>= 1) {                        // Cooley-Tukey, block b of level l uses zetas[2^l + b]
      for (uint32_t e = threadIdx.x; e < n; e += 256) {
        uint32_t* s = e < n / 2 ? sa : sb;
        const uint32_t bf = e < n / 2 ? e : e - n / 2, blk = bf / len, j = bf - blk * len, lo = blk * 2 * len + j, hi = lo + len;
        const uint32_t t = md.mul(zf[(1u << l) + blk], s[hi]), u = s[lo];
        s[hi] = md.sub(u, t);
        s[lo] = md.add(u, t);
      }
      __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < n; e += 256) {                    // leaf e / d: Z_q[X]/(X^d - gamma), gamma = +-zeta of the last level
      const uint32_t leaf = e / d, c = e - leaf * d;
      const uint32_t z = zf[(1u << (L - 1)) + (leaf >> 1)], gamma = (leaf & 1) ? md.neg(z) : z;
      const uint32_t* pa = sa + leaf * d;
      const uint32_t* pb = sb + leaf * d;
      uint32_t lo = 0, hi = 0;
      for (uint32_t i = 0; i < d; ++i) {
        if (i <= c) lo = md.add(lo, md.mul(pa[i], pb[c - i]));
        else hi = md.add(hi, md.mul(pa[i], pb[d + c - i]));
      }
      sc[e] = md.add(lo, md.mul(gamma, hi));                             // every term times R^-1 (gamma is in Montgomery form)
    }
    __syncthreads();
    len = d;
    for (int l = (int)L - 1; l >= 0; --l, len <<= 1) {                   // Gentleman-Sande
      for (uint32_t e = threadIdx.x; e < n / 2; e += 256) {
        const uint32_t blk = e / len, j = e - blk * len, lo = blk * 2 * len + j, hi = lo + len;
        const uint32_t u = sc[lo], v = sc[hi];
        sc[lo] = md.add(u, v);
        sc[hi] = md.mul(zi[(1u << l) + blk], md.sub(u, v));
      }
      __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n; i += 256) out[pr * n + i] = md.mul(p.fin, sc[i]);
  }
}

struct Plan {
  NttPlan pl;
  NttTables tb;
  uint32_t* d_zetas = nullptr;
  int route = 0;
};
std::mutex g_mu;
std::map<std::tuple<int, uint64_t, size_t>, Plan*> g_plans;

// the plan of (device, q, n), built once; nullptr: no NTT for this (q, n)
Plan* plan_for(int device, uint64_t q, size_t n, psf_status* st) {
  *st = PSF_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  const auto key = std::make_tuple(device, q, n);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) return it->second->route ? it->second : nullptr;
  Plan* P = new Plan();
  if (q < (1ull << 31) && n >= 2 && n <= 8192) P->pl = make_ntt_plan(q, (uint32_t)n);
  if (P->pl.ok) {
    P->tb = make_ntt_tables(P->pl);
    P->route = P->tb.wave ? 2 : 1;
    if (device >= 0) {
      if (hipSetDevice(device) != hipSuccess || hipMalloc(&P->d_zetas, P->tb.zetas.size() * sizeof(uint32_t)) != hipSuccess ||
          hipMemcpy(P->d_zetas, P->tb.zetas.data(), P->tb.zetas.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
        delete P;
        *st = PSF_ERR_HIP;
        return nullptr;
      }
    }
  }
  g_plans[key] = P;
  return P->route ? P : nullptr;
}

template <int V> using ic = std::integral_constant<int, V>;
// the shapes that have a wave kernel (make_ntt_tables decides logn, ld, qb)
template <class F> bool for_shape(int logn, int ld, int qb, F&& f) {
#define PSF_SHAPE(LN, LDV, QBV) if (logn == LN && ld == LDV && qb == QBV) { f(ic<LN>{}, ic<LDV>{}, ic<QBV>{}); return true; }
  PSF_SHAPE(7, 0, 12) PSF_SHAPE(8, 1, 12) PSF_SHAPE(9, 2, 12)
  PSF_SHAPE(7, 0, 14) PSF_SHAPE(7, 1, 14) PSF_SHAPE(8, 0, 14) PSF_SHAPE(8, 1, 14) PSF_SHAPE(8, 2, 14) PSF_SHAPE(9, 0, 14) PSF_SHAPE(9, 1, 14) PSF_SHAPE(9, 2, 14)
  PSF_SHAPE(10, 0, 14) PSF_SHAPE(10, 1, 14) PSF_SHAPE(10, 2, 14)
  PSF_SHAPE(7, 0, 0) PSF_SHAPE(7, 1, 0) PSF_SHAPE(8, 0, 0) PSF_SHAPE(8, 1, 0) PSF_SHAPE(8, 2, 0) PSF_SHAPE(9, 0, 0) PSF_SHAPE(9, 1, 0) PSF_SHAPE(9, 2, 0)
  PSF_SHAPE(10, 0, 0) PSF_SHAPE(10, 1, 0) PSF_SHAPE(10, 2, 0)
#undef PSF_SHAPE
  return false;
}

NttDev dev_args(const Plan* P, int e, int e_fa) {
  NttDev a;
  a.q = P->tb.q; a.qinv16 = P->tb.qinv16; a.nqinv32 = P->tb.nqinv32; a.r2 = P->tb.r2;
  a.fin = ntt_final_scale(P->tb, P->pl, e);
  a.fin_fa = ntt_final_scale(P->tb, P->pl, e_fa);
  a.zetas = P->d_zetas;
  return a;
}
unsigned wave_grid(size_t count) {                                       // four products per workgroup at a time, at most 8 workgroups per CU
  static const size_t cap = [] { const char* e = psf_exp_env("PSF_NTT_GRID"); const long v = e ? std::atol(e) : 0; return (size_t)(v > 0 ? v : 2048); }();
  const size_t g = (count + 3) / 4;
  return (unsigned)(g < 1 ? 1 : g > cap ? cap : g);
}

}  // namespace

namespace psf {

int ntt_route(uint64_t q, size_t n) {
  psf_status st;
  Plan* P = plan_for(-1, q, n, &st);
  return P ? P->route : 0;
}

psf_status ntt_polymul_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, const void* d_b, void* d_out, int io_bits, hipStream_t st) {
  if (io_bits != 16 && io_bits != 64) return PSF_ERR_PARAM;
  if (count && (!d_a || !d_b || !d_out)) return PSF_ERR_PARAM;
  psf_status rc;
  Plan* P = plan_for(device, q, n, &rc);
  if (!P) return rc != PSF_OK ? rc : PSF_ERR_UNSUPPORTED;
  if (count == 0) return PSF_OK;
  NTT_TRY(hipSetDevice(device));
  if (P->route == 1) {
    if (io_bits != 64) return PSF_ERR_UNSUPPORTED;
    const NttDev a = dev_args(P, 1, 1);
    const size_t smem = ((2u << P->pl.L) + 3 * n) * sizeof(uint32_t);
    if (smem > 160 * 1024) return PSF_ERR_UNSUPPORTED;                   // gfx950: 160 KiB of LDS per workgroup (n = 8192 with a fully splitting prime needs 160 KiB exactly)
    if (smem > 64 * 1024) {                                              // above the default limit the kernel's attribute is raised once per process and device
      static std::mutex mu; static std::set<int> raised;
      std::lock_guard<std::mutex> lk(mu);
      if (!raised.count(device)) {
        NTT_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_polymul_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.insert(device);
      }
    }
    hipLaunchKernelGGL(k_ntt_polymul_lds, dim3((unsigned)(count > 4096 ? 4096 : count)), dim3(256), smem, st, a, (uint32_t)n, P->pl.L, P->pl.d,
                       (const uint64_t*)d_a, (const int64_t*)d_b, (uint64_t*)d_out, count);
    NTT_TRY(hipGetLastError());
    return PSF_OK;
  }
  if (io_bits == 16 && P->tb.qb == 0) return PSF_ERR_UNSUPPORTED;
  const bool ok = for_shape(P->tb.logn, P->tb.ld, P->tb.qb, [&](auto ln, auto ldv, auto qbv) {
    constexpr int LN = decltype(ln)::value, LDV = decltype(ldv)::value, QBV = decltype(qbv)::value;
    const NttDev a = dev_args(P, Kern<LN, LDV, QBV>::E, Kern<LN, LDV, QBV>::E + 1);
    if constexpr (QBV != 0) {
      if (io_bits == 16) { hipLaunchKernelGGL((k_ntt_polymul<LN, LDV, QBV, 16>), dim3(wave_grid(count)), dim3(256), 0, st, a, d_a, d_b, d_out, count); return; }
    }
    hipLaunchKernelGGL((k_ntt_polymul<LN, LDV, QBV, 64>), dim3(wave_grid(count)), dim3(256), 0, st, a, d_a, d_b, d_out, count);
  });
  if (!ok) return PSF_ERR_UNSUPPORTED;
  NTT_TRY(hipGetLastError());
  return PSF_OK;
}

psf_status ntt_forward_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, int io_bits, uint32_t* d_hat, hipStream_t st) {
  if (io_bits != 16 && io_bits != 64) return PSF_ERR_PARAM;
  if (count && (!d_a || !d_hat)) return PSF_ERR_PARAM;
  psf_status rc;
  Plan* P = plan_for(device, q, n, &rc);
  if (!P) return rc != PSF_OK ? rc : PSF_ERR_UNSUPPORTED;
  if (P->route != 2 || (io_bits == 16 && P->tb.qb == 0)) return PSF_ERR_UNSUPPORTED;
  if (count == 0) return PSF_OK;
  NTT_TRY(hipSetDevice(device));
  const bool ok = for_shape(P->tb.logn, P->tb.ld, P->tb.qb, [&](auto ln, auto ldv, auto qbv) {
    constexpr int LN = decltype(ln)::value, LDV = decltype(ldv)::value, QBV = decltype(qbv)::value;
    const NttDev a = dev_args(P, 1, 1);
    if constexpr (QBV != 0) {
      if (io_bits == 16) { hipLaunchKernelGGL((k_ntt_forward<LN, LDV, QBV, 16, false>), dim3(wave_grid(count)), dim3(256), 0, st, a, d_a, d_hat, count); return; }
    }
    hipLaunchKernelGGL((k_ntt_forward<LN, LDV, QBV, 64, false>), dim3(wave_grid(count)), dim3(256), 0, st, a, d_a, d_hat, count);
  });
  if (!ok) return PSF_ERR_UNSUPPORTED;                      // route 2 without an instantiated shape: nothing was launched
  NTT_TRY(hipGetLastError());
  return PSF_OK;
}

psf_status ntt_mul_hat_dev(int device, uint64_t q, size_t n, size_t count, const uint32_t* d_hat, size_t hat_stride, const void* d_b, void* d_out, int io_bits, hipStream_t st) {
  if (io_bits != 16 && io_bits != 64) return PSF_ERR_PARAM;
  if (count && (!d_hat || !d_b || !d_out)) return PSF_ERR_PARAM;
  psf_status rc;
  Plan* P = plan_for(device, q, n, &rc);
  if (!P) return rc != PSF_OK ? rc : PSF_ERR_UNSUPPORTED;
  if (P->route != 2 || (io_bits == 16 && P->tb.qb == 0)) return PSF_ERR_UNSUPPORTED;
  if (count == 0) return PSF_OK;
  NTT_TRY(hipSetDevice(device));
  const bool ok = for_shape(P->tb.logn, P->tb.ld, P->tb.qb, [&](auto ln, auto ldv, auto qbv) {
    constexpr int LN = decltype(ln)::value, LDV = decltype(ldv)::value, QBV = decltype(qbv)::value;
    const NttDev a = dev_args(P, Kern<LN, LDV, QBV>::E, Kern<LN, LDV, QBV>::E + 1);
    if constexpr (QBV != 0) {
      if (io_bits == 16) { hipLaunchKernelGGL((k_ntt_mul_hat<LN, LDV, QBV, 16>), dim3(wave_grid(count)), dim3(256), 0, st, a, d_hat, hat_stride, d_b, d_out, count); return; }
    }
    hipLaunchKernelGGL((k_ntt_mul_hat<LN, LDV, QBV, 64>), dim3(wave_grid(count)), dim3(256), 0, st, a, d_hat, hat_stride, d_b, d_out, count);
  });
  if (!ok) return PSF_ERR_UNSUPPORTED;                      // route 2 without an instantiated shape: nothing was launched
  NTT_TRY(hipGetLastError());
  return PSF_OK;
}

psf_status ntt_ring_fa_dev(int device, uint64_t q, size_t n, uint32_t K, const uint32_t* d_hat, const int64_t* d_sigma, uint64_t* d_u, size_t B, hipStream_t st) {
  if (B && (!d_hat || !d_sigma || !d_u)) return PSF_ERR_PARAM;
  psf_status rc;
  Plan* P = plan_for(device, q, n, &rc);
  if (!P) return rc != PSF_OK ? rc : PSF_ERR_UNSUPPORTED;
  const size_t smem = (((P->tb.qb == 12 ? 4u : 2u) << P->pl.L) + (size_t)K * n) * sizeof(uint32_t);
  if (P->route != 2 || smem > 64 * 1024) return PSF_ERR_UNSUPPORTED;
  if (B == 0) return PSF_OK;
  NTT_TRY(hipSetDevice(device));
  const bool ok = for_shape(P->tb.logn, P->tb.ld, P->tb.qb, [&](auto ln, auto ldv, auto qbv) {
    constexpr int LN = decltype(ln)::value, LDV = decltype(ldv)::value, QBV = decltype(qbv)::value;
    const NttDev a = dev_args(P, Kern<LN, LDV, QBV>::E, Kern<LN, LDV, QBV>::E + 1);
    hipLaunchKernelGGL((k_ring_fa<LN, LDV, QBV>), dim3(wave_grid(B)), dim3(256), smem, st, a, d_hat, K, d_sigma, d_u, B);
  });
  if (!ok) return PSF_ERR_UNSUPPORTED;                      // route 2 without an instantiated shape: nothing was launched
  NTT_TRY(hipGetLastError());
  return PSF_OK;
}

}  // namespace psf
