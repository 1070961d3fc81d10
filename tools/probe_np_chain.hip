// Measurement harness (not product code): the DEPENDENT CHAIN of one nearest-plane draw (psf_np_kernels.hpp, np_sample_body, G = 1) replayed in isolation, one
// wave per SIMD, segment by segment, to see where the ~1.5 k shader ticks of a serial step go ("a per-instruction account of the step").
// Every variant runs the same loop of N dependent iterations -- the running projection of the next row depends on the z just drawn through the fma update, as in
// the walk -- and adds one more piece of the step:
//   0  v_readlane x2 (t of the row) -> v_mul_f64 (centre) -> v_fma_f64 (update with a constant z): the f64 skeleton
//   1  + v_ceil_f64, v_add_f64, v_cvt_f32_f64, v_cvt_i32_f64 (c_rel, lo)
//   2  + the fp32 screen: v_fma_f32, v_mul_f32, v_exp_f32, 2 compares
//   3  + three ballots, s_ff1, the shift / test of the "certain" mask, v_readlane of the candidate index, z = lo + idx, v_cvt_f64_i32
//   4  + the loads of the helper's record for the NEXT step from LDS (ds_read_b128) and of g (ds_read_b64) -- issued ahead, as the kernel does
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe_np_chain.hip -o tools/bin/probe_np_chain ;  tools/bin/probe_np_chain [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ inline double bcast_d(double x, int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane(__double2loint(x), src), hi = (unsigned)__builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double((int)hi, (int)lo);
}

template <int V>
__global__ __launch_bounds__(256) void k_chain(int iters, const float4* __restrict__ recs, double* __restrict__ out, unsigned long long* __restrict__ ticks) {
  __shared__ float4 s_rec[4][64 * 8];
  __shared__ double s_g[4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < 64 * 8; i += 64) { s_rec[wave][i] = recs[(blockIdx.x * 4 + wave) * 512 + i]; s_g[wave][i] = 1e-3 * (double)((i * 37 + lane) % 101 - 50); }
  __syncthreads();
  double t = 100.0 + 0.37 * lane;                      // running projections, one row per lane
  const double inv_n2 = 0.731;
  const float inv_sk = 0.0521f;
  const int c6 = 120;
  float4 rec = s_rec[wave][lane];
  double gl = s_g[wave][lane];
  long long zsum = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int ls = it & 63;                             // the row whose draw this is (its t sits in lane ls)
    float4 recn = rec; double gln = gl;
    if (V >= 4) { recn = s_rec[wave][((it + 1) & 7) * 64 + lane]; gln = s_g[wave][((it + 1) & 7) * 64 + lane]; }      // next step's operands, in flight during this one
    const double tl = bcast_d(t, ls);
    const double cen = tl * inv_n2;
    long long z = 3;
    if (V >= 1) {
      const double cc = ceil(cen);
      const float c_rel = (float)(cc - cen) - (float)c6;
      const int lo = (int)cc - c6;
      z = lo;
      if (V >= 2) {
        const float ak = fmaf(c_rel, inv_sk, rec.x);
        const float rho = __builtin_amdgcn_exp2f(-(ak * ak));
        const bool cand = rec.y <= fmaf(rho, 1.001f, 1e-9f);
        const bool sure = rec.y + 0x1.0p-16f <= rho * 0.999f;
        if (V >= 3) {
          const bool bad = cc == cen || !(fabs(cen) < 0x1.0p30);
          const unsigned long long mc = __ballot(cand), m1 = __ballot(sure), bw = __ballot(bad);
          if (V == 5) {
            // selection without s_ff1 / v_readlane(SGPR lane): lowest candidate = mc & -mc; its index read by v_readfirstlane under exec = candidates
            const unsigned long long first = mc & (0ull - mc);
            int idx = 0;
            if (cand) idx = __builtin_amdgcn_readfirstlane((int)__float_as_uint(rec.z) & 0xfff);
            idx = __builtin_amdgcn_readfirstlane(idx);          // (uniform again for the lanes that were masked off)
            z = (bw == 0 && (first & m1)) ? (long long)(lo + idx) : (long long)(lo + 1);
          } else {
          const int fl = mc ? __builtin_ctzll(mc) : 0;
          const int idx = __builtin_amdgcn_readlane((int)__float_as_uint(rec.z) & 0xfff, fl);
          z = (bw == 0 && ((m1 >> fl) & 1)) ? (long long)(lo + idx) : (long long)(lo + 1);
          }
        } else {
          z = lo + ((cand ? 1 : 0) + (sure ? 2 : 0));   // keeps the screen alive without the scalar part
        }
      }
    }
    zsum += z;
    const double nz = -(double)(int)(z & 0xff);          // (bounded, so that t stays in range over many iterations)
    t = fma(nz, gl, t) + 0.25 * (double)(int)(z & 0xff) * gl;      // the update of the rows below, then pulled back so that the centre stays O(100)
    rec = recn; gl = gln;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = t + (double)zsum;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 8192;
  const int nwg = 256;
  float4* recs; double* out; unsigned long long* ticks;
  CK(hipMalloc(&recs, (size_t)nwg * 4 * 512 * sizeof(float4))); CK(hipMalloc(&out, (size_t)nwg * 256 * 8)); CK(hipMalloc(&ticks, nwg * 8));
  std::vector<float4> h((size_t)nwg * 4 * 512);
  unsigned s = 12345;
  for (auto& r : h) { s = s * 1664525u + 1013904223u; r.x = (float)((s >> 8) % 480) * 0.0521f; s = s * 1664525u + 1013904223u; r.y = (float)(s >> 16) * 0x1.0p-16f; s = s * 1664525u + 1013904223u; r.z = __builtin_bit_cast(float, (s >> 8) & 0xfffu); r.w = 0.f; }
  CK(hipMemcpy(recs, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
  std::vector<unsigned long long> ht(nwg);
  auto run = [&](const char* name, auto kern) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, iters, recs, out, ticks); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(ht.data(), ticks, nwg * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (auto v : ht) sum += (double)v;
    std::printf("%-78s %8.1f ticks per dependent step\n", name, sum / nwg / iters);
  };
  std::printf("one wave per SIMD, %d dependent steps per wave (s_memtime / clock ticks of the shader clock domain as __builtin_readcyclecounter reports them)\n", iters);
  run("0: readlane x2 -> v_mul_f64 -> v_cvt / v_fma_f64 update (skeleton)", k_chain<0>);
  run("1: + ceil, c_rel (f64 add, cvt f32), lo (cvt i32)", k_chain<1>);
  run("2: + fp32 screen (fma, mul, v_exp_f32, two compares)", k_chain<2>);
  run("3: + three ballots, s_ff1, mask test, v_readlane(idx), z = lo + idx", k_chain<3>);
  run("4: + next step's record / g from LDS (ds_read_b128, ds_read_b64) in flight", k_chain<4>);
  run("5: as 4, selection by v_readfirstlane under exec = candidates (no s_ff1, no v_readlane with an SGPR lane)", k_chain<5>);
  return 0;
}
