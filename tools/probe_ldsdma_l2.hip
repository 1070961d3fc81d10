// Probe: at what rate does LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no staging registers) fill the LDS of EVERY CU at once
// when the bytes come from (a) HBM, (b) the XCD's L2, (c) a panel the size of the recombination's R (237 MB: Infinity Cache) that all CUs sweep together, (d) the
// access shape of k_recombine_mfma_big (a 256-row strip of the panel per workgroup, the workgroups of a super-tile sharing strips)?
// The int8 stages of the headline (k_zq_mfma, k_recombine_mfma_big) stage 7.6 GB per launch out of a 237 MB R and the z planes -- a 32x re-read that never goes to
// HBM.  Round 4 closed the "128 x 128 wave tile" item on the assumption that LDS-DMA delivers ~7 TB/s chip-wide whatever the source (the guide's ldsdma-fill row is an
// HBM stream); this measures it.  No consumer: the waves issue the fills of a 64 KiB stage, wait (vmcnt(0)), meet at a barrier and go on, as the kernels' K loop does.
// hipcc --offload-arch=gfx950 -O3 tools/probe_ldsdma_l2.hip -o tools/bin/probe_ldsdma_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// every workgroup: `stages` stages of 64 KiB; stage s of workgroup w starts at byte offset origin[w] + (base[w] + s * 64 KiB) mod span inside buf
// STRIDED: half of every stage is fetched the way k_recombine_mfma_big fetches its R tile from the row-major matrix -- per wave-instruction sixteen 64-byte
// segments of sixteen rows (row stride 15 424 B) -- the other half contiguous (its z tile)
template <int WAVES, bool STRIDED = false>
__global__ __launch_bounds__(WAVES * 64) void k_fill(const char* __restrict__ buf, size_t span, const size_t* __restrict__ base, const size_t* __restrict__ origin, int stages, int inflight, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // two stages of 64 KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t b0 = base[blockIdx.x], org = origin[blockIdx.x], st = 65536;
  constexpr int PIECES = 64 / WAVES;                                // 1 KiB pieces per wave and stage
  for (int s = 0; s < stages; ++s) {
    const size_t off = org + (b0 + (size_t)s * st) % span;
    const char* g = buf + off + (size_t)lane * 16;
    char* l = smem + (s & 1) * 65536;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int piece = wave * PIECES + i;
      if (STRIDED && piece < 32) {
        const char* gs = buf + off + ((size_t)((piece & 15) * 16 + (lane >> 2))) * 15424 + (size_t)((piece >> 4) * 64 + (lane & 3) * 16);
        __builtin_amdgcn_global_load_lds(gs, (lds_void_ptr)(l + piece * 1024), 16, 0, 0);
      } else
      __builtin_amdgcn_global_load_lds(g + (size_t)piece * 1024, (lds_void_ptr)(l + piece * 1024), 16, 0, 0);
    }
    if (inflight == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    else if (s & 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }      // two stages in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[5] == 77 && smem[65536 + 9] == 78) out[0] = 1;
}

int main(int argc, char** argv) {
  int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t BIG = 8ull << 30, PANEL = 237ull << 20;
  char* buf; if (hipMalloc(&buf, BIG) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, BIG);
  int* out; hipMalloc(&out, 64);
  size_t *dbase, *dstep;
  const int maxwg = 2048;
  hipMalloc(&dbase, maxwg * sizeof(size_t)); hipMalloc(&dstep, maxwg * sizeof(size_t));
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  struct Case { const char* name; int mode; size_t span; int wgs_per_cu; int waves; int inflight; };
  const Case cases[] = {
    {"HBM stream: every workgroup its own contiguous slice of 8 GB", 0, BIG, 1, 4, 1},
    {"HBM stream, 8 waves", 0, BIG, 1, 8, 1},
    {"HBM stream, two stages in flight", 0, BIG, 1, 4, 2},
    {"L2: the workgroups of an XCD sweep the same 2 MB", 1, 2ull << 20, 1, 4, 1},
    {"L2, 8 waves", 1, 2ull << 20, 1, 8, 1},
    {"L2, two stages in flight", 1, 2ull << 20, 1, 4, 2},
    {"237 MB panel, every workgroup sweeps all of it, together (same offsets)", 2, PANEL, 1, 4, 1},
    {"237 MB panel, swept from 256 different offsets", 3, PANEL, 1, 4, 1},
    {"237 MB panel, offsets, two stages in flight", 3, PANEL, 1, 4, 2},
    {"237 MB panel, offsets, 8 waves, two stages", 3, PANEL, 1, 8, 2},
    {"recombination shape: 16 strips of the panel, workgroup (i, j) sweeps strip i (16 workgroups per strip, XCD = i % 8)", 4, PANEL, 1, 4, 1},
    {"recombination shape, two stages in flight", 4, PANEL, 1, 4, 2},
    {"recombination shape, 8 waves, two stages", 4, PANEL, 1, 8, 2},
    {"recombination shape, 8 waves, two stages, R half as 64-byte row segments (the kernel's fetch)", 5, PANEL, 1, 8, 2},
    {"237 MB panel from 256 offsets, 8 waves, two stages, R half as 64-byte row segments", 6, PANEL, 1, 8, 2},
  };
  for (const Case& c : cases) {
    const int nwg = cus * c.wgs_per_cu;
    std::vector<size_t> base(nwg), step(nwg);
    const size_t per_wg_bytes = 64ull << 20;                         // every workgroup stages 64 MiB: 16 GB chip-wide
    const int stages = (int)(per_wg_bytes / 65536);
    for (int w = 0; w < nwg; ++w) {
      step[w] = 0;                                                   // (origin)
      if (c.mode == 0) base[w] = (size_t)w * (c.span / nwg) / 65536 * 65536;
      else if (c.mode == 1) base[w] = (size_t)(w % 8) * 0;          // same region for everybody (each XCD's L2 holds its copy)
      else if (c.mode == 2) base[w] = 0;
      else if (c.mode == 3 || c.mode == 6) base[w] = (size_t)w * (c.span / nwg) / 65536 * 65536;
      else { const int i = w % 16; step[w] = (size_t)i * (c.span / 16) / 65536 * 65536; base[w] = (size_t)(w / 16) * 65536 * 4; }      // strip i, neighbours a few stages apart
    }
    hipMemcpy(dbase, base.data(), nwg * sizeof(size_t), hipMemcpyHostToDevice);
    hipMemcpy(dstep, step.data(), nwg * sizeof(size_t), hipMemcpyHostToDevice);
    const size_t span = ((c.mode == 4 || c.mode == 5) ? c.span / 16 : c.span) / 65536 * 65536 - (c.mode >= 5 ? (4u << 20) : 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (c.mode >= 5) hipLaunchKernelGGL((k_fill<8, true>), dim3(nwg), dim3(512), 131072, 0, buf, span, dbase, dstep, stages, c.inflight, out);
      else if (c.waves == 4) hipLaunchKernelGGL(k_fill<4>, dim3(nwg), dim3(256), 131072, 0, buf, span, dbase, dstep, stages, c.inflight, out);
      else hipLaunchKernelGGL(k_fill<8>, dim3(nwg), dim3(512), 131072, 0, buf, span, dbase, dstep, stages, c.inflight, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)nwg * per_wg_bytes;
    std::printf("%-118s %7.2f ms  %6.2f TB/s chip-wide  %6.1f GB/s per CU\n", c.name, best, bytes / best * 1e-9, bytes / best * 1e-6 / cus);
  }
  if (hipGetLastError() != hipSuccess) { std::printf("HIP error\n"); return 2; }
  return 0;
}
