// psfp.hip -- PSFPerturbation behind the C ABI of include/psf_mi355x.h (mp_perturbation.rs:57-62, :193-403).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <atomic>
#include <mutex>
#include <functional>
#include <vector>
#include "../../include/psf_mi355x.h"
#include "psf_host.hpp"
#include "psf_kernels.hpp"
#include "psf_stream_kernels.hpp"
#include "psf_gpv_kernels.hpp"
#include "psf_np_kernels.hpp"
#include "psf_chol_kernels.hpp"
#include "psf_gemm_kernels.hpp"
#include "psf_sdma.hpp"
#include "psf_ntt_api.hpp"

// PSF_KEYGEN_TIMING=1: wall time of the phases of key generation on stderr (each mark drains the device first; off, a mark is one branch)
struct KeygenClock {
  bool on; const char* what; std::chrono::steady_clock::time_point t0, last;
  explicit KeygenClock(const char* w) : on(psf_exp_env("PSF_KEYGEN_TIMING") != nullptr), what(w) { if (on) { hipDeviceSynchronize(); t0 = last = std::chrono::steady_clock::now(); } }
  void mark(const char* phase) {
    if (!on) return;
    hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[keygen %s] %-28s %8.2f ms (at %8.2f)\n", what, phase, std::chrono::duration<double, std::milli>(now - last).count(), std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
};

#define PSFP_FLAG_NO_PERTURB 1u   // internal: handle used as the Z_q / f_a engine of PSFGPV(Ring); no sqrt(Sigma_2) buffers
// PSFP_FLAG_STRUCTURED_SQRT (2u) is public: include/psf_mi355x.h

using namespace psf;

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      std::fprintf(stderr, "[psf_mi355x] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return PSF_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

static inline unsigned grid_for(size_t total, unsigned block = 256, unsigned cap = 256 * 16) {
  size_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}
static inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// Rule of the handle (include/psf_mi355x.h, "Asynchronous calls"): an entry point that rewrites key material or reuses the handle's per-batch buffers first waits
// for the asynchronous samp_p calls in flight (psfp_wait) and returns their status if one failed -- they run on a non-blocking stream and would otherwise read a
// half-replaced key or share dP / dX / dV / the failure words with the new call.
#define PSFP_QUIESCE(h) do { const psf_status rw__ = psfp_wait(h); if (rw__ != PSF_OK) return rw__; } while (0)

struct TimingSlot { std::string name; hipEvent_t e0, e1; };

struct psfp_handle {
  psfp_params prm;
  size_t n, k, mb, w, m;
  uint64_t q, two64, two31;
  bool wide;            // q >= 2^31: two 31-bit limbs
  bool has_key = false;      // A, R and sqrt(Sigma_2) installed
  bool has_pub = false;      // A installed (f_a, check_domain, samp_d work; samp_p needs has_key)
  bool has_R = false;        // R installed (compute_sqrt_sigma_2 can complete the key; samp_p also needs has_pub)
  // key material
  uint64_t* dA = nullptr;      // n x m
  int8_t* dR = nullptr;        // mb x ldr
  size_t ldr = 0;
  double* dLt = nullptr;       // chunk stream of sqrt(Sigma_2) (structured mode: of L_1, the m_bar x m_bar block)
  size_t M_pad = 0, nbi = 0, nkb = 0;
  // structured sqrt(Sigma_2) (PSFP_FLAG_STRUCTURED_SQRT): x_top = L_1 d_1 - g R d_2, x_bot = h d_2
  bool structured = false;
  size_t mL = 0, nbiL = 0;     // order of the stored triangular factor (m, or m_bar) and its row blocks
  int8_t* dR8 = nullptr;       // R tile-packed (k_pack_R8: 4 KiB tiles of 64 rows x 64 columns, contiguous) for k_recombine_mfma_big and k_rd2_mfma, mb_pad x ldr
  bool r8_valid = false;       // dR8 follows dR (ensure_R8)
  bool r8_pending = false;     // the pack has been enqueued on r8_stream and is not known to have completed: other streams wait for evR8
  hipEvent_t evR8 = nullptr; hipStream_t r8_stream = nullptr;
  // compact copies of the key for calls with a handful of preimages, where reading A and R once IS the time of their stages (psf_stream_kernels.hpp):
  // R as two bits per entry (k_recombine_small2; only a {-1, 0, 1} trapdoor has one), A as 32-bit words (k_syndrome_small32; q <= 2^32)
  uint32_t* dR2 = nullptr; uint32_t* dA32 = nullptr; int* dR2bad = nullptr; int* hR2bad = nullptr; hipEvent_t evSmall = nullptr;
  uint32_t* dA32T = nullptr;   // A transposed, [coordinate][row], 32-bit: the fused tail of k_trmm_stream_fused (q <= 2^32, n a multiple of 8)
  uint64_t* dPartF = nullptr; size_t partF_cap = 0;      // its partial residues, [task][row][preimage]
  int small_state = 0;         // 0: stale (the key changed); 1: being built (evSmall); 2: usable; 3: usable, R is not ternary (A32 only)
  double g_const = 0, h_const = 0;
  // gadget tables
  int32_t* dRng = nullptr;
  int32_t* dSk = nullptr; double* dGso = nullptr; double* dNorm2 = nullptr; SampleZParams* dSz = nullptr;
  uint64_t* dGvec = nullptr;
  int8_t* dA8 = nullptr; int NA = 0; size_t n_pad = 0, K_pad = 0;   // balanced base-256 digit planes of A
  ZqConsts zc;
  std::vector<int64_t> hSk; std::vector<double> hGso;
  SampleZParams szR, szSR;
  uint32_t* dSzTab = nullptr; uint32_t szF = 0;      // table screen of the rounding sampler (psf_rng.hpp, k_perturb_round_tab); szF = 0: none (wide words, or the table would not fit)
  // batch work buffers
  size_t Bcap = 0, ld = 0, nbj = 0;
  double* dDt = nullptr; double* dX = nullptr; int32_t* dP = nullptr; uint64_t* dV = nullptr;
  int8_t* dZlo = nullptr; int8_t* dZhi = nullptr; size_t mb_pad = 0;
  int8_t* dP8 = nullptr;                      // three digit planes of P, [K_pad/16][ld][16] each
  uint64_t* dU = nullptr; int64_t* dE = nullptr; uint8_t* dOk = nullptr;
  int* dFail = nullptr;
  // Two sets of the per-batch intermediates: consecutive samp_p calls alternate between them so that the sampling
  // stages of call i (stream aux) overlap the normals + FP64 product of call i+1 (stream s1).
  struct BatchSet { double* dDt = nullptr; double* dX = nullptr; int32_t* dP = nullptr; int8_t* dP8 = nullptr; uint64_t* dV = nullptr;
                    int8_t* dZlo = nullptr; int8_t* dZhi = nullptr; int* dFail = nullptr; int8_t* dD8 = nullptr; } sets[2];
  int8_t* dD8 = nullptr;                      // five digit planes of d_2 2^32 (structured mode), [ldr/16][ld][16] each
  int32_t* dPf = nullptr; int8_t* dP8f = nullptr;   // scratch of f_a (kept apart from the pipelined sets)
  uint64_t* dPart = nullptr; int zq_split_cap = 1;   // per-split residues of the int8-MFMA Z_q product
  bool gadget_queue = true;   // task-queue gadget sampler (PSF_GADGET_QUEUE=0: lock-step kernel)
  bool keep_fail = false;     // sliced host path: the failure flags accumulate over the slices of one call
  // host path: the targets are first read by the syndrome stage, ~50 ms into a C3 batch -- this runs (once) on the calling thread right before that stage is
  // enqueued, i.e. while the product already executes: staging and upload of u cost the call nothing
  std::function<psf_status()> before_u;
  // Host-pointer calls (psfp_samp_p / psfp_samp_p_async): rows are narrowed to int32 on the device, cross PCIe in chunks into pinned buffers and are
  // widened into the caller's int64 rows by worker threads, while the compute stream already runs the next slice / the next call.
  struct HostPipe {
    static constexpr int NW = 8;                // at most this many worker threads per call (each: its own pinned chunk buffers, copies + widening of chunks c = w mod nw)
    int nw = 4;                                 // workers in use (PSF_HOST_WORKERS)
    int32_t* dE32[2] = {nullptr, nullptr};      // device: narrowed rows of the call in flight, two calls deep
    size_t cap_entries[2] = {0, 0};             // entries dE32[slot] holds
    bool slot_ready[2] = {false, false};        // the slot's flags, events, pinned chunk buffers and signals exist
    bool common_ready = false;                  // streams, overflow word, transport
    int32_t* hbuf[2][NW][2] = {};               // pinned chunk buffers [call slot][worker][double buffer]: two calls in flight never share one
    hipEvent_t evC[2][NW][2] = {};              // chunk landed in its pinned buffer
    size_t chunk_entries = 0;
    hipEvent_t evSlice[2][4] = {};              // slice j of call slot s has been narrowed (compute stream)
    int* hFlags[2] = {nullptr, nullptr};        // pinned: [0] sampler failure, [1] unused, [2] int32 overflow of a row entry
    int* dOvf = nullptr;                        // device: overflow flag of the narrowing kernel
    uint64_t* hU[2] = {nullptr, nullptr};       // pinned staging of the targets (a copy from pageable memory would block the caller behind the stream)
    uint64_t* dU2[2] = {nullptr, nullptr};      // device copy of the targets per call in flight (filled by k_copy_words at the head of the call)
    size_t u_cap[2] = {0, 0};
    std::vector<std::thread> workers[2];
    bool busy[2] = {false, false};
    std::atomic<int> status[2] = {{0}, {0}};     // psf_status of the call in each slot (written by its workers)
    size_t next = 0;                            // slot of the next asynchronous call
    uint64_t seq = 0;                           // ticket of the next asynchronous call (0, 1, 2, ... since the handle was created)
    uint64_t slot_seq[2] = {0, 0};              // ticket of the call in each slot
    struct Done { uint64_t seq; int status; bool used; } done[8] = {};      // the last joined calls and their statuses (psfp_wait_ticket)
    bool slice_tail = false;                    // set by psfp_samp_p around its own asynchronous call: cut a short last slice (single-call latency)
    int copy_mode = 1;                          // how a chunk crosses PCIe: 1 = SDMA engine through the HSA runtime (psf_sdma.hpp), 0 = hipMemcpyAsync, 2 = a copy kernel (PSF_HOST_COPY)
    int copy_grid = 32;                         // workgroups of the copy kernel (mode 2)
    psf::SdmaCopy sdma;
    hsa_signal_t sigC[2][NW][2] = {};           // mode 1: chunk landed in its pinned buffer
    hsa_signal_t sigU = {};                     // mode 1: the call's targets have reached the device
    hipStream_t copy = nullptr;                 // D2H stream (high priority)
    hipStream_t compute = nullptr;              // stream of the asynchronous calls' kernels (normal priority)
  } hp;
  // small host-pointer calls (one preimage is the reference's call): u, e and the flags travel through ONE pinned buffer by kernels in stream order, one
  // synchronisation per call -- the straight form (hipMemcpy in, flags out twice, hipMemcpy out: five blocking runtime calls) cost ~70 us around 47 us of kernels
  std::thread hp_warm;        // hp_prewarm's worker; joined by whoever touches the host-pointer machinery next
  uint8_t* sio_pin = nullptr; size_t sio_cap = 0;
  uint64_t* sio_du = nullptr; int64_t* sio_de = nullptr; size_t sio_du_cap = 0, sio_de_cap = 0;      // device side for handles without their own (PSFGPV / ring)
  int32_t* sio_d32 = nullptr; size_t sio_d32_cap = 0;         // narrowed rows of a PSFGPV / ring batch on their way to the host
  hipEvent_t sio_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // pieces 0..3 of such a batch have landed; [4]: its flags have
  bool no_slice = false;      // stage export wants the intermediates of the whole batch
  bool pipeline = false;   // PSF_PIPELINE=1 enables it; measured zero-sum on MI355X (profiles/r01_notes.md)
  size_t ncall = 0;
  uint32_t normals_ncf = 0;   // layout of dDt after the last samp_p: 0 = chunk stream, else the compact stream with this many column fragments
  hipStream_t s1 = nullptr;
  hipEvent_t evT[2] = {nullptr, nullptr}, evP[2] = {nullptr, nullptr}, evIn = nullptr;
  hipStream_t last_stream = nullptr;
  hipStream_t aux = nullptr;                 // low-priority stream of the sampling stages in pipelined mode
  // timing
  bool timing = false;
  std::vector<TimingSlot> slots;
  // psfp_samp_p_multi: this handle's window inside the last call (host clock, ms since the call began)
  const std::chrono::steady_clock::time_point* multi_t0 = nullptr;
  double multi_launched_ms = -1.0, multi_done_ms = -1.0;
};

// The compact copies follow the key without ever blocking a call: the first small call after a key change launches the two packers on its stream and goes on with
// the full-size matrices; a later call finds their event complete and switches over.  PSF_SMALL_COMPACT=0: never.
static void ensure_small_copies(psfp_handle* h, hipStream_t st) {
  static const bool on = [] { const char* e = psf_exp_env("PSF_SMALL_COMPACT"); return !e || std::atoi(e) != 0; }();
  if (!on || (h->prm.flags & PSFP_FLAG_NO_PERTURB) || h->small_state >= 2) return;
  if (h->small_state == 1) {
    if (hipEventQuery(h->evSmall) == hipSuccess) h->small_state = *h->hR2bad ? 3 : 2;
    return;
  }
  const size_t ng = h->ldr / 16;
  if (!h->dR2) {
    if (hipMalloc(&h->dR2, h->mb * ng * sizeof(uint32_t)) != hipSuccess || hipMalloc(&h->dR2bad, sizeof(int)) != hipSuccess ||
        hipHostMalloc(&h->hR2bad, sizeof(int)) != hipSuccess || hipEventCreateWithFlags(&h->evSmall, hipEventDisableTiming) != hipSuccess) { h->small_state = 4; return; }
    if (h->q <= (1ull << 32) && hipMalloc(&h->dA32, h->n * h->m * sizeof(uint32_t)) != hipSuccess) { h->small_state = 4; return; }
    if (h->q <= (1ull << 32) && h->n % 8 == 0 && hipMalloc(&h->dA32T, h->n * h->m * sizeof(uint32_t)) != hipSuccess) { h->dA32T = nullptr; (void)hipGetLastError(); }
  }
  hipMemsetAsync(h->dR2bad, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_pack_R2, dim3(grid_for(h->mb * ng, 256, 256 * 32)), dim3(256), 0, st, h->dR, h->ldr, h->mb, h->dR2, h->dR2bad);
  if (h->dA32) hipLaunchKernelGGL(k_narrow_A32, dim3(grid_for(h->n * h->m, 256, 256 * 32)), dim3(256), 0, st, h->dA, h->n * h->m, h->dA32);
  if (h->dA32T) hipLaunchKernelGGL(k_transpose_A32, dim3((unsigned)((h->m + 31) / 32), (unsigned)((h->n + 31) / 32)), dim3(256), 0, st, h->dA, h->n, h->m, h->dA32T);
  hipMemcpyAsync(h->hR2bad, h->dR2bad, sizeof(int), hipMemcpyDeviceToHost, st);
  hipEventRecord(h->evSmall, st);
  h->small_state = 1;
}

// PSF_RECOMBINE_PACKED=0: k_recombine_mfma_big fetches its R tiles from the row-major matrix as in rounds 2-4 (comparison arm; same bits)
static bool rcb_packed() {
  static const bool on = [] { const char* e = psf_exp_env("PSF_RECOMBINE_PACKED"); return !e || std::atoi(e) != 0; }();
  return on;
}
// the tile-packed copy of R, rebuilt on `st` when R has changed since
// (as the compact copies above: the stream that packs is ordered behind the pack by itself; every OTHER stream that reads dR8 before the pack is known to have
// completed waits for its event -- the halves of PSF_HALVES on s1 / aux, back-to-back device-pointer calls on different non-blocking streams)
static void ensure_R8(psfp_handle* h, hipStream_t st) {
  if (!h->dR8) return;
  if (!h->r8_valid) {
    hipLaunchKernelGGL(k_pack_R8, dim3(grid_for(h->mb_pad * h->ldr, 256, 256 * 64)), dim3(256), 0, st, h->dR, h->ldr, h->mb, h->w, h->mb_pad, h->ldr, h->dR8);
    h->r8_valid = true;
    h->r8_pending = false;
    if (!h->evR8 && hipEventCreateWithFlags(&h->evR8, hipEventDisableTiming) != hipSuccess) { h->evR8 = nullptr; hipStreamSynchronize(st); return; }
    if (hipEventRecord(h->evR8, st) != hipSuccess) { hipStreamSynchronize(st); return; }
    h->r8_pending = true; h->r8_stream = st;
    return;
  }
  if (!h->r8_pending) return;
  if (hipEventQuery(h->evR8) == hipSuccess) { h->r8_pending = false; return; }
  if (st != h->r8_stream) hipStreamWaitEvent(st, h->evR8, 0);
}

static size_t gadget_lds_bytes(size_t k) { return k * k * 8 + k * 8 + k * sizeof(SampleZParams) + k * k * 4 + k * 256 * 4; }

static void select_set(psfp_handle* h, int i) {
  const auto& t = h->sets[i];
  h->dDt = t.dDt; h->dX = t.dX; h->dP = t.dP; h->dP8 = t.dP8; h->dV = t.dV; h->dZlo = t.dZlo; h->dZhi = t.dZhi; h->dFail = t.dFail; h->dD8 = t.dD8;
}

static void free_batch(psfp_handle* h) {
  for (auto& t : h->sets) {
    hipFree(t.dDt); hipFree(t.dX); hipFree(t.dP); hipFree(t.dP8); hipFree(t.dV); hipFree(t.dZlo); hipFree(t.dZhi); hipFree(t.dD8);
    t.dDt = t.dX = nullptr; t.dP = nullptr; t.dP8 = nullptr; t.dV = nullptr; t.dZlo = t.dZhi = nullptr; t.dD8 = nullptr;
  }
  hipFree(h->dPf); hipFree(h->dP8f); hipFree(h->dPart); hipFree(h->dU); hipFree(h->dE); hipFree(h->dOk);
  h->dPf = nullptr; h->dP8f = nullptr; h->dPart = nullptr; h->dU = nullptr; h->dE = nullptr; h->dOk = nullptr;
  select_set(h, 0);
  h->Bcap = 0;
}

// K splits of the Z_q product for `ncols` preimages: at most 256 K-steps each (int32 exactness of the digit-class sums), and enough workgroups to
// fill the chip -- a single call (one preimage) has 8 row tiles x 1 column tile, so its K range is cut as finely as 4 K-steps per split.  The
// residues are exact integers mod q: the number of splits changes no bit.
static int zq_plan(const psfp_handle* h, size_t ncols, int cap) {
  const int nks = (int)(h->K_pad / 64);
  int splits = (nks + 255) / 256;
  const size_t tiles = ((ncols + 63) / 64) * (h->n_pad / 64);
  const int min_ks = tiles * 8 >= 2048 ? 16 : 4;
  while (splits < cap && tiles * splits < 2048 && nks / (splits + 1) >= min_ks) ++splits;
  return splits;
}

static psf_status ensure_batch(psfp_handle* h, size_t B) {
  if (B <= h->Bcap) {
    h->nbj = round_up(B, TR_BN) / TR_BN;
    return PSF_OK;
  }
  HIP_TRY(hipDeviceSynchronize());
  free_batch(h);
  const size_t ld = round_up(B, TR_BN);
  h->ld = ld;
  h->nbj = ld / TR_BN;
  const bool perturb = !(h->prm.flags & PSFP_FLAG_NO_PERTURB);
  const int nsets = (perturb && h->pipeline) ? 2 : 1;
  for (int i = 0; i < nsets; ++i) {
    auto& t = h->sets[i];
    if (perturb) {
      HIP_TRY(hipMalloc(&t.dDt, (ld / TR_BN * h->nkb * TR_CHUNK + TS_SLACK_DOUBLES) * sizeof(double)));   // slack: k_trmm_stream reads past the diagonal
      HIP_TRY(hipMalloc(&t.dX, h->M_pad * ld * sizeof(double)));
      HIP_TRY(hipMalloc(&t.dP, h->M_pad * ld * sizeof(int32_t)));
      HIP_TRY(hipMalloc(&t.dP8, 3 * h->K_pad * ld));
      HIP_TRY(hipMalloc(&t.dV, h->n * ld * sizeof(uint64_t)));
      HIP_TRY(hipMalloc(&t.dZlo, h->ldr * ld + RS_SLACK_SLOTS * 128 * ld));      // [ldr/16][ld][16] (+ the slots k_recombine_wg's ring reads past the last K group)
      HIP_TRY(hipMalloc(&t.dZhi, h->ldr * ld + RS_SLACK_SLOTS * 128 * ld));
      HIP_TRY(hipMemset(t.dZlo, 0, h->ldr * ld));
      HIP_TRY(hipMemset(t.dZhi, 0, h->ldr * ld));
      HIP_TRY(hipMemset(t.dP, 0, h->M_pad * ld * sizeof(int32_t)));
      if (h->structured) {
        HIP_TRY(hipMalloc(&t.dD8, kFixPlanes * h->ldr * ld));
        HIP_TRY(hipMemset(t.dD8, 0, kFixPlanes * h->ldr * ld));
      }
    }
  }
  HIP_TRY(hipMalloc(&h->dPf, h->M_pad * ld * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&h->dP8f, 3 * h->K_pad * ld));
  {  // K splits of the Z_q product (zq_plan): the partial residues of every split live in dPart, [split][n_pad][ld]
    const int nks = (int)(h->K_pad / 64);
    const size_t per_split = h->n_pad * ld * sizeof(uint64_t);
    int cap = (int)(((size_t)256 << 20) / per_split);                 // small batches take up to 64 splits, within 256 MB
    if (cap > 64) cap = 64;
    const int legacy = zq_plan(h, ld, 8);
    h->zq_split_cap = cap > legacy ? cap : legacy;
    if (h->zq_split_cap > nks) h->zq_split_cap = nks;
    HIP_TRY(hipMalloc(&h->dPart, (size_t)h->zq_split_cap * per_split));
  }
  HIP_TRY(hipMalloc(&h->dU, B * h->n * sizeof(uint64_t)));
  HIP_TRY(hipMalloc(&h->dE, B * h->m * sizeof(int64_t)));
  HIP_TRY(hipMalloc(&h->dOk, B));
  // The clears above run on the null stream; the calls that follow may run on non-blocking streams (the host-pointer path's compute stream, a caller's stream), which do
  // not wait for it -- without this barrier a clear could land AFTER the first kernels had written the same buffer (found in round 5 by tools/host_vs_device_fuzz.py: one
  // whole-batch mismatch in 240 000 first calls, small keys whose product finishes within the clear of dP)
  HIP_TRY(hipDeviceSynchronize());
  select_set(h, 0);
  h->Bcap = B;
  return PSF_OK;
}

struct ScopedTimer {
  psfp_handle* h; hipStream_t st; size_t idx; bool on;
  ScopedTimer(psfp_handle* h_, hipStream_t st_, const char* name) : h(h_), st(st_), on(h_->timing) {
    if (!on) return;
    TimingSlot s; s.name = name;
    hipEventCreate(&s.e0); hipEventCreate(&s.e1);
    hipEventRecord(s.e0, st);
    h->slots.push_back(s);
    idx = h->slots.size() - 1;
  }
  ~ScopedTimer() { if (on) hipEventRecord(h->slots[idx].e1, st); }
};
static void clear_slots(psfp_handle* h) {
  for (auto& s : h->slots) { hipEventDestroy(s.e0); hipEventDestroy(s.e1); }
  h->slots.clear();
}

#ifdef TRMM_CLOCK_PROBE
extern "C" void psf_debug_trmm_clk(unsigned long long* out, int reset) {
  if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trmm_clk), sizeof(unsigned long long) * 4);
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_trmm_clk), z, sizeof(z)); }
}
#endif

extern "C" {

const char* psf_status_string(psf_status s) {
  switch (s) {
    case PSF_OK: return "ok";
    case PSF_ERR_PARAM: return "invalid parameter";
    case PSF_ERR_NOT_PD: return "Sigma_2 is not positive definite";
    case PSF_ERR_DOMAIN: return "sigma is not in the domain";
    case PSF_ERR_MODULUS: return "the modulus is too large, the value is potentially not representable";
    case PSF_ERR_NO_SOLUTION: return "the linear system has no solution";
    case PSF_ERR_NO_KEY: return "no key material installed";
    case PSF_ERR_HIP: return "HIP runtime error";
    case PSF_ERR_UNSUPPORTED: return "unsupported parameter combination";
    case PSF_ERR_SAMPLER: return "rejection sampler exceeded its attempt cap or an intermediate left its range";
    default: return "unknown status";
  }
}

psf_status psf_device_info(int device, char* name, size_t name_len, int* compute_units) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return PSF_ERR_HIP;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (name && name_len) { std::strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  return PSF_OK;
}

psf_status psf_gadget_params_default(uint64_t n, uint64_t q, psf_gadget_params* out) { return gadget_params_default(n, q, out); }
psf_status psf_gadget_params_ring_default(uint64_t n, uint64_t q, psf_gadget_params* out) { return gadget_params_ring_default(n, q, out); }

psf_status psf_gen_gadget_vec(uint64_t k, uint64_t base, int64_t* out) {
  if (k < 1 || !out) return PSF_ERR_PARAM;
  const auto g = gen_gadget_vec(k, base);
  std::memcpy(out, g.data(), k * sizeof(int64_t));
  return PSF_OK;
}
psf_status psf_gen_gadget_mat(uint64_t n, uint64_t k, uint64_t base, int64_t* out) {
  if (n < 1 || k < 1 || !out) return PSF_ERR_PARAM;
  const auto G = gen_gadget_mat(n, k, base);
  std::memcpy(out, G.data(), G.size() * sizeof(int64_t));
  return PSF_OK;
}
psf_status psf_short_basis_gadget(const psf_gadget_params* gp, int64_t* out) {
  if (!gp || !out || gp->n < 1 || gp->k < 1) return PSF_ERR_PARAM;
  const auto S = short_basis_gadget(*gp);
  std::memcpy(out, S.data(), S.size() * sizeof(int64_t));
  return PSF_OK;
}
psf_status psf_gen_short_basis_for_trapdoor(const psf_gadget_params* gp, const uint64_t* tag, const uint64_t* A, const int8_t* R, int64_t* out) {
  if (!gp || !A || !R || !out) return PSF_ERR_PARAM;
  std::vector<int64_t> S;
  const psf_status rc = gen_short_basis_for_trapdoor(*gp, tag, A, R, S);
  if (rc != PSF_OK) return rc;
  std::memcpy(out, S.data(), S.size() * sizeof(int64_t));
  return PSF_OK;
}
psf_status psf_rot_minus_matrix(const int64_t* mat, size_t rows, size_t cols, int64_t* out) {
  if (!mat || !out || rows < 1 || cols < 1) return PSF_ERR_PARAM;
  rot_minus_matrix(mat, rows, cols, out);
  return PSF_OK;
}

psf_status psf_find_solution_gadget_mat(int device, const uint64_t* value, size_t rows, size_t cols, uint64_t q, uint64_t k,
                                        uint64_t base, int64_t* out) {
  if (!value || !out || q <= 1 || k < 1 || base < 2) return PSF_ERR_PARAM;
  if (gadget_too_short(base, k, q)) return PSF_ERR_MODULUS;   // gadget_classical.rs:170-172
  if (rows * cols == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(device));
  uint64_t* dv = nullptr; int64_t* dout = nullptr;
  HIP_TRY(hipMalloc(&dv, rows * cols * sizeof(uint64_t)));
  HIP_TRY(hipMalloc(&dout, k * rows * cols * sizeof(int64_t)));
  HIP_TRY(hipMemcpy(dv, value, rows * cols * sizeof(uint64_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_digits, dim3(grid_for(rows * cols)), dim3(256), 0, 0, dv, rows, cols, q, (uint32_t)k, base, dout);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, k * rows * cols * sizeof(int64_t), hipMemcpyDeviceToHost));
  hipFree(dv); hipFree(dout);
  return PSF_OK;
}

// ------------------------------------------------------------------------------------------------------------
static psf_status psfp_init(psfp_handle* h, const psfp_params* prm);

// The screen table of one s (psf_rng.hpp "table screen"): for every candidate index and every bin of delta = ceil(c) - c the 16-bit bounds between which an
// attempt has to be settled exactly.  rho is monotone on either side of a = 0 and a = (idx - ceil(6 s) + delta) / s changes sign only at integers, so the
// extremes over a bin are at its ends; the bins overlap by 1e-9 (the kernel's delta carries one rounding) and the floors get one unit of slack on either side.
static bool build_sz_table(const SampleZParams& sp, std::vector<uint32_t>& T, uint32_t* F_out) {
  if (sp.sh != 16 || sp.n_int < 2) return false;
  uint32_t F = 64;
  while (F >= 8 && (size_t)sp.n_int * F * sizeof(uint32_t) > 40 * 1024) F >>= 1;
  if (F < 8) return false;
  T.assign((size_t)sp.n_int * F, 0);
  auto rs_at = [&](double idx, double delta) {
    const double a = (idx - (double)sp.c6 + delta) * sp.inv_s;
    return det_exp(-3.14159265358979323846 * (a * a)) * 65536.0;
  };
  for (uint32_t idx = 0; idx < sp.n_int; ++idx)
    for (uint32_t b = 0; b < F; ++b) {
      double d0 = (double)b / F - 1e-9, d1 = (double)(b + 1) / F + 1e-9;
      if (d0 < 0.0) d0 = 0.0;
      if (d1 > 1.0) d1 = 1.0;
      const double r0 = rs_at((double)idx, d0), r1 = rs_at((double)idx, d1);
      double lo = r0 < r1 ? r0 : r1, hi = r0 < r1 ? r1 : r0;
      if ((double)idx - (double)sp.c6 + d0 <= 0.0 && (double)idx - (double)sp.c6 + d1 >= 0.0) hi = 65536.0;      // a = 0 inside the bin
      long long A = (long long)std::floor(lo) - 1, R = (long long)std::floor(hi) + 1;
      if (A < 0) A = 0;
      if (R > 65535) R = 65535;
      T[(size_t)idx * F + b] = ((uint32_t)R << 16) | (uint32_t)A;
    }
  *F_out = F;
  return true;
}

psf_status psfp_create(const psfp_params* prm, psfp_handle** out) {
  if (!prm || !out) return PSF_ERR_PARAM;
  const psf_gadget_params& gp = prm->gp;
  if (gp.n < 1 || gp.k < 1 || gp.base < 2 || gp.q <= 1 || gp.q >= (1ull << 62) || gp.m_bar < 1) return PSF_ERR_PARAM;
  if (!(prm->r > 0.0) || !(prm->s > 0.0)) return PSF_ERR_PARAM;
  if (gp.k > 64) return PSF_ERR_UNSUPPORTED;
  {  // every in-domain coordinate obeys |e_i| <= ||e|| <= s r sqrt(m) (mp_perturbation.rs:396-402); the int8 digit planes of the
     // Z_q products (f_a, syndrome) cover |.| < 2^23, so larger domains are refused here instead of being truncated later
    const double m_all = (double)gp.m_bar + (double)gp.n * (double)gp.k;
    if (!(prm->s * prm->r * std::sqrt(m_all) < 8388607.0)) return PSF_ERR_UNSUPPORTED;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || prm->device < 0 || prm->device >= count) {
    std::fprintf(stderr, "[psf_mi355x] no usable HIP device %d (found %d); this library has no CPU fallback\n", prm->device, count);
    return PSF_ERR_HIP;
  }
  HIP_TRY(hipSetDevice(prm->device));
  psfp_handle* h = new psfp_handle();
  const psf_status rc = psfp_init(h, prm);
  if (rc != PSF_OK) { psfp_destroy(h); return rc; }      // whatever was allocated before the failure is released
  *out = h;
  return PSF_OK;
}

static psf_status psfp_init(psfp_handle* h, const psfp_params* prm) {
  const psf_gadget_params& gp = prm->gp;
  h->prm = *prm;
  h->n = gp.n; h->k = gp.k; h->mb = gp.m_bar; h->w = gp.n * gp.k; h->m = h->mb + h->w; h->q = gp.q;
  h->two64 = (uint64_t)((((u128)1) << 64) % gp.q);
  h->two31 = (uint64_t)((1ull << 31) % gp.q);
  h->wide = gp.q >= (1ull << 31);
  h->M_pad = round_up(h->m, TR_BM);
  h->nbi = h->M_pad / TR_BM;
  h->nkb = h->M_pad / TR_BK;
  h->structured = (prm->flags & PSFP_FLAG_STRUCTURED_SQRT) != 0 && !(prm->flags & PSFP_FLAG_NO_PERTURB);
  h->mL = h->structured ? h->mb : h->m;
  h->nbiL = round_up(h->mL, TR_BM) / TR_BM;
  h->ldr = round_up(h->w, 64);          // K of the int8 MFMA product, zero padded
  h->mb_pad = round_up(h->mb, 256);        // rows of R as allocated (zero below m_bar): the 256-row tiles of k_recombine_mfma_big read them all
  h->n_pad = round_up(h->n, 64);
  h->K_pad = round_up(h->m, 64);
  {  // number of balanced base-256 digits so that the top digit of any a < q fits an int8
    uint64_t bound = gp.q - 1;
    h->NA = 1;
    while (bound > 127) { bound = (bound + 128) >> 8; ++h->NA; }
    h->zc.q = gp.q; h->zc.two64 = h->two64;
    uint64_t pw = 1 % gp.q;
    h->zc.inv_q = 1.0 / (double)gp.q;
    for (int c = 0; c < 12; ++c) { h->zc.pw[c] = pw; h->zc.pwd[c] = (double)pw; pw = mulmod_u64(pw, 256 % gp.q, gp.q); }
  }
  HIP_TRY(hipMalloc(&h->dA8, (size_t)h->NA * h->n_pad * h->K_pad));
  h->szR = make_sample_z_params(prm->r);
  {
    std::vector<uint32_t> T;
    uint32_t F = 0;
    if (!(prm->flags & PSFP_FLAG_NO_PERTURB) && build_sz_table(h->szR, T, &F)) {
      HIP_TRY(hipMalloc(&h->dSzTab, T.size() * sizeof(uint32_t)));
      HIP_TRY(hipMemcpy(h->dSzTab, T.data(), T.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      h->szF = F;
    }
  }
  h->szSR = make_sample_z_params(prm->s * prm->r);                    // mp_perturbation.rs:266
  HIP_TRY(hipMalloc(&h->dA, h->n * h->m * sizeof(uint64_t)));
  HIP_TRY(hipMalloc(&h->dR, h->mb_pad * h->ldr + 4096));      // (+ what k_recombine_wg's ring reads past the last row)
  HIP_TRY(hipMemset(h->dR, 0, h->mb_pad * h->ldr));
  if (!(prm->flags & PSFP_FLAG_NO_PERTURB)) HIP_TRY(hipMalloc(&h->dLt, (tr_total_chunks(h->nbiL) * TR_CHUNK + TS_SLACK_DOUBLES) * sizeof(double)));
  if (!(prm->flags & PSFP_FLAG_NO_PERTURB)) HIP_TRY(hipMalloc(&h->dR8, h->mb_pad * h->ldr));      // tile-packed copy of R: k_recombine_mfma_big, k_rd2_mfma
  for (auto& t : h->sets) {                             // [0] sampler failure, [1] some |z| > 127
    HIP_TRY(hipMalloc(&t.dFail, 4 * sizeof(int)));
    HIP_TRY(hipMemset(t.dFail, 0, 4 * sizeof(int)));
  }
  h->dFail = h->sets[0].dFail;
  {  // the FP64 product gets the high-priority queue, the sampling stages the low one
    int lo_prio = 0, hi_prio = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
    HIP_TRY(hipStreamCreateWithPriority(&h->s1, hipStreamNonBlocking, hi_prio));
    HIP_TRY(hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, lo_prio));
  }
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&h->evT[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->evP[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreateWithFlags(&h->evIn, hipEventDisableTiming));
  if (const char* env = psf_exp_env("PSF_PIPELINE")) h->pipeline = std::atoi(env) != 0;
  // gadget part of the trapdoor: (S, S~) of mp_perturbation.rs:233-234, block form
  h->hSk = short_basis_gadget_block(gp);
  std::vector<double> norm2;
  gso_columns(h->hSk, h->k, h->hGso, norm2);
  const double sG = prm->r * std::sqrt((double)(gp.base * gp.base + 1));   // mp_perturbation.rs:180
  std::vector<SampleZParams> sz(h->k);
  for (size_t i = 0; i < h->k; ++i) sz[i] = make_sample_z_params(sG / std::sqrt(norm2[i]));
  std::vector<int32_t> sk32(h->hSk.begin(), h->hSk.end());
  std::vector<int32_t> rng(4 * h->k);
  for (size_t col = 0; col < h->k; ++col) {        // non-zero row ranges of b~_col and b_col (zeros are exact: they can be skipped)
    int glo = (int)h->k, ghi = -1, slo = (int)h->k, shi = -1;
    for (size_t r = 0; r < h->k; ++r) {
      if (h->hGso[r * h->k + col] != 0.0) { if ((int)r < glo) glo = (int)r; ghi = (int)r; }
      if (h->hSk[r * h->k + col] != 0) { if ((int)r < slo) slo = (int)r; shi = (int)r; }
    }
    rng[col] = glo; rng[h->k + col] = ghi; rng[2 * h->k + col] = slo; rng[3 * h->k + col] = shi;
  }
  HIP_TRY(hipMalloc(&h->dRng, rng.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(h->dRng, rng.data(), rng.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  const auto gvec = gen_gadget_vec_mod(h->k, gp.base, gp.q);
  HIP_TRY(hipMalloc(&h->dSk, sk32.size() * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&h->dGso, h->hGso.size() * sizeof(double)));
  HIP_TRY(hipMalloc(&h->dNorm2, h->k * sizeof(double)));
  HIP_TRY(hipMalloc(&h->dSz, h->k * sizeof(SampleZParams)));
  HIP_TRY(hipMalloc(&h->dGvec, h->k * sizeof(uint64_t)));
  HIP_TRY(hipMemcpy(h->dSk, sk32.data(), sk32.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dGso, h->hGso.data(), h->hGso.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dNorm2, norm2.data(), h->k * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dSz, sz.data(), h->k * sizeof(SampleZParams), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dGvec, gvec.data(), h->k * sizeof(uint64_t), hipMemcpyHostToDevice));
#ifdef PSF_EXPERIMENTS
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_diag), hipFuncAttributeMaxDynamicSharedMemorySize, CH_NB * (CH_NB + 1) * 8));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, (CH_NB * (CH_NB + 1) / 2 + CH_NB * 64) * 8));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_f64), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TR_CHUNK * sizeof(double)));
#endif
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_mfma_big), hipFuncAttributeMaxDynamicSharedMemorySize, RCB_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_small<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_small<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_small<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_small2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_small2<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_small2<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rd2_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (1 + kFixPlanes) * 4096));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gadget), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gadget_lds_bytes(h->k)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_wg<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RW_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_wg<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RW_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_wg<3, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RW_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recombine_wg<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RW_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_stream_wg32<TSW_H, TSW_NBUF, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TSW32_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_stream_wg32<TSW_H, TSW_NBUF, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TSW32_LDS));
#ifdef PSF_EXPERIMENTS
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_stream_wg<2, TSW_NBUF, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TSW128_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_stream_wg<2, TSW_NBUF, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TSW128_LDS));
#endif
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_stream_wg<TSW64_H, TSW64_NBUF, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TSW_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trmm_stream_wg<TSW64_H, TSW64_NBUF, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TSW_LDS));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gadget_queue<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gadget_queue_lds_bytes(h->k)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gadget_queue<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gadget_queue_lds_bytes(h->k)));
  if (const char* env = psf_exp_env("PSF_GADGET_QUEUE")) h->gadget_queue = std::atoi(env) != 0;
  for (int64_t v : h->hSk) if (v > 32767 || v < -32768) h->gadget_queue = false;      // the queue kernel keeps S_k in int16
  return PSF_OK;
}

static void hp_release(psfp_handle* h);
static psf_status hp_ensure(psfp_handle* h, int slot, size_t entries, size_t u_words, bool from_prewarm = false);
// A key whose batches will cross PCIe (m >= 8192: a preimage is >= 64 KiB) gets the batch-independent part of the host-pointer machinery -- streams, the DMA path,
// one slot's pinned rings -- when the key is installed, next to a factorisation that takes a quarter of a second, instead of inside the first samp_p call
// (PSF_HOST_PREWARM=0: on first use, as for small keys).
static void hp_prewarm(psfp_handle* h) {
  if (h->m < 8192) return;
  if (const char* env = psf_exp_env("PSF_HOST_PREWARM")) if (std::atoi(env) == 0) return;
  if (h->hp_warm.joinable()) h->hp_warm.join();
  try {
    h->hp_warm = std::thread([h]() { if (hipSetDevice(h->prm.device) == hipSuccess) (void)hp_ensure(h, 0, 0, 0, true); });      // beside the factorisation, not behind it
  } catch (...) { }                                         // no thread: on first use then
}
void psfp_destroy(psfp_handle* h) {
  if (!h) return;
  hipSetDevice(h->prm.device);
  hp_release(h);
  if (h->sio_pin) hipHostFree(h->sio_pin);
  hipFree(h->sio_du); hipFree(h->sio_de); hipFree(h->sio_d32);
  for (auto& ev : h->sio_ev) if (ev) hipEventDestroy(ev);
  free_batch(h);
  clear_slots(h);
  if (h->aux) hipStreamDestroy(h->aux);
  hipFree(h->dA); hipFree(h->dR); hipFree(h->dLt); hipFree(h->dR8); hipFree(h->dR2); hipFree(h->dA32); hipFree(h->dR2bad); hipFree(h->dA32T); hipFree(h->dPartF);
  if (h->hR2bad) hipHostFree(h->hR2bad);
  if (h->evSmall) hipEventDestroy(h->evSmall);
  if (h->evR8) hipEventDestroy(h->evR8);
  for (auto& t : h->sets) hipFree(t.dFail);
  if (h->s1) hipStreamDestroy(h->s1);
  for (int i = 0; i < 2; ++i) { if (h->evT[i]) hipEventDestroy(h->evT[i]); if (h->evP[i]) hipEventDestroy(h->evP[i]); }
  if (h->evIn) hipEventDestroy(h->evIn);
  hipFree(h->dA8); hipFree(h->dSzTab);
  hipFree(h->dRng);
  hipFree(h->dSk); hipFree(h->dGso); hipFree(h->dNorm2); hipFree(h->dSz); hipFree(h->dGvec);
  delete h;
}

size_t psfp_m(const psfp_handle* h) { return h ? h->m : 0; }

// Sigma_2 assembly (dense lower) + Cholesky + repack.  mp_perturbation.rs:111-139.
// Structured mode factors Sigma_2 = c [[alpha I - kappa R R^t, -kappa R], [-kappa R^t, beta I]]  (c = r^2 / 2 pi, kappa = b^2 + 1, alpha = s^2 - 1,
// beta = alpha - kappa) as B B^t with B = [[L_1 / sqrt c, -kappa R / sqrt beta], [0, sqrt beta I]] sqrt c, where L_1 is the Cholesky factor of
// c (alpha I - kappa (alpha / beta) R R^t): only that m_bar x m_bar block is assembled, factored and stored.
// Cholesky of Sigma_2 directly on the key's chunk stream (psf_chol_kernels.hpp, "Cholesky directly on the key's chunk stream"): no dense m x m matrix.
// factor + invert one 128 x 128 diagonal block: the blocked kernel (round 6); PSF_CHOL_DIAG=steps (experiments build): the step-by-step kernel of rounds 3-5
static void launch_chol_diag(hipStream_t st, double* P, size_t ld, size_t off, int nb, double* dLi, int* dinfo, size_t report_base) {
#ifdef PSF_EXPERIMENTS
  static const bool steps = [] { const char* e = psf_exp_env("PSF_CHOL_DIAG"); return e && !std::strcmp(e, "steps"); }();
  if (steps) {
    hipLaunchKernelGGL(k_chol_diag_inv_steps, dim3(1), dim3(256), ((size_t)CH_NB * (CH_NB + 1) + 2 * CH_NB) * sizeof(double), st, P, ld, off, nb, dLi, dinfo, report_base);
    return;
  }
#endif
  hipLaunchKernelGGL(k_chol_diag_inv, dim3(1), dim3(256), CH_DIAG_LDS, st, P, ld, off, nb, dLi, dinfo, report_base);
}
static bool prepare_chol_diag() {
  bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_diag_inv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_DIAG_LDS) == hipSuccess;
#ifdef PSF_EXPERIMENTS
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_diag_inv_steps), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)CH_NB * (CH_NB + 1) + 2 * CH_NB) * sizeof(double))) == hipSuccess;
#endif
  return ok;
}

// dS_dense: nullptr, or Sigma_2 already assembled as a dense m x m matrix (lower triangle; the hybrid of build_sqrt_sigma2): the panels are then copied out of it
// instead of being assembled one by one -- R R^t on the int8 matrix cores takes 5.5 ms for the whole of C3's Sigma_2 at once and 76 ms in 121 panel-sized pieces.
static psf_status build_sqrt_sigma2_stream(psfp_handle* h, double nf_r2, double s2, double b2p1, const double* d_sig, const double* dS_dense = nullptr) {
  const size_t m = h->mL;
  const int nbi = (int)h->nbiL;
  const int nP = (nbi + 1) / 2;                                       // panels of two column blocks (256 columns)
  const size_t PW = 2 * TR_BM;                                        // leading dimension of a panel buffer
  const size_t prow = (size_t)nbi * TR_BM;                            // its rows (panel 0 needs them all)
  // K splits of the update.  One workgroup occupies a CU (356 registers per lane), so a launch runs in rounds of 256 workgroups, each as long as one split
  // plus its epilogue (the 128 x 256 partial tile goes to the workspace and is read back by the reduce: about 1.5 units' worth of time, a unit being two
  // chunks = 32 columns of L).  The split count minimises rounds x (units per split + 1.5) over at most 4096 workgroups: without it the middle panels of a large key, whose 257..511 row
  // blocks are just over one round, leave up to half of the chip idle.
  auto splits_for = [](int wgs, int units) {
    if (wgs <= 0 || units <= 1) return 1;
    int best = 1; double best_cost = 1e300;
    for (int sp = 1; sp <= units && sp <= 64; ++sp) {
      if (sp > 1 && (size_t)wgs * sp > 4096) break;                    // workspace: 256 KB per workgroup, 1 GB at most
      const double cost = (double)(((size_t)wgs * sp + 255) / 256) * ((double)((units + sp - 1) / sp) + 1.5);
      if (cost < best_cost * 0.999) { best_cost = cost; best = sp; }
    }
    return best;
  };
  size_t ws_doubles = 0;
  for (int J = 1; J < nP; ++J) {
    const int nrb = nbi - 2 * J, head = nrb < 2 ? nrb : 2;
    const size_t a = (size_t)splits_for(head, 8 * J) * head * TR_BM * PW;
    const size_t b = nrb > 2 ? (size_t)splits_for(nrb - 2, 8 * J) * (size_t)(nrb - 2) * TR_BM * PW : 0;
    if (a > ws_doubles) ws_doubles = a;
    if (b > ws_doubles) ws_doubles = b;
  }
  double *dPn[2] = {nullptr, nullptr}, *dLi = nullptr, *dWs = nullptr; int* dinfo = nullptr;
  hipStream_t sm = nullptr, ss = nullptr, sd = nullptr;               // factorisation / Sigma_2 panels (one panel ahead) / the diagonal blocks' chain (look-ahead)
  hipEvent_t evSig[2] = {nullptr, nullptr}, evPack[2] = {nullptr, nullptr}, evHead = nullptr, evDiag = nullptr;
  GemmWorkspace w, w2;                                                // the in-panel products have K = 128: never cut
  bool lookahead = true;
  if (const char* e = psf_exp_env("PSF_CHOL_LOOKAHEAD")) lookahead = std::atoi(e) != 0;
  auto cleanup = [&]() {
    for (hipStream_t st : {sm, ss, sd}) if (st) hipStreamDestroy(st);
    for (hipEvent_t ev : {evSig[0], evSig[1], evPack[0], evPack[1], evHead, evDiag}) if (ev) hipEventDestroy(ev);
    hipFree(dPn[0]); hipFree(dPn[1]); hipFree(dLi); hipFree(dWs); hipFree(dinfo);
  };
  bool ok = gemm_prepare() == hipSuccess && prepare_chol_diag() &&
            hipMalloc(&dPn[0], prow * PW * sizeof(double)) == hipSuccess && hipMalloc(&dPn[1], prow * PW * sizeof(double)) == hipSuccess &&
            hipMalloc(&dLi, 2 * CH_NB * CH_NB * sizeof(double)) == hipSuccess && (!ws_doubles || hipMalloc(&dWs, ws_doubles * sizeof(double)) == hipSuccess) &&
            hipMalloc(&dinfo, sizeof(int)) == hipSuccess && hipMemset(dinfo, 0, sizeof(int)) == hipSuccess &&
            hipMemset(dPn[0], 0, prow * PW * sizeof(double)) == hipSuccess && hipMemset(dPn[1], 0, prow * PW * sizeof(double)) == hipSuccess;
  for (hipStream_t* st : {&sm, &ss, &sd}) ok = ok && hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
  for (hipEvent_t* ev : {&evSig[0], &evSig[1], &evPack[0], &evPack[1], &evHead, &evDiag}) ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
  if (!ok) { cleanup(); return PSF_ERR_HIP; }
  if (hipStreamSynchronize(nullptr) != hipSuccess) { cleanup(); return PSF_ERR_HIP; }      // R, the dense Sigma_2 (and k_pack_R8) were produced on the default stream (not a device-wide wait: psfp_trap_gen computes A on a side stream meanwhile)
  // Sigma_2 restricted to panel J (rows off.., columns off..off+255), dense with leading dimension 256; it does not depend on the factorisation, so it
  // is assembled one panel ahead on its own stream into the other of two panel buffers
  auto sigma_panel = [&](int J) {
    const size_t off = (size_t)J * PW;
    const size_t cols = m - off < PW ? m - off : PW;
    const dim3 sg((unsigned)((cols + 63) / 64), (unsigned)((m - off + 63) / 64));
    if (dS_dense) {
      hipLaunchKernelGGL(k_chol_copy_panel, dim3(grid_for((m - off) * cols, 256, 4096)), dim3(256), 0, ss, dS_dense, m, m, off, cols, dPn[J & 1], PW);
    } else {
      hipLaunchKernelGGL(k_sigma2_rrt, sg, dim3(256), 3 * 2 * 4096, ss, h->dR, h->ldr, h->mb, m, nf_r2, s2, b2p1, d_sig, dPn[J & 1], PW, off, off);
      hipLaunchKernelGGL(k_sigma2, sg, dim3(256), 0, ss, h->dR, h->ldr, h->mb, h->w, m, nf_r2, s2, b2p1, d_sig, dPn[J & 1], PW, off, off, 1);
    }
    hipEventRecord(evSig[J & 1], ss);
  };
  sigma_panel(0);
  for (int J = 0; J < nP; ++J) {
    const size_t off = (size_t)J * PW;
    const int ncb = 2 * J + 1 < nbi ? 2 : 1;                          // column blocks of this panel
    const int nrb = nbi - 2 * J;                                      // its row blocks
    const size_t nb0 = m - off < (size_t)TR_BM ? m - off : (size_t)TR_BM;
    const size_t below0 = m - off - nb0;                              // rows under the first diagonal block
    const size_t nb1 = ncb == 2 ? (below0 < (size_t)TR_BM ? below0 : (size_t)TR_BM) : 0;
    const size_t below1 = ncb == 2 ? below0 - nb1 : 0;                // rows under the second diagonal block
    double* P = dPn[J & 1];
    if (J + 1 < nP) {
      if (J >= 1) hipStreamWaitEvent(ss, evPack[(J + 1) & 1], 0);     // the buffer's previous panel (J - 1) has been packed
      sigma_panel(J + 1);
    }
    hipStreamWaitEvent(sm, evSig[J & 1], 0);
    auto update = [&](int rb0, int count) {                            // P -= L[rows of the panel, columns < off] L[panel's row blocks, columns < off]^t
      const int sp = splits_for(count, 8 * J);
      const size_t stride = (size_t)count * TR_BM * PW;               // doubles per split in the workspace
      const size_t first = (size_t)rb0 * TR_BM * PW;
      hipLaunchKernelGGL(k_chol_update_big, dim3((unsigned)count, (unsigned)sp), dim3(256), 0, sm, h->dLt, J, nbi, rb0, 8 * J, dWs - first, stride);
      hipLaunchKernelGGL(k_chol_panel_reduce, dim3(grid_for(stride, 256, 2048)), dim3(256), 0, sm, P + first, dWs, stride, sp, (size_t)0, stride);
    };
    if (J > 0) {
      const int head = nrb < 2 ? nrb : 2;
      update(0, head);                                                // the two row blocks that hold the diagonal blocks (cut finely along K), then the rest
      if (lookahead) hipEventRecord(evHead, sm);
      else if (nrb > 2) update(2, nrb - 2);
    } else if (lookahead) hipEventRecord(evHead, sm);
    double* const Li0 = dLi;
    double* const Li1 = lookahead ? dLi + (size_t)CH_NB * CH_NB : dLi;
    if (lookahead) {
      // Round 6: the two diagonal blocks of the panel need only the HEAD of the update.  Their chain -- factor + invert block (0,0), L10 = P10 L00^-t, P11 -= L10 L10^t,
      // factor + invert block (1,1): two single-workgroup kernels and two 128^3 products, 0.6 ms of pure latency per panel -- runs on a second stream while the rest
      // of the update (the rows below, ~1 ms of matrix work per panel in the middle of the factorisation) occupies the chip; the panel's triangular solves wait for both.
      hipStreamWaitEvent(sd, evHead, 0);
      launch_chol_diag(sd, P, PW, (size_t)0, (int)nb0, Li0, dinfo, off);
      if (ncb == 2) {
        launch_gemm<true>(sd, GemmArgs{P + TR_BM * PW, PW, Li0, (size_t)CH_NB, P + TR_BM * PW, PW, nb1, nb0, nb0, 1.0, 0.0, nullptr, nullptr, 0}, w2);
        launch_gemm<true>(sd, GemmArgs{P + TR_BM * PW, PW, P + TR_BM * PW, PW, P + TR_BM * PW + TR_BM, PW, nb1, nb1, (size_t)TR_BM, -1.0, 1.0, nullptr, nullptr, 0}, w2);
        launch_chol_diag(sd, P, PW, (size_t)TR_BM, (int)nb1, Li1, dinfo, off + TR_BM);
      }
      hipEventRecord(evDiag, sd);
      if (J > 0 && nrb > 2) update(2, nrb - 2);
      hipStreamWaitEvent(sm, evDiag, 0);
      if (below1) {                                                   // the rows under both diagonal blocks: solve, take column block 0 out of column block 1, solve
        double* const Pr = P + 2 * TR_BM * PW;
        launch_gemm<true>(sm, GemmArgs{Pr, PW, Li0, (size_t)CH_NB, Pr, PW, below1, nb0, nb0, 1.0, 0.0, nullptr, nullptr, 0}, w);
        launch_gemm<true>(sm, GemmArgs{Pr, PW, P + TR_BM * PW, PW, Pr + TR_BM, PW, below1, nb1, (size_t)TR_BM, -1.0, 1.0, nullptr, nullptr, 0}, w);
        launch_gemm<true>(sm, GemmArgs{Pr + TR_BM, PW, Li1, (size_t)CH_NB, Pr + TR_BM, PW, below1, nb1, nb1, 1.0, 0.0, nullptr, nullptr, 0}, w);
      }
    } else {
    // inside the panel: factor + invert the first diagonal block, solve its rows below, take its contribution out of the second column block, the same again
    launch_chol_diag(sm, P, PW, (size_t)0, (int)nb0, dLi, dinfo, off);
    if (below0)
      launch_gemm<true>(sm, GemmArgs{P + TR_BM * PW, PW, dLi, (size_t)CH_NB, P + TR_BM * PW, PW, below0, nb0, nb0, 1.0, 0.0, nullptr, nullptr, 0}, w);
    if (ncb == 2) {
      launch_gemm<true>(sm, GemmArgs{P + TR_BM * PW, PW, P + TR_BM * PW, PW, P + TR_BM * PW + TR_BM, PW, below0, nb1, (size_t)TR_BM, -1.0, 1.0, nullptr, nullptr, 0}, w);
      launch_chol_diag(sm, P, PW, (size_t)TR_BM, (int)nb1, dLi, dinfo, off + TR_BM);
      if (below1)
        launch_gemm<true>(sm, GemmArgs{P + 2 * TR_BM * PW + TR_BM, PW, dLi, (size_t)CH_NB, P + 2 * TR_BM * PW + TR_BM, PW, below1, nb1, nb1, 1.0, 0.0, nullptr, nullptr, 0}, w);
    }
    }
    hipLaunchKernelGGL(k_chol_pack_panel, dim3(grid_for((size_t)nrb * ncb * 8 * TR_CHUNK, 256, 4096)), dim3(256), 0, sm, P, J, ncb, nbi, m, h->dLt);
    hipEventRecord(evPack[J & 1], sm);
  }
  hipError_t ce = hipStreamSynchronize(sm);
  if (ce == hipSuccess) ce = hipStreamSynchronize(ss);
  if (ce == hipSuccess) ce = hipStreamSynchronize(sd);
  if (ce == hipSuccess) ce = hipGetLastError();
  int info = -1;
  if (ce == hipSuccess) ce = hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost);
  cleanup();
  if (ce != hipSuccess) return PSF_ERR_HIP;
  return info != 0 ? PSF_ERR_NOT_PD : PSF_OK;                         // mp_perturbation.rs:109-110
}

static psf_status build_sqrt_sigma2(psfp_handle* h, double s_cov, const double* d_sigma_packed = nullptr) {
  const double TWO_PI = 6.283185307179586476925;
  const double nf_r2 = (1.0 / TWO_PI) * (h->prm.r * h->prm.r);
  const double s2 = s_cov * s_cov;
  double b2p1 = (double)(h->prm.gp.base * h->prm.gp.base + 1);
  const size_t m = h->mL;
  if (h->structured) {
    const double kappa = b2p1, alpha = s2 - 1.0, beta = alpha - kappa;
    if (!(beta > 0.0)) return PSF_ERR_NOT_PD;
    b2p1 = kappa * (alpha / beta);                                   // the R R^t block of the Schur-complemented top-left corner
    h->g_const = (std::sqrt(nf_r2) * kappa) / std::sqrt(beta);
    h->h_const = std::sqrt(nf_r2 * beta);
    ensure_R8(h, nullptr);
  }
  {
    // "stream": left-looking on the key's chunk stream, no dense matrix (C5: 10.1 s against 21.1 s, and 121 GB less memory; C3: 0.28 s either way); "gemm": left-looking on a
    // dense m x m matrix with the LDS-staged GEMM; "right": the right-looking kernels of rounds 1-2.  Default by size: the
    // dense form while the matrix stays below 16 GB (m < 46 341), the stream form above.  (The dense form factors the diagonal blocks beside the updates on a
    // second stream; beside k_chol_update_big that overlap returns nothing -- FP64 MFMAs and the vector work of the triangular kernel share
    // one pipe, profiles/r03_notes.md -- so the stream form runs them in line: 0.3 s of C5's total.)
    const char* ce = psf_exp_env("PSF_CHOL");
    const bool stream = ce ? !std::strcmp(ce, "stream") : m * m * sizeof(double) > (16ull << 30);
    if (stream) return build_sqrt_sigma2_stream(h, nf_r2, s2, b2p1, d_sigma_packed);
  }
  struct DenseGuard { double* dS = nullptr; int* dinfo = nullptr; ~DenseGuard() { hipFree(dinfo); hipFree(dS); } } dg;      // every exit releases both
  double*& dS = dg.dS;
  HIP_TRY(hipMalloc(&dS, m * m * sizeof(double)));
  HIP_TRY(hipMemset(dS, 0, m * m * sizeof(double)));
  const unsigned tiles = (unsigned)((m + 63) / 64);
  // the R R^t block on the int8 matrix cores, everything else (the rows from m_bar on) in k_sigma2
  hipLaunchKernelGGL(k_sigma2_rrt, dim3(tiles, tiles), dim3(256), 3 * 2 * 4096, 0, h->dR, h->ldr, h->mb, m, nf_r2, s2, b2p1, d_sigma_packed, dS, m, (size_t)0, (size_t)0);
  hipLaunchKernelGGL(k_sigma2, dim3(tiles, tiles), dim3(256), 0, 0, h->dR, h->ldr, h->mb, h->w, m, nf_r2, s2, b2p1, d_sigma_packed, dS, m, (size_t)0, (size_t)0, 1);
  HIP_TRY(hipGetLastError());
  {
    // The hybrid (round 6, the default below 16 GB): Sigma_2 dense AT ONCE (above), the factorisation on the key's chunk stream (k_chol_update_big: 64 TFLOP/s against
    // the 44 of the LDS-staged GEMM of the dense left-looking form), panels copied out of the dense matrix.  PSF_CHOL=gemm (experiments build): the dense form.
    const char* ce2 = psf_exp_env("PSF_CHOL");
    if (!(ce2 && (!std::strcmp(ce2, "gemm") || !std::strcmp(ce2, "right")))) return build_sqrt_sigma2_stream(h, nf_r2, s2, b2p1, d_sigma_packed, dS);
  }
  // blocked Cholesky of the lower triangle, panel width 128 (psf_chol_kernels.hpp): left-looking on the FP64 GEMM; PSF_CHOL=right: the right-looking
  // kernels of rounds 1-2 (comparison arm)
  int*& dinfo = dg.dinfo;
  HIP_TRY(hipMalloc(&dinfo, sizeof(int)));
  HIP_TRY(hipMemset(dinfo, 0, sizeof(int)));
#ifdef PSF_EXPERIMENTS
  const char* chol_env = psf_exp_env("PSF_CHOL");
  if (chol_env && !std::strcmp(chol_env, "right")) {
    for (size_t off = 0; off < m; off += CH_NB) {
      const int nb = (int)(m - off < (size_t)CH_NB ? m - off : (size_t)CH_NB);
      hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), (size_t)nb * (CH_NB + 1) * sizeof(double), 0, dS, m, off, nb, dinfo);
      const size_t rest = m - off - nb;
      if (rest == 0) break;
      const size_t trsm_lds = ((size_t)nb * (nb + 1) / 2 + (size_t)nb * 64) * sizeof(double);
      hipLaunchKernelGGL(k_chol_trsm, dim3((unsigned)((rest + 63) / 64)), dim3(64), trsm_lds, 0, dS, m, off, nb, m, dinfo);
      const size_t nt = (rest + 127) / 128;
      hipLaunchKernelGGL(k_chol_syrk, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 2 * 4096 * sizeof(double), 0, dS, m, off, m, (int)nt, dinfo);
    }
  } else
#endif
  {
    if (gemm_prepare() != hipSuccess) return PSF_ERR_HIP;
    GemmWorkspace w;
    w.bytes = (size_t)900 * GM_T * GM_T * sizeof(double);              // < 384 + 512 (tile, split) pairs per launch, see launch_gemm
    double* dLi = nullptr;
    hipStream_t sm = nullptr, sd = nullptr;                            // products / diagonal blocks
    hipEvent_t evTile = nullptr, evDiag = nullptr;
    auto cleanup = [&]() { if (sm) hipStreamDestroy(sm); if (sd) hipStreamDestroy(sd); if (evTile) hipEventDestroy(evTile); if (evDiag) hipEventDestroy(evDiag); hipFree(w.ws); hipFree(dLi); };
    if (hipMalloc(&w.ws, w.bytes) != hipSuccess || hipMalloc(&dLi, CH_NB * CH_NB * sizeof(double)) != hipSuccess ||
        hipStreamCreateWithFlags(&sm, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&sd, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&evTile, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&evDiag, hipEventDisableTiming) != hipSuccess ||
        !prepare_chol_diag() ||
        hipDeviceSynchronize() != hipSuccess) {                        // (k_sigma2 ran on the default stream)
      cleanup(); return PSF_ERR_HIP;
    }
    // Look-ahead: the factorisation of a diagonal block is one workgroup walking 128 dependent steps (~0.2-0.35 ms, pure latency).  The update of
    // panel j is therefore cut in two: its first row tile (the diagonal block) goes first, its factorisation + inversion then runs on a second
    // stream BESIDE the update of the rows below, and the triangular solve of those rows (a product with the inverse) joins the two.
    for (size_t off = 0; off < m; off += CH_NB) {
      const size_t nb = m - off < (size_t)CH_NB ? m - off : (size_t)CH_NB;
      const size_t rest = m - off - nb;
      double* P = dS + off * m + off;                                  // the panel: rows off.., columns off..off+nb
      double* P2 = P + nb * m;                                         // its rows below the diagonal block
      if (off > 0) {                                                   // P -= L[off.., 0..off) L[off..off+nb, 0..off)^t
        launch_gemm<true>(sm, GemmArgs{dS + off * m, m, dS + off * m, m, P, m, nb, nb, off, -1.0, 1.0, nullptr, nullptr, 0}, w);
        hipEventRecord(evTile, sm);
        if (rest) launch_gemm<true>(sm, GemmArgs{dS + (off + nb) * m, m, dS + off * m, m, P2, m, rest, nb, off, -1.0, 1.0, nullptr, nullptr, 0}, w);
        hipStreamWaitEvent(sd, evTile, 0);
      }
      launch_chol_diag(sd, dS, m, off, (int)nb, dLi, dinfo, off);
      hipEventRecord(evDiag, sd);
      hipStreamWaitEvent(sm, evDiag, 0);
      if (rest == 0) break;
      // rows below = panel L11^-t; in place: one column tile, a workgroup reads only its own rows
      launch_gemm<true>(sm, GemmArgs{P2, m, dLi, (size_t)CH_NB, P2, m, rest, nb, nb, 1.0, 0.0, nullptr, nullptr, 0}, w);
    }
    const hipError_t ce = hipStreamSynchronize(sm);
    hipStreamSynchronize(sd);
    cleanup();
    if (ce != hipSuccess) return PSF_ERR_HIP;
  }
  HIP_TRY(hipGetLastError());
  int info = -1;
  HIP_TRY(hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost));
  if (info != 0) return PSF_ERR_NOT_PD;                              // mp_perturbation.rs:109-110
  hipLaunchKernelGGL(k_repack_L<false>, dim3(grid_for(tr_total_chunks(h->nbiL) * TR_CHUNK)), dim3(256), 0, 0, dS, m, m, h->dLt, h->nbiL);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return PSF_OK;
}

// A[:, m_bar:] = G - A_bar R  (gadget_classical.rs:66): the one product still on the limb kernel (setup path, R in int8)
static void launch_zq_trapdoor(psfp_handle* h, const uint64_t* d_tag = nullptr, hipStream_t st = nullptr) {
  dim3 grid((unsigned)((h->w + 63) / 64), (unsigned)((h->n + 63) / 64));
  if (h->wide)
    hipLaunchKernelGGL((k_zq_matmul<int8_t, true>), grid, dim3(256), 0, st, (int)ZQ_TRAPDOOR, h->dA, h->m, (size_t)0, h->n, h->mb, h->dR, h->ldr, h->w,
                       h->q, h->two64, h->two31, (const uint64_t*)nullptr, h->dA, h->m, h->mb, h->dGvec, (uint64_t)h->k, d_tag);
  else
    hipLaunchKernelGGL((k_zq_matmul<int8_t, false>), grid, dim3(256), 0, st, (int)ZQ_TRAPDOOR, h->dA, h->m, (size_t)0, h->n, h->mb, h->dR, h->ldr, h->w,
                       h->q, h->two64, h->two31, (const uint64_t*)nullptr, h->dA, h->m, h->mb, h->dGvec, (uint64_t)h->k, d_tag);
}

static void split_A(psfp_handle* h, hipStream_t st = nullptr) {
  hipLaunchKernelGGL(k_split_A, dim3(grid_for(h->n_pad * h->K_pad)), dim3(256), 0, st, h->dA, h->m, h->n, h->m, h->n_pad, h->K_pad, h->NA, h->dA8);
}

// out = (mode syndrome) U - A P  or  (mode f_a) A P for the columns [col0, col0 + ncols), with P (K x ld int32) first cut into digit planes
static void launch_zq_mfma(psfp_handle* h, hipStream_t st, int mode, const int32_t* P, int8_t* P8, size_t ncols, const uint64_t* U, uint64_t* out, size_t ldo, size_t col0 = 0) {
  const size_t ld = h->ld;
  {  // a handful of preimages: A streamed once as 64-bit words (k_syndrome_small; PSF_SYNDROME_SMALL = largest batch it serves, 0: never)
    size_t small_max = 1;                                             // measured at C3: 39 vs 48 us at one preimage, 52 vs 50 at two, 81 vs 50 at four (64-bit multiply-adds)
    if (const char* e = psf_exp_env("PSF_SYNDROME_SMALL")) small_max = (size_t)std::atol(e);
    if (small_max > 4) small_max = 4;
    const int splits = (int)((h->m + SYN_KLEN - 1) / SYN_KLEN);
    if (mode == ZQ_SYNDROME && P == h->dP && ncols <= small_max && splits <= h->zq_split_cap && splits <= 64) {
      const int rows_per_wg = 16;
      dim3 grid((unsigned)splits, (unsigned)((h->n + rows_per_wg - 1) / rows_per_wg));
      ensure_small_copies(h, st);
      if (h->dA32 && (h->small_state == 2 || h->small_state == 3)) {
        if (ncols == 1) hipLaunchKernelGGL(k_syndrome_small32<1>, grid, dim3(512), 0, st, h->dA32, h->n, h->m, P, ld, ncols, h->q, rows_per_wg, h->dPart, h->n_pad, col0);
        else if (ncols == 2) hipLaunchKernelGGL(k_syndrome_small32<2>, grid, dim3(512), 0, st, h->dA32, h->n, h->m, P, ld, ncols, h->q, rows_per_wg, h->dPart, h->n_pad, col0);
        else hipLaunchKernelGGL(k_syndrome_small32<4>, grid, dim3(512), 0, st, h->dA32, h->n, h->m, P, ld, ncols, h->q, rows_per_wg, h->dPart, h->n_pad, col0);
      } else
      if (ncols == 1) hipLaunchKernelGGL(k_syndrome_small<1>, grid, dim3(512), 0, st, h->dA, h->n, h->m, P, ld, ncols, h->q, rows_per_wg, h->dPart, h->n_pad, col0);
      else if (ncols == 2) hipLaunchKernelGGL(k_syndrome_small<2>, grid, dim3(512), 0, st, h->dA, h->n, h->m, P, ld, ncols, h->q, rows_per_wg, h->dPart, h->n_pad, col0);
      else hipLaunchKernelGGL(k_syndrome_small<4>, grid, dim3(512), 0, st, h->dA, h->n, h->m, P, ld, ncols, h->q, rows_per_wg, h->dPart, h->n_pad, col0);
      hipLaunchKernelGGL((k_zq_combine_wave<false>), dim3((unsigned)((h->n * ncols + 3) / 4)), dim3(256), 0, st, mode, h->dPart, splits, h->n, h->n_pad, ld, ncols, h->q, U, out, ldo, col0);
      return;
    }
  }
  size_t cw = round_up(ncols, 64);                                    // the product works on 64-column tiles
  if (col0 + cw > ld) cw = ld - col0;
  hipLaunchKernelGGL(k_split_P, dim3(grid_for(h->K_pad / 16 * cw, 256, 256 * 64)), dim3(256), 0, st, P, h->m, ld, h->K_pad / 16, P8, h->dFail, col0, cw);
  {  // PSF_ZQ_PLANES3=1: the third digit plane of p is multiplied whether or not it holds anything (comparison arm; same residues)
    static const bool force3 = [] { const char* e = psf_exp_env("PSF_ZQ_PLANES3"); return e && std::atoi(e) != 0; }();
    if (force3) hipMemsetAsync(h->dFail + 2, 1, sizeof(int), st);
  }
  const int nks = (int)(h->K_pad / 64);
  const int splits = zq_plan(h, ncols, h->zq_split_cap), zq_ks = (nks + splits - 1) / splits;
  dim3 grid((unsigned)((ncols + 63) / 64), (unsigned)(h->n_pad / 64), (unsigned)splits);
  int fold128 = zq_ks <= 32 ? 1 : 0;                                  // short splits (few preimages): one 128-bit fold per output; long ones: the per-class fold
  if (const char* e = psf_exp_env("PSF_ZQ_FOLD128")) fold128 = std::atoi(e);
  // a power-of-two modulus covered by the digits of A: the classes from NA on vanish mod q (pw[NA] = 0) and their digit pairs are skipped (PSF_ZQ_POW2=0: multiplied anyway)
  bool pow2 = (h->q & (h->q - 1)) == 0 && h->NA <= 8 && h->zc.pw[h->NA] == 0;
  if (const char* e = psf_exp_env("PSF_ZQ_POW2")) pow2 = pow2 && std::atoi(e) != 0;
#define ZQL(NA_, F_, P_) hipLaunchKernelGGL((k_zq_mfma<NA_, F_, P_>), grid, dim3(256), 2 * (NA_ + 3) * 4096, st, h->dA8, h->n_pad, h->K_pad, P8, ld, zq_ks, h->zc, (int)h->wide, h->dPart, col0, h->dFail)
#define ZQM(NA_)                                                                                                        \
  case NA_:                                                                                                             \
    if (fold128) { if (pow2) ZQL(NA_, true, true); else ZQL(NA_, true, false); }                                         \
    else { if (pow2) ZQL(NA_, false, true); else ZQL(NA_, false, false); }                                               \
    break;
  switch (h->NA) { ZQM(1) ZQM(2) ZQM(3) ZQM(4) ZQM(5) ZQM(6) ZQM(7) ZQM(8) default: break; }
  if (splits >= 16 && h->n * ncols <= 16384)      // a single call: few outputs, many splits -- one wave per output
    hipLaunchKernelGGL((k_zq_combine_wave<false>), dim3((unsigned)((h->n * ncols + 3) / 4)), dim3(256), 0, st, mode, h->dPart, splits, h->n, h->n_pad, ld, ncols,
                       h->q, U, out, ldo, col0);
  else
    hipLaunchKernelGGL(k_zq_combine, dim3(grid_for(h->n * ncols, 256, 256 * 32)), dim3(256), 0, st, mode, h->dPart, splits, h->n, h->n_pad, ld, ncols,
                       h->q, U, out, ldo, col0);
#undef ZQM
#undef ZQL
}

// A_bar <- U(Z_q^{n x m_bar}), R <- PlusMinusOneZero, A = [A_bar | G - A_bar R] (gen_trapdoor, gadget_classical.rs:56-68, tag = I)
// side: nullptr, or a non-blocking stream on which A = [A_bar | G - A_bar R] and its digit planes are computed while the caller goes on with R alone (the
// factorisation of Sigma_2 needs R, not A: psfp_trap_gen); the caller synchronises `side` before anything reads A
static psf_status gen_A_R(psfp_handle* h, uint64_t seed, hipStream_t side = nullptr) {
  if (gadget_too_short(h->prm.gp.base, h->k, h->q)) return PSF_ERR_MODULUS;
  // mp_perturbation.rs:222 / gpv.rs:84 ; gadget_classical.rs:62-64
  hipLaunchKernelGGL(k_sample_abar, dim3(grid_for(h->n * h->mb)), dim3(256), 0, 0, seed, h->n, h->mb, h->m, h->q, h->dA);
  h->r8_valid = false; h->small_state = 0;
  hipLaunchKernelGGL(k_sample_R, dim3(grid_for(h->mb * h->ldr)), dim3(256), 0, 0, seed, h->mb, h->w, h->ldr, h->dR);
  if (side) {
    HIP_TRY(hipEventRecord(h->evIn, nullptr));
    HIP_TRY(hipStreamWaitEvent(side, h->evIn, 0));
  }
  // gadget_classical.rs:66
  launch_zq_trapdoor(h, nullptr, side);
  HIP_TRY(hipGetLastError());
  split_A(h, side);
  return PSF_OK;
}

// gen_trapdoor (gadget_classical.rs:56-68) as a free function: caller-supplied A_bar and tag H; R <- PlusMinusOneZero from `seed` (the stream
// psfp_trap_gen uses) or, with R_in, the caller's own draw from whatever TrapdoorDistribution it uses (`params.distribution.sample(...)`,
// gadget_classical.rs:62-64 -- a trait object in the reference, trapdoor_distribution.rs:21-48); A = [A_bar | H G - A_bar R] on the device
static psf_status gen_trapdoor_core(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const uint64_t* tag, uint64_t seed, const int64_t* R_in,
                                    uint64_t* A, int8_t* R_out) {
  if (!gp || !a_bar || !A) return PSF_ERR_PARAM;
  psfp_params prm;
  prm.gp = *gp; prm.r = 1.0; prm.s = 1.0; prm.device = device; prm.flags = PSFP_FLAG_NO_PERTURB;
  psfp_handle* h = nullptr;
  psf_status rc = psfp_create(&prm, &h);
  if (rc != PSF_OK) return rc;
  if (gadget_too_short(gp->base, h->k, h->q)) { psfp_destroy(h); return PSF_ERR_MODULUS; }
  uint64_t* dtag = nullptr;
  auto fail = [&](psf_status st) { hipFree(dtag); psfp_destroy(h); return st; };
  if (hipMemcpy2D(h->dA, h->m * sizeof(uint64_t), a_bar, h->mb * sizeof(uint64_t), h->mb * sizeof(uint64_t), h->n, hipMemcpyHostToDevice) != hipSuccess) return fail(PSF_ERR_HIP);
  if (tag) {
    if (hipMalloc(&dtag, h->n * h->n * sizeof(uint64_t)) != hipSuccess) return fail(PSF_ERR_HIP);
    if (hipMemcpy(dtag, tag, h->n * h->n * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return fail(PSF_ERR_HIP);
  }
  if (R_in) {
    // the trapdoor lives in int8 on the device (operand of the int8 matrix cores in e = p + [R; I] z and of the dot4 assembly of Sigma_2)
    std::vector<int8_t> r8(h->mb * h->w);
    for (size_t i = 0; i < r8.size(); ++i) {
      if (R_in[i] > 127 || R_in[i] < -127) return fail(PSF_ERR_UNSUPPORTED);
      r8[i] = (int8_t)R_in[i];
    }
    h->r8_valid = false; h->small_state = 0;
    if (hipMemset(h->dR, 0, h->mb_pad * h->ldr) != hipSuccess) return fail(PSF_ERR_HIP);
    if (hipMemcpy2D(h->dR, h->ldr, r8.data(), h->w, h->w, h->mb, hipMemcpyHostToDevice) != hipSuccess) return fail(PSF_ERR_HIP);
  } else {
    h->r8_valid = false; h->small_state = 0;
    hipLaunchKernelGGL(k_sample_R, dim3(grid_for(h->mb * h->ldr)), dim3(256), 0, 0, seed, h->mb, h->w, h->ldr, h->dR);    // gadget_classical.rs:62-64
  }
  launch_zq_trapdoor(h, dtag);                                                                                            // :66
  if (hipGetLastError() != hipSuccess) return fail(PSF_ERR_HIP);
  if (hipMemcpy(A, h->dA, h->n * h->m * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return fail(PSF_ERR_HIP);
  if (R_out && hipMemcpy2D(R_out, h->w, h->dR, h->ldr, h->w, h->mb, hipMemcpyDeviceToHost) != hipSuccess) return fail(PSF_ERR_HIP);
  return fail(PSF_OK);
}
psf_status psf_gen_trapdoor(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const uint64_t* tag, uint64_t seed, uint64_t* A, int8_t* R) {
  if (!R) return PSF_ERR_PARAM;
  return gen_trapdoor_core(device, gp, a_bar, tag, seed, nullptr, A, R);
}
psf_status psf_gen_trapdoor_with_r(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const uint64_t* tag, const int64_t* R, uint64_t* A) {
  if (!R) return PSF_ERR_PARAM;
  return gen_trapdoor_core(device, gp, a_bar, tag, 0, R, A, nullptr);
}

// contiguous shares of `total` rows for `world` workers: the first total % world workers get one row more
psf_status psf_shard_range(size_t total, int world, int rank, size_t* first, size_t* count) {
  if (world < 1 || rank < 0 || rank >= world || !first || !count) return PSF_ERR_PARAM;
  const size_t base = total / (size_t)world, rem = total % (size_t)world;
  *count = base + ((size_t)rank < rem ? 1 : 0);
  *first = (size_t)rank * base + ((size_t)rank < rem ? (size_t)rank : rem);
  return PSF_OK;
}

psf_status psfp_trap_gen(psfp_handle* h, uint64_t seed) {
  if (!h) return PSF_ERR_PARAM;
  if (h->prm.flags & PSFP_FLAG_NO_PERTURB) return PSF_ERR_UNSUPPORTED;
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  KeygenClock kc("psfp");
  // A = [A_bar | G - A_bar R] (13 ms at C3) on the handle's low-priority stream beside the factorisation of Sigma_2, which needs R alone
  const psf_status rcg = gen_A_R(h, seed, h->aux);
  if (rcg != PSF_OK) return rcg;
  kc.mark("A_bar, R (A on the side stream)");
  h->has_pub = h->has_R = true;
  const psf_status rc = build_sqrt_sigma2(h, h->prm.s);            // mp_perturbation.rs:227-231
  HIP_TRY(hipStreamSynchronize(h->aux));
  if (rc != PSF_OK) { h->has_key = false; return rc; }
  kc.mark("sqrt(Sigma_2)");
  h->has_key = true;
  hp_prewarm(h);
  kc.mark("prewarm (returns at once)");
  return PSF_OK;
}

psf_status psfp_compute_sqrt_sigma_2(psfp_handle* h, double s_cov) {
  if (!h || !(s_cov > 0.0)) return PSF_ERR_PARAM;
  if (!h->has_R || (h->prm.flags & PSFP_FLAG_NO_PERTURB)) return PSF_ERR_NO_KEY;
  // The structured factor's constants g, h depend on the covariance parameter, and an exported key carries only L_1: psfp_load_key rebuilds them from
  // the handle's s.  A factor for another s_cov would therefore be paired with the wrong constants after an export / load round trip, silently; refused.
  if (h->structured && s_cov != h->prm.s) return PSF_ERR_UNSUPPORTED;
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  const psf_status rc = build_sqrt_sigma2(h, s_cov);
  h->has_key = rc == PSF_OK;
  return rc;
}

// compute_sqrt_sigma_2 with a general covariance (mp_perturbation.rs:111: `mat_sigma: &MatQ` is any symmetric matrix, used as a full matrix at :125-126)
psf_status psfp_compute_sqrt_sigma_2_dense(psfp_handle* h, const double* sigma_lower_packed) {
  if (!h || !sigma_lower_packed) return PSF_ERR_PARAM;
  if (!h->has_R || (h->prm.flags & PSFP_FLAG_NO_PERTURB)) return PSF_ERR_NO_KEY;
  if (h->structured) return PSF_ERR_UNSUPPORTED;           // the structured factor exists for Sigma = s^2 I only
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  double* dsg = nullptr;
  const size_t np = h->m * (h->m + 1) / 2;
  HIP_TRY(hipMalloc(&dsg, np * sizeof(double)));
  if (hipMemcpy(dsg, sigma_lower_packed, np * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { hipFree(dsg); return PSF_ERR_HIP; }
  const psf_status rc = build_sqrt_sigma2(h, 0.0, dsg);
  hipFree(dsg);
  h->has_key = rc == PSF_OK;
  return rc;
}

psf_status psfp_load_key(psfp_handle* h, const uint64_t* A, const int8_t* R, const double* Lp) {
  if (!h || !A || (!R && Lp)) return PSF_ERR_PARAM;
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  h->has_key = h->has_R = h->has_pub = false;
  HIP_TRY(hipMemcpy(h->dA, A, h->n * h->m * sizeof(uint64_t), hipMemcpyHostToDevice));
  split_A(h);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  h->has_pub = true;
  if (!R) return PSF_OK;                                    // public key only: the verifier's handle (f_a, check_domain, samp_d)
  h->r8_valid = false; h->small_state = 0;
  HIP_TRY(hipMemset(h->dR, 0, h->mb_pad * h->ldr));
  HIP_TRY(hipMemcpy2D(h->dR, h->ldr, R, h->w, h->w, h->mb, hipMemcpyHostToDevice));
  h->has_R = true;
  if (h->prm.flags & PSFP_FLAG_NO_PERTURB) return PSF_OK;
  if (!Lp) {                                                // trapdoor without its factor: recompute it as trap_gen does (mp_perturbation.rs:227-231)
    const psf_status rc = build_sqrt_sigma2(h, h->prm.s);
    h->has_key = rc == PSF_OK;
    return rc;
  }
  double* dp = nullptr;
  const size_t np = h->mL * (h->mL + 1) / 2;
  HIP_TRY(hipMalloc(&dp, np * sizeof(double)));
  HIP_TRY(hipMemcpy(dp, Lp, np * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_repack_L<true>, dim3(grid_for(tr_total_chunks(h->nbiL) * TR_CHUNK)), dim3(256), 0, 0, dp, (size_t)0, h->mL, h->dLt, h->nbiL);
  if (h->structured) {      // the constants of the structured factor follow from (r, s): the same expressions as build_sqrt_sigma2
    const double nf_r2 = (1.0 / 6.283185307179586476925) * (h->prm.r * h->prm.r), kappa = (double)(h->prm.gp.base * h->prm.gp.base + 1);
    const double alpha = h->prm.s * h->prm.s - 1.0, beta = alpha - kappa;
    if (!(beta > 0.0)) { hipFree(dp); return PSF_ERR_NOT_PD; }
    h->g_const = (std::sqrt(nf_r2) * kappa) / std::sqrt(beta);
    h->h_const = std::sqrt(nf_r2 * beta);
    ensure_R8(h, nullptr);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  hipFree(dp);
  h->has_key = true;
  hp_prewarm(h);
  return PSF_OK;
}

// (A, R) without a factor: what PSFPerturbation::compute_sqrt_sigma_2 (mp_perturbation.rs:111) needs installed -- it is a pure function of mat_r and
// mat_sigma in the reference, so A may be NULL (the handle's public matrix, if any, stays).  No Sigma_2 is assembled and nothing is factored here
// (psfp_load_key(A, R, NULL) runs a full Cholesky with the handle's s); samp_p answers PSF_ERR_NO_KEY until psfp_compute_sqrt_sigma_2(_dense) has run.
psf_status psfp_load_trapdoor(psfp_handle* h, const uint64_t* A, const int8_t* R) {
  if (!h || !R) return PSF_ERR_PARAM;
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  h->has_key = h->has_R = false;
  if (A) {
    h->has_pub = false;
    HIP_TRY(hipMemcpy(h->dA, A, h->n * h->m * sizeof(uint64_t), hipMemcpyHostToDevice));
    split_A(h);
    HIP_TRY(hipGetLastError());
    h->has_pub = true;
  }
  h->r8_valid = false; h->small_state = 0;
  HIP_TRY(hipMemset(h->dR, 0, h->mb_pad * h->ldr));
  HIP_TRY(hipMemcpy2D(h->dR, h->ldr, R, h->w, h->w, h->mb, hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  h->has_R = true;
  return PSF_OK;
}

psf_status psfp_export_key(const psfp_handle* h, uint64_t* A, int8_t* R, double* Lp) {
  if (!h) return PSF_ERR_PARAM;
  if (!h->has_pub || ((R || Lp) && !h->has_R) || (Lp && !h->has_key)) return PSF_ERR_NO_KEY;
  HIP_TRY(hipSetDevice(h->prm.device));
  if (A) HIP_TRY(hipMemcpy(A, h->dA, h->n * h->m * sizeof(uint64_t), hipMemcpyDeviceToHost));
  if (R) HIP_TRY(hipMemcpy2D(R, h->w, h->dR, h->ldr, h->w, h->mb, hipMemcpyDeviceToHost));
  if (Lp) {
    double* dp = nullptr;
    const size_t np = h->mL * (h->mL + 1) / 2;
    HIP_TRY(hipMalloc(&dp, np * sizeof(double)));
    hipLaunchKernelGGL(k_unpack_L, dim3(grid_for(np)), dim3(256), 0, 0, h->dLt, (size_t)0, np, dp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(Lp, dp, np * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(dp);
  }
  return PSF_OK;
}

// rows [row0, row0 + nrows) of the factor, packed (row i: i + 1 entries): keys of tens of GB are read back in pieces
psf_status psfp_export_sqrt_sigma2_rows(const psfp_handle* h, size_t row0, size_t nrows, double* out) {
  if (!h || (nrows && !out) || row0 + nrows > h->mL || (h->prm.flags & PSFP_FLAG_NO_PERTURB)) return PSF_ERR_PARAM;
  if (!h->has_key) return PSF_ERR_NO_KEY;
  if (nrows == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  const size_t first = row0 * (row0 + 1) / 2, total = (row0 + nrows) * (row0 + nrows + 1) / 2 - first;
  double* dp = nullptr;
  HIP_TRY(hipMalloc(&dp, total * sizeof(double)));
  hipLaunchKernelGGL(k_unpack_L, dim3(grid_for(total)), dim3(256), 0, 0, h->dLt, first, total, dp);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dp, total * sizeof(double), hipMemcpyDeviceToHost));
  hipFree(dp);
  return PSF_OK;
}

psf_status psfp_export_gadget_basis(const psfp_handle* h, int64_t* Sk, double* gso) {
  if (!h) return PSF_ERR_PARAM;
  if (Sk) std::memcpy(Sk, h->hSk.data(), h->hSk.size() * sizeof(int64_t));
  if (gso) std::memcpy(gso, h->hGso.data(), h->hGso.size() * sizeof(double));
  return PSF_OK;
}

// ---- the hot path -------------------------------------------------------------------------------------------
// One samp_p pass over B rows (mp_perturbation.rs:304-336).  By default everything runs in order on the caller's stream.
// With PSF_PIPELINE=1 consecutive calls alternate between two sets of intermediates: normals + FP64 product of call i+1 on
// stream s1, sampling stages of call i on stream aux, the caller's stream joins at the end (measured zero-sum on MI355X,
// profiles/r01_notes.md, hence off by default; covered by tests/test_gpu_pipeline_mode.py).
static psf_status run_samp_p(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_e, hipStream_t st) {
  const size_t ld = h->ld, m = h->m;
  const size_t nbj = h->nbj;
  const bool pipe = h->pipeline;
  hipStream_t user_st = st, s2 = st;
  int cur = 0;
  if (pipe) {
    cur = (int)(h->ncall++ & 1);
    select_set(h, cur);
    HIP_TRY(hipEventRecord(h->evIn, user_st));            // u is ready on the caller's stream from here on
    HIP_TRY(hipStreamWaitEvent(h->s1, h->evP[cur], 0));   // the previous user of this buffer set has finished
    st = h->s1;
    s2 = h->aux;
  }
  if (!h->keep_fail) hipMemsetAsync(h->dFail, 0, 4 * sizeof(int), st);      // [0] sampler failure, [1] some |z| > 127, [2] some |p| >= 2^15 (third digit plane of the syndrome product in use)
  psf_status gate_rc = PSF_OK;
  auto u_gate = [&]() { if (h->before_u) { auto f = std::move(h->before_u); h->before_u = nullptr; gate_rc = f(); } };
  {  // small parameter sets, few preimages (the reference's own benchmarks: n = 8, one call; benches/psf.rs:51-66): the whole call in ONE launch, one
     // workgroup per preimage (k_samp_p_small).  PSF_FUSED_MAX = largest batch it serves (0: never).  Stage exports need the intermediates: not here.
    size_t fused_max = 64;
    if (const char* e = psf_exp_env("PSF_FUSED_MAX")) fused_max = (size_t)std::atol(e);
    if (!pipe && !h->structured && !h->no_slice && h->gadget_queue && m <= (size_t)FS_MAX_M && h->n <= 64 && B <= fused_max) {
      u_gate();
      if (gate_rc != PSF_OK) return gate_rc;
      ScopedTimer t(h, st, "k_samp_p_small");
      GadgetTablesQ tq{h->dSk, h->dGso, h->dNorm2, h->dSz, h->dRng};
      hipLaunchKernelGGL(k_samp_p_small, dim3((unsigned)B), dim3(FS_THREADS), 0, st, seed, first_index, (uint32_t)h->n, (uint32_t)h->k, (uint32_t)h->mb, h->q, h->two64,
                         h->prm.gp.base, h->dLt, h->dA, h->dR, h->ldr, h->szR, tq, d_u, d_e, h->dFail);
      HIP_TRY(hipGetLastError());
      h->last_stream = user_st;
      if (h->multi_t0 && h->multi_launched_ms < 0.0)
        h->multi_launched_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - *h->multi_t0).count();
      return PSF_OK;
    }
  }
  // Small batches (one samp_p call of the reference is ONE preimage, psf.rs:48-80): the streaming product, bound by reading the factor once, fed by
  // the compact normals stream.  PSF_TRMM_STREAM_MAX = largest batch it serves (0 switches it off); PSF_TRMM_STREAM_SHAPE = "RT,NB" forces a tile
  // shape, PSF_COMPACT_D=0 the chunk-stream layout of the normals (experiments; same bits).
  // measured at C3: the streaming forms win up to 1472 preimages and again at 1537 ... 1728 (round 6, tools/tail_ab.py k_trmm_f64: k_trmm_f64_big costs 22.2-22.5 ms for anything
  // between 1025 and 1536 preimages and 27.2 up to 2048, the 64 x 64 tiles ~0.97 ms per round of 256 workgroups: 16.5 ms at 1152, 18.6 at 1280, 21.8 at 1472, 23.8 at 1600, 26.9 at 1792)
  size_t stream_max = 1472;
  bool stream = B <= stream_max || (B >= 1537 && B <= 1728) || (h->mL < 16384 && B <= 2048);      // (a small factor: k_trmm_f64_big is a fixed 0.23 ms at m = 932 whatever the batch, the
                                                                                                  // tiles 0.04 ms at 1100 ... 2048 preimages: no reason to leave them before the stream's limit)
  if (const char* e = psf_exp_env("PSF_TRMM_STREAM_MAX")) {
    stream_max = (size_t)std::atol(e);
    if (stream_max > 2048) stream_max = 2048;      // 128 column fragments: beyond, the over-read of the compact normals stream would leave TS_SLACK_DOUBLES
    stream = B <= stream_max;
  }
  // tile shape per wave (RT 16-row tiles x NB fragments of 16 preimages) and workgroup form, from tools/probe_stream.hip at the C3 shape
  // (profiles/r04_probe_stream.log): <= 16 preimages the launch is bound by reading the factor, beyond that by the longest MFMA chain
  int RT = 2, NB = B <= 16 ? 1 : B <= 64 ? 2 : 4;
  // more than wg_min preimages: 64 x 64 tiles whose four waves share the operands through LDS (k_trmm_stream_wg); PSF_STREAM_WG=0 keeps the one-wave tasks,
  // PSF_STREAM_WG=<B> moves the threshold (experiments build; same bits)
  size_t wg_min = 33;
  if (const char* e = psf_exp_env("PSF_STREAM_WG")) wg_min = std::atol(e) > 0 ? (size_t)std::atol(e) : (size_t)-1;
  size_t wg_max = 64;            // measured at C3 (tools/stream_wg_ab.py): 0.91-0.96 against 1.26 ms at 33 ... 64 preimages; 241 workgroups on 256 CUs, each as long as its pair of
                                 // chains: 0.87 ms would be the matrix pipe's time on 241 CUs.  Beyond 64 preimages the forms tried (column groups of 64 with one or two
                                 // workgroups per CU, column groups of 128 on halves of eight waves) end within 5 % of the one-wave tasks: PSF_STREAM_WG_MAX (<= 1024)
  if (const char* e = psf_exp_env("PSF_STREAM_WG_MAX")) wg_max = std::min<size_t>((size_t)std::atol(e), 1024);      // (beyond 64: the experiments build's halves of eight waves)
  // Beyond 128 preimages with an ODD number of column groups of 64 (129-192, 257-320, ... 897-960): the same tiles, the column groups of a tile group on one XCD -- the
  // one-wave tasks take as long as for the next even count there (192 preimages cost what 256 do: their eight-wave workgroups hold 4 + 4 tasks, column groups of a
  // tile group side by side), the tiles' workgroups are all of one length: 2.8-2.9 against 3.7-3.9 ms at 129-192, 4.8 / 5.9 at 320, 6.6 / 7.9 at 448, 8.6 / 9.9 at 576,
  // 14.5 / 15.9 at 960; with an even count the two forms tie (1.96 / 1.94 at 128, 7.70 / 7.69 at 512, 15.5 / 15.0 at 1024).  PSF_STREAM_WG192=0 keeps the one-wave
  // tasks, "lo:hi" forces the tiles for every batch size in the range (experiments build; same bits)
  // a small factor (m < 16 384: few tile groups, the launch lasts one chain): the tiles' waves hold 2 x 2 MFMA tiles against the one-wave tasks' 2 x 4 -- half the chain
  // (m = 932: 0.035 against 0.061 ms at 97 ... 1024 preimages, whatever the parity): the tiles from 65 preimages on
  const bool small_factor = h->mL < 16384;
  bool wg192 = stream && ((B > 128 && B <= 960 && (((B + 63) / 64) & 1) != 0) || B > 1088 || (small_factor && B > 64)) && wg_max <= 64 && !psf_exp_env("PSF_TRMM_STREAM_SHAPE");      // (1025-1088: 15.9 against 16.5)
  if (const char* e = psf_exp_env("PSF_STREAM_WG192")) {
    long lo = 0, hi = 0;
    wg192 = std::sscanf(e, "%ld:%ld", &lo, &hi) == 2 && lo >= 65 && hi <= 2048 && stream && B >= (size_t)lo && B <= (size_t)hi && wg_max <= 64 && !psf_exp_env("PSF_TRMM_STREAM_SHAPE");
  }
  const bool wg = stream && ((B >= wg_min && B <= wg_max) || wg192) && !psf_exp_env("PSF_TRMM_STREAM_SHAPE");
  // 17 ... 32 preimages: 64 x 32 tiles of the same ring (k_trmm_stream_wg32); PSF_STREAM_WG32=0 keeps the one-wave tasks (experiments build; same bits)
  const bool wg32 = stream && !wg && B >= 17 && B <= 32 && !psf_exp_env("PSF_TRMM_STREAM_SHAPE") && !(psf_exp_env("PSF_STREAM_WG32") && std::atoi(psf_exp_env("PSF_STREAM_WG32")) == 0);
  // 65 ... 96 preimages: the first 64 on the 64 x 64 tiles, the rest on the 64 x 32 tiles, two launches over one normals stream of six fragments (0.94 + 0.65 ms against 1.80 for
  // the one-wave tasks, which pay for 128 columns); PSF_STREAM_WG96=0 keeps those (experiments build; same bits)
  const bool wg96 = stream && !wg && !small_factor && B >= 65 && B <= 96 && !psf_exp_env("PSF_TRMM_STREAM_SHAPE") && !(psf_exp_env("PSF_STREAM_WG96") && std::atoi(psf_exp_env("PSF_STREAM_WG96")) == 0);
  if (wg) { RT = 2; NB = (B <= 64 || wg192) ? 4 : 8; }      // column groups of 64 (halves of four waves) or 128 preimages (halves of eight waves)
  if (wg32 || wg96) { RT = 2; NB = 2; }          // (wg96: three column groups of 32 = the six fragments of the stream)
  if (const char* e = psf_exp_env("PSF_TRMM_STREAM_SHAPE")) std::sscanf(e, "%d,%d", &RT, &NB);
  if (!(NB == 1 || NB == 2 || NB == 4 || NB == 8)) NB = 1;
  const int ncg = (int)((B + 16 * (size_t)NB - 1) / (16 * (size_t)NB));
  bool compact = stream && !h->structured;
  if (const char* e = psf_exp_env("PSF_COMPACT_D")) compact = compact && std::atoi(e) != 0;
  // <= 16 preimages in one fragment: the dense stream of bc = 1, 2, 4, 8, 16 preimages (PSF_COMPACT_D=1 keeps the fragment stream)
  int bc = 0;
  if (compact && NB == 1 && ncg == 1 && !(psf_exp_env("PSF_COMPACT_D") && std::atoi(psf_exp_env("PSF_COMPACT_D")) == 1)) { bc = 1; while ((size_t)bc < B) bc <<= 1; }
  const uint32_t ncf = bc ? 0x100u + (uint32_t)bc : compact ? (uint32_t)(ncg * NB) : 0u;      // layout code of the normals stream (k_normals_wave)
  h->normals_ncf = ncf;
  {  // mp_perturbation.rs:315 -- d <- N(0,1)^m
    ScopedTimer t(h, st, "k_normals");
    const size_t npos = bc ? h->nkb * 4 * 4 * (size_t)bc : ncf ? h->nkb * 4 * (size_t)ncf * 64 : nbj * h->nkb * TR_CHUNK;
    uint32_t nseg = nr_segment(npos);
    if (const char* e = psf_exp_env("PSF_NR_SEG")) { const long v = std::atol(e); if (v >= 64 && v <= NR_SEG && v % 64 == 0) nseg = (uint32_t)v; }      // positions per wave (experiments)
    const size_t nwaves = (npos + nseg - 1) / nseg;
    const NormalsFixed fx = h->structured ? NormalsFixed{h->mb, h->dD8, h->ldr * ld, ld, h->dX, h->h_const} : NormalsFixed{0, nullptr, 0, 0, nullptr, 0.0};
    hipLaunchKernelGGL(k_normals_wave, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, seed, first_index, m, B, h->nkb, nbj, h->dDt, h->dFail, fx, ncf, nseg);
  }
  // One or two preimages: rounding and syndrome in ONE launch behind the product (k_round_syndrome_small, psf_stream_kernels.hpp) once the transposed compact copy
  // of A is there (built beside the first small calls after a key change, as the other compact copies)
  bool fused_tail = false;
  int fused_ntask = 0, fused_rt = 2;      // two 16-row tiles per wave: 27 + 10 us against 34 + 14 with one (tools/fused_tail_ab.sh)
  if (stream && bc && B <= 2 && RT == 2 && NB == 1 && !h->structured && !pipe && h->szR.sh == 16 && !(h->prm.flags & PSFP_FLAG_NO_PERTURB)) {
    ensure_small_copies(h, st);
    const char* fe = psf_exp_env("PSF_FUSED_TAIL");
    if (h->dA32T && (h->small_state == 2 || h->small_state == 3) && !(fe && std::atoi(fe) == 0)) {
      if (const char* e = psf_exp_env("PSF_FUSED_RT")) fused_rt = std::atoi(e) == 1 ? 1 : 2;
      fused_ntask = ((int)((h->mL + 15) / 16) + fused_rt - 1) / fused_rt;      // one wave per fused_rt 16-row tiles of x
      const size_t need = (size_t)fused_ntask * h->n * 2;
      if (need > h->partF_cap) {
        hipFree(h->dPartF); h->dPartF = nullptr; h->partF_cap = 0;
        if (hipMalloc(&h->dPartF, need * sizeof(uint64_t)) == hipSuccess) h->partF_cap = need; else (void)hipGetLastError();
      }
      fused_tail = h->dPartF != nullptr;
    }
  }
  {  // x = sqrt(Sigma_2) d   (structured: the m_bar x m_bar block L_1 d_1; rows from m_bar on already hold x_bot = h d_2)
    ScopedTimer t(h, st, "k_trmm_f64");
    // default: k_trmm_f64_big (one workgroup per CU, accumulators in AccVGPRs); PSF_TRMM_VARIANT=1: k_trmm_f64_reg (two 128 x 128 workgroups per CU,
    // operands streamed into registers), 0: k_trmm_f64 (LDS-staged, round 1).  Same bits from all three.
    const char* venv = psf_exp_env("PSF_TRMM_VARIANT");      // read per call: the tests compare the kernels inside one process
    const int variant = venv ? std::atoi(venv) : 2;
    const size_t row_hi = h->structured ? h->mb : h->M_pad;
    if (stream && wg96) {
      StreamGeom g;
      g.ntile = ((int)((h->mL + 15) / 16) + 3) / 4;
      g.ncg = 1;
      g.ntask = g.ntile;
      g.bc = 0;
      g.ncf = 6;
      const unsigned grid = (unsigned)((g.ntask + 1) / 2);
      g.cf_base = 0;
      if (compact) hipLaunchKernelGGL((k_trmm_stream_wg<TSW64_H, TSW64_NBUF, 1, 2>), dim3(grid), dim3(512), TSW_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
      else hipLaunchKernelGGL((k_trmm_stream_wg<TSW64_H, TSW64_NBUF, 0, 2>), dim3(grid), dim3(512), TSW_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
      g.cf_base = 4;
      if (compact) hipLaunchKernelGGL((k_trmm_stream_wg32<TSW_H, TSW_NBUF, 1>), dim3(grid), dim3(512), TSW32_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
      else hipLaunchKernelGGL((k_trmm_stream_wg32<TSW_H, TSW_NBUF, 0>), dim3(grid), dim3(512), TSW32_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
    }
    else if (stream && wg32) {
      StreamGeom g;
      g.ntile = ((int)((h->mL + 15) / 16) + 3) / 4;
      g.ncg = ncg;
      g.ntask = g.ntile * g.ncg;
      g.bc = 0;
      const unsigned grid = (unsigned)((g.ntask + 1) / 2);
      if (compact) hipLaunchKernelGGL((k_trmm_stream_wg32<TSW_H, TSW_NBUF, 1>), dim3(grid), dim3(512), TSW32_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
      else hipLaunchKernelGGL((k_trmm_stream_wg32<TSW_H, TSW_NBUF, 0>), dim3(grid), dim3(512), TSW32_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
    }
    else if (stream && wg) {
      StreamGeom g;
      g.ntile = ((int)((h->mL + 15) / 16) + 3) / 4;
      g.ncg = ncg;
      g.ntask = g.ntile * g.ncg;
      g.bc = 0;
      const unsigned nwg = (unsigned)((g.ntask + 1) / 2);
      if (NB == 4) {         // column groups of 64 preimages: one workgroup of 2 x 4 waves per CU, rounds of four k-steps; several groups: those of a tile group on one XCD
        const unsigned grid64 = ncg > 1 ? 8 * ((nwg + 7) / 8) : nwg;
        if (compact) hipLaunchKernelGGL((k_trmm_stream_wg<TSW64_H, TSW64_NBUF, 1, 2>), dim3(grid64), dim3(512), TSW_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
        else hipLaunchKernelGGL((k_trmm_stream_wg<TSW64_H, TSW64_NBUF, 0, 2>), dim3(grid64), dim3(512), TSW_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
      }
#ifdef PSF_EXPERIMENTS
      else {                 // column groups of 128 preimages: 2 x 8 waves, rounds of two k-steps (96 KiB); the column groups of a tile group on one XCD.  Measured against the
                             // one-wave tasks (tools/stream_wg_ab.py): 1.93 / 1.91 ms at 128, 3.55 / 3.72 at 256, 7.16 / 7.58 at 512, 14.25 / 14.33 at 1024 preimages -- not kept
        const unsigned grid = ncg > 1 ? 8 * ((nwg + 7) / 8) : nwg;
        if (compact) hipLaunchKernelGGL((k_trmm_stream_wg<2, TSW_NBUF, 1, 4>), dim3(grid), dim3(1024), TSW128_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
        else hipLaunchKernelGGL((k_trmm_stream_wg<2, TSW_NBUF, 0, 4>), dim3(grid), dim3(1024), TSW128_LDS, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
      }
#endif
    }
    else if (stream) {
      const int ntile16 = (int)((h->mL + 15) / 16);
      auto go = [&](auto kern, int rt, int half) {
        StreamGeom g;
        g.ntile = (ntile16 + rt - 1) / rt;
        g.ncg = ncg;
        g.ntask = g.ntile * g.ncg;
        g.bc = bc;
        hipLaunchKernelGGL(kern, dim3((unsigned)((g.ntask + 2 * half - 1) / (2 * half))), dim3(128 * half), 0, st, h->dLt, h->dDt, h->dX, g, h->nkb, ld, row_hi);
      };
#define TS_GO(rt, nb, pd, half) { if (compact) go(k_trmm_stream<rt, nb, pd, half, 1>, rt, half); else go(k_trmm_stream<rt, nb, pd, half, 0>, rt, half); }
      if (bc && RT == 2 && NB == 1) go(k_trmm_stream<2, 1, 12, 2, 2>, 2, 2);
      else if (bc && NB == 1) go(k_trmm_stream<1, 1, 8, 4, 2>, 1, 4);
      else if (RT == 2 && NB == 1) TS_GO(2, 1, 12, 2)
      else if (RT == 2 && NB == 2 && B <= 32) TS_GO(2, 2, 8, 2)
      else if (RT == 2 && NB == 2) TS_GO(2, 2, 16, 4)
      else if (RT == 2 && NB == 4) TS_GO(2, 4, 8, 4)
      else if (RT == 4 && NB == 2) TS_GO(4, 2, 10, 4)
      else if (RT == 1 && NB == 8) TS_GO(1, 8, 8, 4)
      else if (RT == 1 && NB == 4) TS_GO(1, 4, 8, 4)
      else if (RT == 1 && NB == 2) TS_GO(1, 2, 8, 4)
      else TS_GO(1, 1, 8, 4)
#undef TS_GO
    }
    else if (variant == 2 || !psf_experiments_build) {
      int GR = 8, GC = 4;                                     // super-tile of an XCD's 32 resident workgroups; PSF_TRMM_GR x PSF_TRMM_GC for experiments (product = 32)
      // below 4096 preimages, or with a number of 128-column blocks that is not a multiple of four: 16 x 2 -- the grid is padded to whole super-columns, and 8 x 4 pays for up
      // to three empty column blocks (round 6, tools/tail_ab.py: 34.9 -> 32.3 ms at 2176, 47.8 -> 44.1 at 3200, 27.2 -> 26.3 at 2048; 51.9 against 52.5 at 4096: 8 x 4 stays there)
      if (B < 4096 || nbj % 4 != 0) { GR = 16; GC = 2; }
      if (const char* e1 = psf_exp_env("PSF_TRMM_GR")) if (const char* e2 = psf_exp_env("PSF_TRMM_GC")) { GR = std::atoi(e1); GC = std::atoi(e2); }
      if (GR < 1 || GC < 1 || GR * GC != 32) { GR = 8; GC = 4; }
      hipLaunchKernelGGL(k_trmm_f64_big, dim3(tr_grid_size(((int)h->nbiL + 1) / 2, (int)nbj, GR, GC)), dim3(256), 0, st, h->dLt, h->dDt, h->dX, (int)h->nbiL, (int)nbj, h->nkb, ld, GR, GC, row_hi);
    }
#ifdef PSF_EXPERIMENTS
    else if (variant == 1)
      hipLaunchKernelGGL(k_trmm_f64_reg, dim3(tr_grid_size((int)h->nbiL, (int)nbj, 8, 8)), dim3(256), 0, st, h->dLt, h->dDt, h->dX, (int)h->nbiL, (int)nbj, h->nkb, ld, 8, 8, row_hi);
    else
      hipLaunchKernelGGL(k_trmm_f64, dim3(tr_grid_size((int)h->nbiL, (int)nbj, 8, 8)), dim3(256), 4 * TR_CHUNK * sizeof(double), st, h->dLt, h->dDt, h->dX, (int)h->nbiL, (int)nbj, h->nkb, ld, 8, 8, row_hi);
#endif
  }
  if (h->structured) {  // x_top -= g R d_2 (exact integer sum on the int8 matrix cores)
    ensure_R8(h, st);
    ScopedTimer t(h, st, "k_rd2_mfma");
    hipLaunchKernelGGL(k_rd2_mfma, dim3((unsigned)(ld / 64), (unsigned)(round_up(h->mb, 64) / 64)), dim3(256), 3 * (1 + kFixPlanes) * 4096, st, h->dR8, h->ldr, h->dD8, h->ldr * ld, ld,
                       h->mb, h->g_const, h->dX);
  }
  if (pipe) {
    HIP_TRY(hipEventRecord(h->evT[cur], st));
    HIP_TRY(hipStreamWaitEvent(s2, h->evT[cur], 0));
    HIP_TRY(hipStreamWaitEvent(s2, h->evIn, 0));
  }
  // The stages behind the product, for the columns [b0, b0 + Bh) of the batch on stream sx.  Column offsets: [coord][b] matrices move by b0 elements,
  // digit planes ([group][b][16]) by 16 b0 bytes, row-major API matrices by b0 rows; the Z_q product takes its window as (col0, ncols).
  auto tail = [&](hipStream_t sx, size_t b0, size_t Bh) {
    if (fused_tail) {   // p_i <- D_{Z,r,x_i} and every 16-row tile's share of A p, one launch (k_round_syndrome_small)
      ScopedTimer t(h, sx, "k_round+A p");
      StreamGeom g;
      g.ntile = fused_ntask; g.ncg = 1; g.ntask = fused_ntask; g.bc = bc;
      const StreamFuse fz{seed, first_index, m, h->szR, h->dP, ld, h->dA32T, h->n, h->q, h->dPartF, h->dFail};
      const size_t row_hi2 = h->structured ? h->mb : h->M_pad;
      if (fused_rt == 1) hipLaunchKernelGGL((k_round_syndrome_small<1>), dim3((unsigned)((g.ntask + 3) / 4)), dim3(256), 0, sx, h->dX, ld, row_hi2, g, fz);
      else hipLaunchKernelGGL((k_round_syndrome_small<2>), dim3((unsigned)((g.ntask + 3) / 4)), dim3(256), 0, sx, h->dX, ld, row_hi2, g, fz);
    }
    if (!fused_tail) {  // p_i <- D_{Z,r,x_i}
      ScopedTimer t(h, sx, "k_perturb_round");
      const char* renv = psf_exp_env("PSF_ROUND");                 // "wave": the round-2 kernel (comparison arm; same bits)
      if (h->szR.sh == 16 && !(renv && !std::strcmp(renv, "wave"))) {
        uint32_t seg = prl_segment(m * Bh);
        if (const char* e = psf_exp_env("PSF_PRL_SEG")) { const long v = std::atol(e); if (v >= 64 && v <= PRL_SEG && v % 64 == 0) seg = (uint32_t)v; }      // samples per wave (experiments)
        const size_t waves = (m * Bh + seg - 1) / seg;
        if (h->szF && !(renv && !std::strcmp(renv, "lean"))) {     // the table screen ("lean": the fp32 screen of rounds 3-4, comparison arm; same bits)
          // a segment that is one row of the [coordinate][preimage] matrix never wraps: the sample's position is its offset (no division per sample)
          uint32_t segt = seg;
          if (Bh % 64 == 0 && Bh >= 1024 && Bh <= (size_t)PRL_SEG && (size_t)seg > Bh) segt = (uint32_t)Bh;      // (short rows: every workgroup loads the table, 0.14 against 0.09 ms at 64 preimages)
          const size_t wavest = (m * Bh + segt - 1) / segt;
          if ((size_t)segt == Bh)
            hipLaunchKernelGGL(k_perturb_round_tab<true>, dim3((unsigned)((wavest + 3) / 4)), dim3(256), (size_t)h->szR.n_int * h->szF * sizeof(uint32_t), sx, seed, first_index + b0, m, Bh, ld,
                               h->dX + b0, h->szR, h->dP + b0, h->dFail, segt, SzTable{h->dSzTab, h->szF, h->szR.n_int});
          else
            hipLaunchKernelGGL(k_perturb_round_tab<false>, dim3((unsigned)((wavest + 3) / 4)), dim3(256), (size_t)h->szR.n_int * h->szF * sizeof(uint32_t), sx, seed, first_index + b0, m, Bh, ld,
                               h->dX + b0, h->szR, h->dP + b0, h->dFail, segt, SzTable{h->dSzTab, h->szF, h->szR.n_int});
        } else
        hipLaunchKernelGGL(k_perturb_round_lean, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, sx, seed, first_index + b0, m, Bh, ld, h->dX + b0, h->szR, h->dP + b0, h->dFail, seg);
      } else {
        const size_t waves = (m * Bh + PR_SEG - 1) / PR_SEG;
        hipLaunchKernelGGL(k_perturb_round_wave, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, sx, seed, first_index + b0, m, Bh, ld, h->dX + b0, h->szR, h->dP + b0, h->dFail);
      }
    }
    u_gate();                                                      // (host path: u reaches the device now)
    {  // mp_perturbation.rs:318 -- v = u - A p
      ScopedTimer t(h, sx, "k_zq_matmul(syndrome)");
      if (fused_tail)      // the tasks of the product left their shares of A p in dPartF: summed and taken from u, one wave per output
        hipLaunchKernelGGL((k_zq_combine_wave<true>), dim3((unsigned)((h->n * Bh + 3) / 4)), dim3(256), 0, sx, ZQ_SYNDROME, h->dPartF, fused_ntask, h->n, h->n, (size_t)bc, Bh, h->q, d_u, h->dV, ld, (size_t)0);
      else launch_zq_mfma(h, sx, ZQ_SYNDROME, h->dP, h->dP8, Bh, d_u, h->dV, ld, b0);
    }
    {  // mp_perturbation.rs:321-326 -- z <- D_{Lambda_v(G), r sqrt(b^2+1)}
      ScopedTimer t(h, sx, "k_gadget");
      const char* genv = psf_exp_env("PSF_GADGET_WAVE");            // max n B served by the one-wave-per-problem kernel (0: never)
      const size_t wave_max = genv ? (size_t)std::atol(genv) : 2048;    // measured at C3 (n = 512): 45 / 44 / 55 us at 1 / 2 / 4 preimages (a row per problem: 58); 91 against 58 at 8
#ifdef PSF_EXPERIMENTS
      const char* genv16 = psf_exp_env("PSF_GADGET_WAVE16");        // max n B served by the sixteen-lanes-per-problem kernel (0: never)
      const size_t wave16_max = genv16 ? (size_t)std::atol(genv16) : 49152;     // measured at C3: 0.33 vs 0.48 ms at 64 preimages, 0.65 vs 0.60 at 128
#endif
      const char* genvq = psf_exp_env("PSF_GADGET_QUAD");          // max n B served by the four-lanes-per-problem kernel (0: never)
      const size_t quad_max = genvq ? (size_t)std::atol(genvq) : 98304;      // measured at C3 (tools/gadget_mid_ab.py): 0.092 / 0.092 / 0.12 / 0.24 ms at 16 / 32 / 64 / 128 preimages against
                                                                                // 0.12 / 0.18 / 0.33 / 0.35; 0.39 against 0.33 (queue kernel) at 256
      const char* genvr = psf_exp_env("PSF_GADGET_ROW");           // max n B served by the sixteen-lanes-per-problem form of k_gadget_quad (0: never)
      const size_t row_max = genvr ? (size_t)std::atol(genvr) : 10240;      // measured at C3 (tools/tail_ab.py k_gadget): 0.058 / 0.071 / 0.088 ms at 8 / 16 / 24 preimages against 0.091 (one
                                                                              // wave per problem at 8, a quad per problem at 16 and 24); 0.107 against 0.091 at 32
      if (h->gadget_queue && h->n * Bh <= wave_max) {               // a single call / a handful of preimages: the chain of k draws is the launch time
        GadgetTablesQ tq{h->dSk, h->dGso, h->dNorm2, h->dSz, h->dRng};
        hipLaunchKernelGGL(k_gadget_wave, dim3((unsigned)((h->n * Bh + 3) / 4)), dim3(256), 0, sx, seed, first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q,
                           h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail);
      } else if (h->gadget_queue && h->n * Bh <= row_max && h->k <= 64) {       // a few thousand problems: a DPP row per problem, one round per draw
        GadgetTablesQ tq{h->dSk, h->dGso, h->dNorm2, h->dSz, h->dRng};
        if (h->k <= 32)
          hipLaunchKernelGGL((k_gadget_quad<2, 16>), dim3((unsigned)((h->n * Bh + 15) / 16)), dim3(256), 0, sx, seed, first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q,
                             h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail);
        else
          hipLaunchKernelGGL((k_gadget_quad<4, 16>), dim3((unsigned)((h->n * Bh + 15) / 16)), dim3(256), 0, sx, seed, first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q,
                             h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail);
      } else if (h->gadget_queue && h->n * Bh <= quad_max && h->k <= 64) {      // tens to a few hundred preimages: a quad per problem
        GadgetTablesQ tq{h->dSk, h->dGso, h->dNorm2, h->dSz, h->dRng};
        if (h->k <= 32)
          hipLaunchKernelGGL(k_gadget_quad<8>, dim3((unsigned)((h->n * Bh + 63) / 64)), dim3(256), 0, sx, seed, first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q,
                             h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail);
        else
          hipLaunchKernelGGL(k_gadget_quad<16>, dim3((unsigned)((h->n * Bh + 63) / 64)), dim3(256), 0, sx, seed, first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q,
                             h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail);
#ifdef PSF_EXPERIMENTS
      } else if (h->gadget_queue && h->n * Bh <= wave16_max) {      // up to a few hundred preimages: four problems per wave
        GadgetTablesQ tq{h->dSk, h->dGso, h->dNorm2, h->dSz, h->dRng};
        hipLaunchKernelGGL(k_gadget_wave16, dim3((unsigned)((h->n * Bh + 15) / 16)), dim3(256), 0, sx, seed, first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q,
                           h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail);
#endif
      } else if (h->gadget_queue) {
        GadgetTablesQ tq{h->dSk, h->dGso, h->dNorm2, h->dSz, h->dRng};
        int P = gq_problems_for((uint32_t)h->k, h->n * Bh);
        if (const char* e = psf_exp_env("PSF_GQ_P")) { const int v = std::atoi(e); if (v >= 1 && v <= 128 && (v & (v - 1)) == 0) P = v; }      // problems per wave (experiments)
        const size_t per_wg = (size_t)GQ_WAVES * P;
        if (P == 128)
          hipLaunchKernelGGL(k_gadget_queue<true>, dim3((unsigned)((h->n * Bh + per_wg - 1) / per_wg)), dim3(256), gadget_queue_lds_bytes(h->k, P), sx, seed,
                             first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q, h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail, P);
        else
          hipLaunchKernelGGL(k_gadget_queue<false>, dim3((unsigned)((h->n * Bh + per_wg - 1) / per_wg)), dim3(256), gadget_queue_lds_bytes(h->k, P), sx, seed,
                             first_index + b0, (uint32_t)h->n, (uint32_t)h->k, h->q, h->prm.gp.base, Bh, ld, h->dV + b0, tq, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail, P);
      } else {
        GadgetTables tb{h->dSk, h->dGso, h->dNorm2, h->dSz};
        hipLaunchKernelGGL(k_gadget, dim3((unsigned)((Bh + 255) / 256), (unsigned)h->n), dim3(256), gadget_lds_bytes(h->k), sx, seed, first_index + b0,
                           (uint32_t)h->n, (uint32_t)h->k, h->q, h->prm.gp.base, Bh, ld, h->dV + b0, tb, h->dZlo + 16 * b0, h->dZhi + 16 * b0, h->dFail);
      }
    }
    {  // mp_perturbation.rs:328-335 -- e = p + [R; I] z
      ScopedTimer t(h, sx, "k_recombine");
      // a handful of preimages: R streamed once by one wave per row (PSF_RECOMBINE_SMALL = largest batch it serves, 0: never)
      size_t small_max = 4;
      if (const char* e = psf_exp_env("PSF_RECOMBINE_SMALL")) small_max = (size_t)std::atol(e);
      if (small_max > 4) small_max = 4;
      const size_t small_lds = 32 * (h->ldr / 16) * Bh;
      if (Bh <= small_max && small_lds <= 150 * 1024) {
        const unsigned wgs = (unsigned)std::min<size_t>((h->mb + 7) / 8, small_lds > 64 * 1024 ? 256 : 512);
        ensure_small_copies(h, sx);
        if (h->small_state == 2) {
          if (Bh == 1) hipLaunchKernelGGL(k_recombine_small2<1>, dim3(wgs), dim3(512), small_lds, sx, h->dR2, h->ldr, h->mb, h->w, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m);
          else if (Bh == 2) hipLaunchKernelGGL(k_recombine_small2<2>, dim3(wgs), dim3(512), small_lds, sx, h->dR2, h->ldr, h->mb, h->w, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m);
          else hipLaunchKernelGGL(k_recombine_small2<4>, dim3(wgs), dim3(512), small_lds, sx, h->dR2, h->ldr, h->mb, h->w, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m);
          return;
        }
        if (Bh == 1) hipLaunchKernelGGL(k_recombine_small<1>, dim3(wgs), dim3(512), small_lds, sx, h->dR, h->ldr, h->mb, h->w, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m);
        else if (Bh == 2) hipLaunchKernelGGL(k_recombine_small<2>, dim3(wgs), dim3(512), small_lds, sx, h->dR, h->ldr, h->mb, h->w, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m);
        else hipLaunchKernelGGL(k_recombine_small<4>, dim3(wgs), dim3(512), small_lds, sx, h->dR, h->ldr, h->mb, h->w, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m);
        return;
      }
      // 5 ... 448 preimages: 64 x 64 tiles over all of K, operands through an LDS-DMA ring, no atomics (k_recombine_wg); PSF_RECOMBINE_STREAM=0: the tiled kernel below
      // (experiments build; same rows): 0.067 against 0.091 ms at 16, 0.081 against 0.155 at 64 preimages of C3 (tools/tail_ab.py)
      size_t rs_max = 448;      // column groups of 64 preimages beyond 64 (blockIdx.y; R comes from L2 / the Infinity Cache for all but the first): 0.157 -> 0.107 ms at 65, 0.266 -> 0.115 at 128,
                                // 0.247 -> 0.174 at 192, 0.238 -> 0.210 at 256, 0.353 -> 0.299 at 384; 0.248 -> 0.390 at 512 (the 256 x 256 tiles), 0.575 -> 0.729 at 1000 preimages
      if (const char* e = psf_exp_env("PSF_RECOMBINE_STREAM")) rs_max = (size_t)std::min<long>(std::atol(e), 1024);
      if (Bh <= rs_max && h->ldr % 128 == 0 && h->mb >= 64) {
        const int nbf = Bh > 64 ? 4 : (int)((Bh + 15) / 16), nk2 = (int)(h->ldr / 128);
        const unsigned ngy = (unsigned)((Bh + 63) / 64);
        hipLaunchKernelGGL(k_recombine_bottom, dim3((unsigned)((Bh + 63) / 64), (unsigned)((h->w + 63) / 64)), dim3(256), 0, sx, h->mb, h->w,
                           h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m, 0);
        const unsigned grid = (unsigned)((h->mb + 63) / 64);
#define RW_GO(nb, nw) hipLaunchKernelGGL((k_recombine_wg<nb, nw>), dim3(grid, ngy), dim3(64 * nw), RW_LDS, sx, h->dR, h->ldr, h->mb, nk2, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, \
                                         h->dFail, h->dP + b0, Bh, d_e + b0 * m, m)
        if (nbf == 1) RW_GO(1, 4); else if (nbf == 2) RW_GO(2, 8); else if (nbf == 3) RW_GO(3, 8); else RW_GO(4, 8);      // (four waves at 33 ... 64 preimages: 0.147 against 0.081 ms)
#undef RW_GO
        return;
      }
      // one digit plane (decided on the device by the gadget kernel): 256 x 256 tiles; otherwise, or for shapes the big tile does not fit, the 128 x 128 kernel
      // (beyond 448 preimages also for batches that are not multiples of 256: the last tile is ragged -- its loads of z past the batch stay inside the planes or their slack,
      // the stores are masked -- and still cheaper than the 128 x 128 kernel: 0.437 -> 0.29 ms at 704, 0.548 -> 0.30 at 832 preimages; PSF_RECOMBINE_RAGGED=0: multiples only)
      const bool ragged_ok = !(psf_exp_env("PSF_RECOMBINE_RAGGED") && std::atoi(psf_exp_env("PSF_RECOMBINE_RAGGED")) == 0);
      const bool big = (Bh % 256 == 0 || (Bh > rs_max && ragged_ok)) && b0 % 256 == 0 && h->mb >= 512 && (h->ldr / 64) % 2 == 0;
      if (big) {
        const unsigned nbx = (unsigned)((Bh + 255) / 256), nby = (unsigned)(h->mb_pad / 256);
        const unsigned nsup = ((nbx + 3) / 4) * ((nby + 7) / 8);                 // super-tiles of 4 x 8 tiles, dealt to the XCDs in rounds of eight
        ensure_R8(h, sx);
        hipLaunchKernelGGL(k_recombine_mfma_big, dim3(((nsup + 7) / 8) * 8 * 32), dim3(512), RCB_LDS, sx, rcb_packed() ? h->dR8 : h->dR, rcb_packed() ? 1 : 0, h->ldr, h->mb,
                           (int)(h->ldr / 128), h->dZlo + 16 * b0, ld, h->dFail, h->dP + b0, Bh, d_e + b0 * m, m, nbx, nby);
      }
      // few preimages: cut K over blockIdx.z (an even number of K steps each, at least 8) until ~2048 workgroups; the partial sums are added into a zeroed E
      const unsigned tiles = (unsigned)((Bh + 127) / 128) * (unsigned)((h->mb + 127) / 128);
      const int nks = (int)(h->ldr / 64);
      int rsplits = 1, kps = nks;
      if (!big && tiles < 1024 && nks >= 16) {
        rsplits = (int)(2048 / tiles);
        kps = (nks + rsplits - 1) / rsplits;
        if (kps < 8) kps = 8;
        kps += kps & 1;
        rsplits = (nks + kps - 1) / kps;
      }
      const size_t bot_cols = rsplits > 1 && h->mb > h->w ? h->mb : h->w;
      hipLaunchKernelGGL(k_recombine_bottom, dim3((unsigned)((Bh + 63) / 64), (unsigned)((bot_cols + 63) / 64)), dim3(256), 0, sx, h->mb, h->w,
                         h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dP + b0, Bh, d_e + b0 * m, m, rsplits > 1 ? 1 : 0);      // also zeroes the top part for the split-K form
      hipLaunchKernelGGL(k_recombine_mfma, dim3((unsigned)((Bh + 127) / 128), (unsigned)((h->mb + 127) / 128), (unsigned)rsplits), dim3(256), RC_LDS, sx, h->dR,
                         h->ldr, h->mb, nks, h->dZlo + 16 * b0, h->dZhi + 16 * b0, ld, h->dFail, h->dP + b0, Bh, d_e + b0 * m, m, big ? 1 : 0, kps);
    }
  };
  // PSF_HALVES=1: the batch's two halves run these stages on two streams, so that the int8 matrix-core kernels of one half (Z_q product, recombination)
  // can share the chip with the vector-bound samplers of the other (vector work hides behind int8 / bf16 MFMAs, unlike behind FP64 ones: profiles/r03_notes.md)
  const char* henv = psf_exp_env("PSF_HALVES");
  const bool halves = !pipe && henv && std::atoi(henv) != 0 && B >= 512 && B % 256 == 0;      // (at 32 / 64 preimages the two halves last as long as the whole: profiles/r06_notes.md)
  if (!halves) {
    tail(s2, 0, B);
  } else {
    const size_t Bh = B / 2;
    HIP_TRY(hipEventRecord(h->evT[0], st));
    HIP_TRY(hipStreamWaitEvent(h->s1, h->evT[0], 0));
    HIP_TRY(hipStreamWaitEvent(h->aux, h->evT[0], 0));
    tail(h->s1, 0, Bh);
    tail(h->aux, Bh, B - Bh);
    HIP_TRY(hipEventRecord(h->evP[0], h->s1));
    HIP_TRY(hipEventRecord(h->evP[1], h->aux));
    HIP_TRY(hipStreamWaitEvent(user_st, h->evP[0], 0));
    HIP_TRY(hipStreamWaitEvent(user_st, h->evP[1], 0));
  }
  if (pipe) {
    HIP_TRY(hipEventRecord(h->evP[cur], h->aux));
    HIP_TRY(hipStreamWaitEvent(user_st, h->evP[cur], 0));   // results of this call are ordered before later work on the caller's stream
  }
  HIP_TRY(hipGetLastError());
  if (gate_rc != PSF_OK) return gate_rc;
  h->last_stream = user_st;
  if (h->multi_t0 && h->multi_launched_ms < 0.0)
    h->multi_launched_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - *h->multi_t0).count();
  return PSF_OK;
}

psf_status psfp_last_status(psfp_handle* h) {
  if (!h) return PSF_ERR_PARAM;
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  int f = 0, f2 = 0;
  HIP_TRY(hipMemcpy(&f, h->sets[0].dFail, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&f2, h->sets[1].dFail, sizeof(int), hipMemcpyDeviceToHost));
  return (f | f2) ? PSF_ERR_SAMPLER : PSF_OK;
}

psf_status psfp_samp_p_dev(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_e, void* stream) {
  if (!h || (B && (!d_u || !d_e))) return PSF_ERR_PARAM;
  if (!h->has_key || !h->has_pub) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  psf_status rc = ensure_batch(h, B);
  if (rc != PSF_OK) return rc;
  if (h->timing) clear_slots(h);       // once per public call: the slices of a host-pointer call add up in psfp_get_timing
  return run_samp_p(h, seed, first_index, B, d_u, d_e, (hipStream_t)stream);
}

// ---- host-pointer entry points ---------------------------------------------------------------------------------------------------------------
constexpr size_t SIO_MAX_BYTES = (size_t)1 << 20;           // calls whose u + e fit this take the one-buffer form (at 4 MB the runtime's copies are faster again: 1.27 vs 1.17 ms at C3, 16 preimages)
// flags of a small call into the pinned buffer: [0] = a[0] | b[0] (the two failure words psfp_last_status reads), [1 ..] = c[0 .. nc)
__global__ void k_sio_flags(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ c, int nc, int* __restrict__ out) {
  const int t = threadIdx.x;
  if (t == 0) out[0] = (a ? a[0] : 0) | (b ? b[0] : 0);
  if (t >= 1 && t <= nc) out[t] = c[t - 1];
}
static psf_status sio_ensure(psfp_handle* h, size_t bytes) {
  if (bytes <= h->sio_cap) return PSF_OK;
  if (h->sio_pin) { hipHostFree(h->sio_pin); h->sio_pin = nullptr; h->sio_cap = 0; }
  const size_t cap = round_up(bytes, (size_t)64 << 10);
  HIP_TRY(hipHostMalloc(&h->sio_pin, cap, hipHostMallocDefault));
  h->sio_cap = cap;
  return PSF_OK;
}
static inline unsigned sio_grid(size_t words) { const size_t g = (words / 2 + 255) / 256; return (unsigned)(g < 1 ? 1 : g > 64 ? 64 : g); }
// u -> pinned -> d_u (kernel); [the caller's launches]; d_e -> pinned, flags -> pinned (kernels); one synchronisation; pinned -> e.  `flags_out` receives
// 1 + nc ints.  `run` enqueues the call on the null stream and returns its status.
static psf_status sio_call(psfp_handle* h, size_t nu, size_t ne, const uint64_t* u, int64_t* e, uint64_t* d_u, int64_t* d_e, const int* fa, const int* fb, const int* fc, int nc,
                           int* flags_out, const std::function<psf_status()>& run) {
  const size_t ub = round_up(nu * 8, 64), eb = round_up(ne * 8, 64);
  psf_status rc = sio_ensure(h, ub + eb + 64);
  if (rc != PSF_OK) return rc;
  uint64_t* hu = reinterpret_cast<uint64_t*>(h->sio_pin);
  int64_t* he = reinterpret_cast<int64_t*>(h->sio_pin + ub);
  int* hf = reinterpret_cast<int*>(h->sio_pin + ub + eb);
  std::memcpy(hu, u, nu * 8);
  hipLaunchKernelGGL(k_copy_words, dim3(sio_grid(nu)), dim3(256), 0, nullptr, hu, d_u, nu);
  rc = run();
  if (rc != PSF_OK) { hipStreamSynchronize(nullptr); return rc; }
  hipLaunchKernelGGL(k_copy_words, dim3(sio_grid(ne)), dim3(256), 0, nullptr, reinterpret_cast<const uint64_t*>(d_e), reinterpret_cast<uint64_t*>(he), ne);
  hipLaunchKernelGGL(k_sio_flags, dim3(1), dim3(64), 0, nullptr, fa, fb, fc, nc, hf);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(nullptr));
  for (int i = 0; i <= nc; ++i) flags_out[i] = hf[i];
  std::memcpy(e, he, ne * 8);
  return PSF_OK;
}

// int32 -> int64 into the caller's rows with streaming stores: the destination is written once and not read here, so there is no point in pulling its
// lines into the cache first (a plain loop moves 20 bytes per entry through the memory system, this one 12) -- the widening of a C3 batch is 1.5 GB of
// host memory traffic per call and has to fit under the next call's 60 ms on a handful of threads.
static void widen_rows(int64_t* __restrict__ dst, const int32_t* __restrict__ src, size_t cnt) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef int v2i __attribute__((ext_vector_type(2)));
  typedef long long v2l __attribute__((ext_vector_type(2)));
  size_t i = 0;
  while (i < cnt && (reinterpret_cast<uintptr_t>(dst + i) & 15)) { dst[i] = (int64_t)src[i]; ++i; }
  for (; i + 4 <= cnt; i += 4) {
    v4i x;
    std::memcpy(&x, src + i, sizeof(x));
    const v2i a = __builtin_shufflevector(x, x, 0, 1), b = __builtin_shufflevector(x, x, 2, 3);
    __builtin_nontemporal_store(__builtin_convertvector(a, v2l), reinterpret_cast<v2l*>(dst + i));
    __builtin_nontemporal_store(__builtin_convertvector(b, v2l), reinterpret_cast<v2l*>(dst + i + 2));
  }
  for (; i < cnt; ++i) dst[i] = (int64_t)src[i];
  std::atomic_thread_fence(std::memory_order_seq_cst);      // (streaming stores are weakly ordered: fence before the thread reports its chunk done)
}

// The flags of an asynchronous call: cleared and sent to pinned host memory by one-wave kernels in stream order.  (hipMemsetAsync / hipMemcpyAsync on the
// compute stream go through the runtime's copy path, where they queue behind the chunk copies of the call before: the next call's kernels then waited for them.)
__global__ void k_host_flags_clear(int* __restrict__ fail, int* __restrict__ ovf) { if (threadIdx.x < 4) fail[threadIdx.x] = 0; if (threadIdx.x < 2) ovf[threadIdx.x] = 0; }
// extra: the eight flag words of a PSFGPV / PSFGPVRing call (psfgpv_impl.hpp: [0] and [4] = a sampler failure of the first / second pass), or nullptr
__global__ void k_host_flags_send(const int* __restrict__ fail, const int* __restrict__ ovf, const int* __restrict__ extra, int* __restrict__ host_flags) {
  if (threadIdx.x == 0) {
    host_flags[0] = fail[0] | (extra ? (extra[0] | extra[4]) : 0);
    host_flags[1] = fail[1]; host_flags[2] = ovf[0]; host_flags[3] = extra ? 1 : 0;        // [3]: a nearest-plane call (an overflow of the narrowing is not a sampler failure there)
    __threadfence_system();
  }
}

// wait for the asynchronous call in `slot` (its workers have copied and widened every row), release it, return its status
static psf_status hp_join(psfp_handle* h, int slot) {
  auto& hp = h->hp;
  if (!hp.busy[slot]) return PSF_OK;
  for (auto& t : hp.workers[slot]) if (t.joinable()) t.join();
  hp.workers[slot].clear();
  hp.busy[slot] = false;
  psf_status rc = (psf_status)hp.status[slot].load();
  if (rc == PSF_OK && hp.hFlags[slot][0]) rc = PSF_ERR_SAMPLER;
  // an entry beyond 32 bits: impossible for PSFPerturbation (every entry is checked on the device: a sampler failure); a PSFGPV / PSFGPVRing row of that size
  // needs the synchronous call, which copies 64-bit rows then
  if (rc == PSF_OK && hp.hFlags[slot][2]) rc = hp.hFlags[slot][3] ? PSF_ERR_UNSUPPORTED : PSF_ERR_SAMPLER;
  hp.done[hp.slot_seq[slot] & 7] = psfp_handle::HostPipe::Done{hp.slot_seq[slot], (int)rc, true};      // whoever joins consumes the status: the ticket keeps it
  return rc;
}

static void hp_release(psfp_handle* h) {
  if (h->hp_warm.joinable()) h->hp_warm.join();
  auto& hp = h->hp;
  for (int s = 0; s < 2; ++s) hp_join(h, s);
  for (int s = 0; s < 2; ++s) {
    hipFree(hp.dE32[s]); hp.dE32[s] = nullptr;
    if (hp.hFlags[s]) { hipHostFree(hp.hFlags[s]); hp.hFlags[s] = nullptr; }
    for (auto& ev : hp.evSlice[s]) if (ev) { hipEventDestroy(ev); ev = nullptr; }
  }
  for (int s = 0; s < 2; ++s)
    for (int w = 0; w < psfp_handle::HostPipe::NW; ++w)
      for (int k = 0; k < 2; ++k) {
        if (hp.hbuf[s][w][k]) { hipHostFree(hp.hbuf[s][w][k]); hp.hbuf[s][w][k] = nullptr; }
        if (hp.evC[s][w][k]) { hipEventDestroy(hp.evC[s][w][k]); hp.evC[s][w][k] = nullptr; }
        if (hp.sigC[s][w][k].handle) { hp.sdma.drop_signal(hp.sigC[s][w][k]); hp.sigC[s][w][k].handle = 0; }
      }
  if (hp.sigU.handle) { hp.sdma.drop_signal(hp.sigU); hp.sigU.handle = 0; }
  hp.sdma.close();
  hipFree(hp.dOvf); hp.dOvf = nullptr;
  for (int s = 0; s < 2; ++s) {
    if (hp.hU[s]) { hipHostFree(hp.hU[s]); hp.hU[s] = nullptr; }
    hipFree(hp.dU2[s]); hp.dU2[s] = nullptr;
  }
  hp.u_cap[0] = hp.u_cap[1] = 0;
  if (hp.copy) { hipStreamDestroy(hp.copy); hp.copy = nullptr; }
  if (hp.compute) { hipStreamDestroy(hp.compute); hp.compute = nullptr; }
  hp.cap_entries[0] = hp.cap_entries[1] = 0; hp.chunk_entries = 0;
  hp.slot_ready[0] = hp.slot_ready[1] = false; hp.common_ready = false;
}

// streams, transport and the rings of call slot `slot` (which the caller has joined).  Everything is allocated on first use and per slot: a caller that only ever
// makes synchronous calls pays for one slot (pinning memory is the expensive part of a handle's first host-pointer call).
static psf_status hp_ensure(psfp_handle* h, int slot, size_t entries, size_t u_words, bool from_prewarm) {
  if (!from_prewarm && h->hp_warm.joinable()) h->hp_warm.join();      // (the prewarm worker itself never looks at h->hp_warm: the owner may still be assigning it)
  auto& hp = h->hp;
  constexpr int NW = psfp_handle::HostPipe::NW;
  if (!hp.common_ready) {
    {  // (matters for the HIP-copy transports only: their copies are shader kernels, which on a queue of lower priority than the compute stream ran only when
       // that stream was idle -- copies on the HIGH-priority queue, the asynchronous calls' kernels on a normal one)
      int lo_prio = 0, hi_prio = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
      int pc = hi_prio, pk = (lo_prio + hi_prio) / 2;
      if (const char* env = psf_exp_env("PSF_HOST_PRIO")) { if (std::atoi(env) == 0) pc = pk; else if (std::atoi(env) == 2) { pc = pk; pk = hi_prio; } }      // experiments: 0 = equal, 2 = compute high
      if (!hp.copy) HIP_TRY(hipStreamCreateWithPriority(&hp.copy, hipStreamNonBlocking, pc));
      if (!hp.compute) HIP_TRY(hipStreamCreateWithPriority(&hp.compute, hipStreamNonBlocking, pk));
    }
    if (!hp.dOvf) HIP_TRY(hipMalloc(&hp.dOvf, 2 * sizeof(int)));
    hp.chunk_entries = (size_t)2 << 20;                                  // 8 MiB of int32 per chunk
    if (const char* env = std::getenv("PSF_HOST_WORKERS")) { const int v = std::atoi(env); if (v >= 1 && v <= NW) hp.nw = v; }
    if (const char* env = psf_exp_env("PSF_HOST_CHUNK_MB")) { const long v = std::atol(env); if (v >= 1 && v <= 256) hp.chunk_entries = (size_t)v << 18; }
    if (const char* env = psf_exp_env("PSF_HOST_COPY")) {                 // sdma (default) | runtime | kernel[:workgroups]
      if (std::strncmp(env, "runtime", 7) == 0) hp.copy_mode = 0;
      else if (std::strncmp(env, "kernel", 6) == 0) { hp.copy_mode = 2; if (env[6] == ':') { const int g = std::atoi(env + 7); if (g >= 1 && g <= 4096) hp.copy_grid = g; } }
    }
    if (hp.copy_mode == 1) {
      int dom = 0, bus = 0, dv = 0;
      HIP_TRY(hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, h->prm.device));
      HIP_TRY(hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, h->prm.device));
      HIP_TRY(hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, h->prm.device));
      if (!hp.sdma.open(dom, bus, dv)) hp.copy_mode = 0;                  // no HSA agent for this device: the HIP copies (slower under overlap, same rows)
      else if (!hp.sigU.handle && !hp.sdma.make_signal(&hp.sigU)) return PSF_ERR_HIP;
    }
    hp.common_ready = true;
  }
  if (!hp.slot_ready[slot]) {                                 // (every piece behind its own test: a call that failed half-way is completed, not repeated, by the next)
    if (!hp.hFlags[slot]) HIP_TRY(hipHostMalloc(&hp.hFlags[slot], 4 * sizeof(int), hipHostMallocDefault));
    for (auto& ev : hp.evSlice[slot]) if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int w = 0; w < hp.nw; ++w)
      for (int k = 0; k < 2; ++k) {
        if (!hp.hbuf[slot][w][k]) HIP_TRY(hipHostMalloc(&hp.hbuf[slot][w][k], hp.chunk_entries * sizeof(int32_t), hipHostMallocDefault));
        if (!hp.evC[slot][w][k]) HIP_TRY(hipEventCreateWithFlags(&hp.evC[slot][w][k], hipEventDisableTiming));
        if (hp.copy_mode == 1 && !hp.sigC[slot][w][k].handle && !hp.sdma.make_signal(&hp.sigC[slot][w][k])) return PSF_ERR_HIP;
      }
    hp.slot_ready[slot] = true;
  }
  if (entries > hp.cap_entries[slot]) {
    hipFree(hp.dE32[slot]); hp.dE32[slot] = nullptr;
    hp.cap_entries[slot] = 0;
    HIP_TRY(hipMalloc(&hp.dE32[slot], entries * sizeof(int32_t) + 8));      // (+8: the copy kernel moves 8-byte words)
    hp.cap_entries[slot] = entries;
  }
  if (u_words > hp.u_cap[slot]) {
    if (hp.hU[slot]) hipHostFree(hp.hU[slot]);
    hp.hU[slot] = nullptr; hipFree(hp.dU2[slot]); hp.dU2[slot] = nullptr;
    hp.u_cap[slot] = 0;
    HIP_TRY(hipHostMalloc(&hp.hU[slot], u_words * sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(hipMalloc(&hp.dU2[slot], u_words * sizeof(uint64_t)));
    hp.u_cap[slot] = u_words;
  }
  return PSF_OK;
}

// Asynchronous samp_p on host buffers: returns once the work is enqueued (the targets have been staged); `e` is complete when psfp_wait returns.
// At most two calls are in flight: a third waits for the first.  The compute stream runs the slices of the call back to back (row b draws from global
// index first_index + b, so slicing changes no bit); behind each slice its rows are narrowed to int32 (every entry of a preimage is below 2^31: |p| < 2^23
// and |R z| <= w 2^15, both checked on the device), copied in chunks to pinned memory on a second stream and widened into `e` by NW worker threads --
// while the compute stream is already in the next slice or the next call.  A single call therefore ends one short slice after its product
// (slices: all but the last ~1024 rows, then the rest), and back-to-back calls run at the device-resident rate.
// What an asynchronous host-pointer call of any of the three types is made of: `compute(off, cnt, d_u, d_e, cs)` enqueues the samp_p pipeline of the rows
// [off, off + cnt) on the stream cs (targets d_u: cnt x n, preimages d_e: cnt x m, both on the device); everything around it -- slots, staging of the targets,
// int32 narrowing, chunk transfers by the DMA engines, widening workers, the flags -- is the same.  defer_u: the targets are uploaded when the pipeline asks for them
// (h->before_u, PSFPerturbation: behind the product); otherwise at the head of the call.  extra_flags: see k_host_flags_send.
using HpCompute = std::function<psf_status(size_t, size_t, const uint64_t*, int64_t*, hipStream_t)>;
static psf_status hp_async(psfp_handle* h, size_t B, const uint64_t* u, int64_t* e, bool defer_u, bool allow_slices, const int* extra_flags, const HpCompute& compute);

psf_status psfp_samp_p_async(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e) {
  if (!h || (B && (!u || !e))) return PSF_ERR_PARAM;
  if (!h->has_key || !h->has_pub) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  if (h->pipeline) return PSF_ERR_UNSUPPORTED;              // PSF_PIPELINE=1 alternates two sets of failure words per call: the synchronous forms only (as psfp_samp_p does)
  struct FailGuard { psfp_handle* h; size_t B; ~FailGuard() { h->keep_fail = false; h->nbj = round_up(B, TR_BN) / TR_BN; } } guard{h, B};
  return hp_async(h, B, u, e, true, true, nullptr, [&](size_t off, size_t cnt, const uint64_t* d_u, int64_t* d_e, hipStream_t cs) -> psf_status {
    h->keep_fail = true;                                    // the slices of a call share its failure words (cleared once, in front of the first)
    h->nbj = round_up(cnt, TR_BN) / TR_BN;
    return run_samp_p(h, seed, first_index + off, cnt, d_u, d_e, cs);
  });
}

static psf_status hp_async(psfp_handle* h, size_t B, const uint64_t* u, int64_t* e, bool defer_u, bool allow_slices, const int* extra_flags, const HpCompute& compute) {
  HIP_TRY(hipSetDevice(h->prm.device));
  auto& hp = h->hp;
  const size_t m = h->m, total = B * m;
  if (!hp.busy[0] && !hp.busy[1]) hp.next = 0;              // nothing in flight: slot 0 (a caller that only makes synchronous calls never needs -- or allocates -- the second)
  const int slot = (int)(hp.next & 1);
  psf_status rc = hp_join(h, slot);                         // the call before last used this slot
  if (rc != PSF_OK) return rc;
  if (B > h->Bcap) { rc = hp_join(h, slot ^ 1); if (rc != PSF_OK) return rc; }      // ensure_batch reallocates: nothing may be in flight
  rc = ensure_batch(h, B);
  if (rc != PSF_OK) return rc;
  rc = hp_ensure(h, slot, total, B * h->n);
  if (rc != PSF_OK) return rc;
  ++hp.next;
  hp.slot_seq[slot] = hp.seq++;
  hipStream_t cs = hp.compute;
  // targets: pageable -> pinned (this thread) -> this call's device copy -- deferred until the syndrome stage of the first slice is about to be enqueued
  // (h->before_u, run_samp_p): by then the product is executing, and neither the staging copy nor the upload delays the call
  h->before_u = [h, slot, u, B, cs]() -> psf_status {
    auto& hp = h->hp;
    std::memcpy(hp.hU[slot], u, B * h->n * sizeof(uint64_t));
    if (hp.copy_mode == 1) {                                // by the DMA engine (dU2[slot]'s last reader was joined before this call began)
      if (!hp.sdma.start_upload(hp.dU2[slot], hp.hU[slot], B * h->n * sizeof(uint64_t), hp.sigU) || !hp.sdma.wait(hp.sigU)) return PSF_ERR_HIP;
    } else {
      hipLaunchKernelGGL(k_copy_words, dim3(64), dim3(256), 0, cs, hp.hU[slot], hp.dU2[slot], B * h->n);  // (see k_copy_words: a HIP copy would queue behind the download before)
    }
    return PSF_OK;
  };
  struct GateGuard { psfp_handle* h; ~GateGuard() { h->before_u = nullptr; } } gate_guard{h};      // (an error return must not leave a callback with dead captures behind)
  if (hp.copy_mode != 1 || !defer_u) {                      // the copy kernel is ordered by the compute stream only: at the head of the call, as before
    auto f = std::move(h->before_u); h->before_u = nullptr;
    rc = f();
    if (rc != PSF_OK) return rc;
  }
  const uint64_t* dUcall = hp.dU2[slot];
  if (h->timing) clear_slots(h);
  // slices: everything but a short tail, then the tail (its transfer is all that remains exposed behind the last kernel)
  size_t cuts[5] = {0, B, B, B, B};
  int nsl = 1;
  // (two slices cost the product ~4 ms)
  size_t tail = 1024;
  if (const char* env = psf_exp_env("PSF_HOST_TAIL")) { const long v = std::atol(env); if (v >= 128) tail = (size_t)v; }
  bool cut = hp.slice_tail;                                 // the synchronous form only: behind an asynchronous call the next call's compute covers the transfer
  if (const char* env = psf_exp_env("PSF_HOST_ASYNC_SLICE")) cut = cut || std::atoi(env) != 0;      // experiments: 1 = asynchronous calls cut the tail slice too
  if (cut && allow_slices && !h->no_slice && !h->pipeline && B >= 2 * tail) { cuts[1] = B - tail; cuts[2] = B; nsl = 2; }
  if (const char* env = psf_exp_env("PSF_HOST_SLICE")) {    // experiments: equal slices of this many rows (at most four)
    const long v = std::atol(env);
    if (v >= 128 && allow_slices && !h->no_slice && !h->pipeline && (size_t)v < B) {
      nsl = 0;
      for (size_t off = 0; off < B && nsl < 4; off += (size_t)v) cuts[nsl++] = off;
      cuts[nsl] = B;
    }
  }
  hipLaunchKernelGGL(k_host_flags_clear, dim3(1), dim3(64), 0, cs, h->dFail, hp.dOvf);
  for (int j = 0; j < nsl; ++j) {
    const size_t off = cuts[j], cnt = cuts[j + 1] - cuts[j];
    rc = compute(off, cnt, dUcall + off * h->n, h->dE + off * m, cs);
    if (rc != PSF_OK) return rc;
    hipLaunchKernelGGL(k_narrow_rows, dim3(grid_for(cnt * m / 2 + 1, 256, 256 * 16)), dim3(256), 0, cs, h->dE + off * m, hp.dE32[slot] + off * m, cnt * m, hp.dOvf);
    if (j == nsl - 1) {                                     // the call's flags travel with its last slice
      hipLaunchKernelGGL(k_host_flags_send, dim3(1), dim3(64), 0, cs, h->dFail, hp.dOvf, extra_flags, hp.hFlags[slot]);
    }
    HIP_TRY(hipEventRecord(hp.evSlice[slot][j], cs));
  }
  HIP_TRY(hipGetLastError());
  h->last_stream = cs;
  // workers: chunk c of the call's entries belongs to worker c % NW; a worker copies its chunk into one of its two pinned buffers and widens the
  // previous one meanwhile
  const size_t CE = hp.chunk_entries, nchunks = (total + CE - 1) / CE;
  hp.status[slot] = (int)PSF_OK;
  hp.busy[slot] = true;
  const int32_t* src = hp.dE32[slot];
  const int device = h->prm.device;
  size_t slice_end[4]; hipEvent_t slice_ev[4];
  for (int j = 0; j < nsl; ++j) { slice_end[j] = cuts[j + 1] * m; slice_ev[j] = hp.evSlice[slot][j]; }
  const int nw = hp.nw;
  int dbg = 0;
  if (const char* env = psf_exp_env("PSF_HOST_DEBUG")) dbg = std::atoi(env);      // measurement only: 1 = no widening, 2 = no copies either (e is NOT filled)
  const int copy_mode = hp.copy_mode, copy_grid = hp.copy_grid;
  const bool plain_widen = psf_exp_env("PSF_HOST_PLAIN_WIDEN") != nullptr;      // measurement only: the scalar loop with ordinary stores
  auto worker = [&hp, slot, src, e, total, CE, nchunks, nsl, device, slice_end, slice_ev, nw, dbg, copy_mode, copy_grid, plain_widen](int w) {
    if (hipSetDevice(device) != hipSuccess) { hp.status[slot] = (int)PSF_ERR_HIP; return; }
    auto widen = [&](size_t c, int k) {
      if (copy_mode == 1 ? (dbg < 2 && !hp.sdma.wait(hp.sigC[slot][w][k])) : hipEventSynchronize(hp.evC[slot][w][k]) != hipSuccess) { hp.status[slot] = (int)PSF_ERR_HIP; return; }
      const size_t b0 = c * CE, cnt = total - b0 < CE ? total - b0 : CE;
      const int32_t* hs = hp.hbuf[slot][w][k];
      int64_t* dst = e + b0;
      if (dbg) return;
      if (plain_widen) { for (size_t i = 0; i < cnt; ++i) dst[i] = (int64_t)hs[i]; }
      else widen_rows(dst, hs, cnt);
    };
    auto move_chunk = [&](int32_t* dst, const int32_t* from, size_t cnt) -> hipError_t {
      if (copy_mode == 0) return hipMemcpyAsync(dst, from, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, hp.copy);
      hipLaunchKernelGGL(k_copy_words, dim3(copy_grid), dim3(256), 0, hp.copy, reinterpret_cast<const uint64_t*>(from), reinterpret_cast<uint64_t*>(dst), (cnt + 1) / 2);
      return hipGetLastError();
    };
    long prev = -1; int pk = 0, k = 0;
    for (size_t c = (size_t)w; c < nchunks; c += (size_t)nw) {
      const size_t b0 = c * CE, cnt = total - b0 < CE ? total - b0 : CE;
      int j = 0;
      while (j < nsl - 1 && b0 + cnt > slice_end[j]) ++j;                // the last slice this chunk touches
      if (copy_mode == 1) {                                              // the slice's rows are complete (host wait), then the DMA engine moves the chunk
        if (hipEventSynchronize(slice_ev[j]) != hipSuccess ||
            (dbg < 2 && !hp.sdma.start(hp.hbuf[slot][w][k], src + b0, cnt * sizeof(int32_t), hp.sigC[slot][w][k]))) { hp.status[slot] = (int)PSF_ERR_HIP; break; }
      } else if (hipStreamWaitEvent(hp.copy, slice_ev[j], 0) != hipSuccess ||
                 (dbg < 2 ? move_chunk(hp.hbuf[slot][w][k], src + b0, cnt) : hipSuccess) != hipSuccess ||
                 hipEventRecord(hp.evC[slot][w][k], hp.copy) != hipSuccess) { hp.status[slot] = (int)PSF_ERR_HIP; break; }
      if (prev >= 0) widen((size_t)prev, pk);
      prev = (long)c; pk = k; k ^= 1;
    }
    if (prev >= 0) widen((size_t)prev, pk);
  };
  try {
    for (int w = 0; w < nw && (size_t)w < nchunks; ++w) hp.workers[slot].emplace_back(worker, w);
  } catch (...) {                                                        // no thread available: the started ones finish, the rest of the rows are missing
    hp.status[slot] = (int)PSF_ERR_HIP;
  }
  // (the call's flags were copied on the compute stream in front of the last slice's event, which the worker of the last chunk waits for: once the
  // workers have been joined the flags have landed)
  return PSF_OK;
}

// all asynchronous calls of this handle have completed: their rows are in the callers' buffers; the first non-OK status (oldest call first)
psf_status psfp_wait(psfp_handle* h) {
  if (!h) return PSF_ERR_PARAM;
  HIP_TRY(hipSetDevice(h->prm.device));
  auto& hp = h->hp;
  psf_status first = PSF_OK;
  for (int i = 0; i < 2; ++i) {
    const int slot = (int)((hp.next + (size_t)i) & 1);                   // oldest first
    const psf_status rc = hp_join(h, slot);
    if (first == PSF_OK) first = rc;
  }
  return first;
}

// the ticket the next asynchronous call of this handle will carry, and the status of ONE asynchronous call by its ticket (waits for it and for the older call
// in flight, nothing newer): psfp_wait returns the first failure of everything outstanding and thereby consumes the statuses of calls the caller may not be
// asking about -- a caller that keeps several batches (the shim's PendingBatch) asks per ticket.  PSF_ERR_PARAM: a ticket never issued or older than the last 8 joined calls.
uint64_t psfp_async_next_ticket(const psfp_handle* h) { return h ? h->hp.seq : 0; }
psf_status psfp_wait_ticket(psfp_handle* h, uint64_t ticket) {
  if (!h) return PSF_ERR_PARAM;
  HIP_TRY(hipSetDevice(h->prm.device));
  auto& hp = h->hp;
  if (ticket >= hp.seq) return PSF_ERR_PARAM;
  for (int i = 0; i < 2; ++i) {                                         // oldest first, up to the ticket's own call
    const int slot = (int)((hp.next + (size_t)i) & 1);
    if (hp.busy[slot] && hp.slot_seq[slot] <= ticket) hp_join(h, slot);
  }
  const auto& d = hp.done[ticket & 7];
  return (d.used && d.seq == ticket) ? (psf_status)d.status : PSF_ERR_PARAM;
}

psf_status psfp_samp_p(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e) {
  if (!h || (B && (!u || !e))) return PSF_ERR_PARAM;
  if (!h->has_key || !h->has_pub) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  if (B * h->m < ((size_t)1 << 20) || h->no_slice || h->pipeline) {
    // a single call / a handful of preimages (or a stage export): nothing to overlap -- straight through on the default stream
    psf_status rc = psfp_wait(h);
    if (rc != PSF_OK) return rc;
    rc = ensure_batch(h, B);
    if (rc != PSF_OK) return rc;
    if (!h->no_slice && !h->pipeline && B * (h->n + h->m) * 8 <= SIO_MAX_BYTES && !psf_exp_env("PSF_HOST_STRAIGHT")) {
      if (h->timing) clear_slots(h);
      int fl[1] = {0};
      rc = sio_call(h, B * h->n, B * h->m, u, e, h->dU, h->dE, h->sets[0].dFail, h->sets[1].dFail, nullptr, 0, fl,
                    [&]() { return run_samp_p(h, seed, first_index, B, h->dU, h->dE, nullptr); });
      if (rc != PSF_OK) return rc;
      return fl[0] ? PSF_ERR_SAMPLER : PSF_OK;
    }
    HIP_TRY(hipMemcpy(h->dU, u, B * h->n * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (h->timing) clear_slots(h);
    rc = run_samp_p(h, seed, first_index, B, h->dU, h->dE, nullptr);
    if (rc != PSF_OK) return rc;
    rc = psfp_last_status(h);
    HIP_TRY(hipMemcpy(e, h->dE, B * h->m * sizeof(int64_t), hipMemcpyDeviceToHost));
    return rc;
  }
  h->hp.slice_tail = true;
  const psf_status rc = psfp_samp_p_async(h, seed, first_index, B, u, e);
  h->hp.slice_tail = false;
  const psf_status rw = psfp_wait(h);
  return rc != PSF_OK ? rc : rw;
}

// One job over several handles (one per GPU of the node, each with the same key): rows are cut into contiguous shares
// (psf_shard_range), share i is computed by handles[i] on its own device and stream; row b draws from the global index first_index + b,
// so the result equals the single-handle one bit for bit.  Host buffers; no collective is involved -- the "gather" of SURVEY.md 8e is
// each device's copy into its slice of e.
// Concurrency: ONE WORKER THREAD PER HANDLE.  The caller's u / e are pageable memory, and HIP's "asynchronous" copies to or from pageable
// memory block the calling host thread until the device has drained -- issued from one thread (round 2) the devices ran one after another.
// Each worker owns its device for the call: upload, the samp_p launch sequence, download (through the handle's own slicing, so the rows of
// the first half cross PCIe while the second half is computed), status.  A failure on one device does not leave work of another in flight:
// every worker runs to completion and synchronises its own stream before the call returns; the first non-OK status in handle order is returned.
// Per handle the call records, relative to its own start (host clock, ms): when the worker had enqueued its first samp_p launch sequence and when its
// last byte had landed in e -- psfp_get_multi_timing; overlap between the handles' [launched, done] windows is what a test can assert even with
// every handle on one GPU.
psf_status psfp_samp_p_multi(psfp_handle* const* handles, int count, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e) {
  if (!handles || count < 1 || (B && (!u || !e))) return PSF_ERR_PARAM;
  for (int i = 0; i < count; ++i) {
    if (!handles[i]) return PSF_ERR_PARAM;
    if (!handles[i]->has_key || !handles[i]->has_pub) return PSF_ERR_NO_KEY;
    if (handles[i]->n != handles[0]->n || handles[i]->m != handles[0]->m || handles[i]->q != handles[0]->q) return PSF_ERR_PARAM;
    for (int j = 0; j < i; ++j) if (handles[j] == handles[i]) return PSF_ERR_PARAM;      // one worker per handle: a handle may appear once
  }
  if (B == 0) return PSF_OK;
  std::vector<size_t> first(count), cnt(count);
  for (int i = 0; i < count; ++i) psf_shard_range(B, count, i, &first[i], &cnt[i]);
  std::vector<psf_status> rc(count, PSF_OK);
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(t - t0).count(); };
  auto work = [&](int i) {
    psfp_handle* h = handles[i];
    h->multi_launched_ms = h->multi_done_ms = -1.0;
    if (!cnt[i]) return;
    h->multi_t0 = &t0;
    rc[i] = psfp_samp_p(h, seed, first_index + first[i], cnt[i], u + first[i] * h->n, e + first[i] * h->m);
    h->multi_t0 = nullptr;
    h->multi_done_ms = ms_since(std::chrono::steady_clock::now());
  };
  std::vector<std::thread> pool;
  int started = 1;
  try {                                                  // nothing may unwind across the extern "C" boundary
    pool.reserve(count > 1 ? count - 1 : 0);
    for (int i = 1; i < count; ++i) { pool.emplace_back(work, i); started = i + 1; }
  } catch (...) {
    for (int i = started; i < count; ++i) rc[i] = PSF_ERR_HIP;      // these shares were never started (no thread available)
  }
  work(0);                                               // the calling thread serves the first handle
  for (auto& t : pool) t.join();
  for (int i = 0; i < count; ++i) if (rc[i] != PSF_OK) return rc[i];
  return PSF_OK;
}

// [launched, done] window of this handle inside the last psfp_samp_p_multi call, in ms since that call began (-1: the handle had no rows)
psf_status psfp_get_multi_timing(const psfp_handle* h, double* launched_ms, double* done_ms) {
  if (!h) return PSF_ERR_PARAM;
  if (launched_ms) *launched_ms = h->multi_launched_ms;
  if (done_ms) *done_ms = h->multi_done_ms;
  return PSF_OK;
}

psf_status psfp_samp_p_stages(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, double* d, double* x,
                              int64_t* p, uint64_t* v, int64_t* z, int64_t* e) {
  if (!h || !u || B == 0) return PSF_ERR_PARAM;
  std::vector<int64_t> etmp(B * h->m);
  h->no_slice = true;
  psf_status rc = psfp_samp_p(h, seed, first_index, B, u, etmp.data());
  h->no_slice = false;
  if (rc != PSF_OK && rc != PSF_ERR_SAMPLER) return rc;
  const size_t m = h->m, ld = h->ld;
  if (e) std::memcpy(e, etmp.data(), etmp.size() * sizeof(int64_t));
  void* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, B * m * sizeof(double)));
  if (d) {
    hipLaunchKernelGGL(k_export_normals, dim3(grid_for(B * m)), dim3(256), 0, 0, h->dDt, m, B, h->nkb, (double*)tmp, h->normals_ncf);
    HIP_TRY(hipMemcpy(d, tmp, B * m * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (x) {
    hipLaunchKernelGGL((k_export_T<double>), dim3(grid_for(B * m)), dim3(256), 0, 0, h->dX, m, B, ld, (double*)tmp);
    HIP_TRY(hipMemcpy(x, tmp, B * m * sizeof(double), hipMemcpyDeviceToHost));
  }
  std::vector<int64_t> ptmp;
  if (p || z) {
    ptmp.resize(B * m);
    hipLaunchKernelGGL(k_export_P, dim3(grid_for(B * m)), dim3(256), 0, 0, h->dP, m, B, ld, (int64_t*)tmp);
    HIP_TRY(hipMemcpy(ptmp.data(), tmp, B * m * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (p) std::memcpy(p, ptmp.data(), ptmp.size() * sizeof(int64_t));
  }
  if (v) {
    hipLaunchKernelGGL((k_export_T<uint64_t>), dim3(grid_for(B * h->n)), dim3(256), 0, 0, h->dV, h->n, B, ld, (uint64_t*)tmp);
    HIP_TRY(hipMemcpy(v, tmp, B * h->n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  if (z)  // e_bottom = p_bottom + z  (mp_perturbation.rs:335 with the identity block of [R; I])
    for (size_t b = 0; b < B; ++b)
      for (size_t c = 0; c < h->w; ++c) z[b * h->w + c] = etmp[b * m + h->mb + c] - ptmp[b * m + h->mb + c];
  hipFree(tmp);
  return rc;
}

psf_status psfp_samp_d_dev(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, int64_t* d_e, void* stream) {
  if (!h || (B && !d_e)) return PSF_ERR_PARAM;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(h->dFail, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_samp_d, dim3(grid_for(B * h->m, 256, 256 * 32)), dim3(256), 0, st, seed, first_index, h->m, B, h->szSR, d_e, h->dFail);
  HIP_TRY(hipGetLastError());
  h->last_stream = st;
  return PSF_OK;
}

psf_status psfp_samp_d(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, int64_t* e) {
  if (!h || (B && !e)) return PSF_ERR_PARAM;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  int64_t* de = nullptr;
  HIP_TRY(hipMalloc(&de, B * h->m * sizeof(int64_t)));
  psf_status rc = psfp_samp_d_dev(h, seed, first_index, B, de, nullptr);
  if (rc == PSF_OK) rc = psfp_last_status(h);
  HIP_TRY(hipMemcpy(e, de, B * h->m * sizeof(int64_t), hipMemcpyDeviceToHost));
  hipFree(de);
  return rc;
}

static double domain_bound(const psfp_handle* h) {   // s^2 * m * r^2, mp_perturbation.rs:401
  return ((h->prm.s * h->prm.s) * (double)h->m) * (h->prm.r * h->prm.r);
}

psf_status psfp_check_domain(psfp_handle* h, size_t B, const int64_t* e, size_t len, uint8_t* ok) {
  if (!h || (B && (!e || !ok))) return PSF_ERR_PARAM;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  if (len == 0) { std::memset(ok, 0, B); return PSF_OK; }
  int64_t* de = nullptr; uint8_t* dok = nullptr;
  HIP_TRY(hipMalloc(&de, B * len * sizeof(int64_t)));
  HIP_TRY(hipMalloc(&dok, B));
  HIP_TRY(hipMemcpy(de, e, B * len * sizeof(int64_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check_domain, dim3((unsigned)B), dim3(256), 0, 0, de, len, h->m, domain_bound(h), dok);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(ok, dok, B, hipMemcpyDeviceToHost));
  hipFree(de); hipFree(dok);
  return PSF_OK;
}

psf_status psfp_f_a_dev(psfp_handle* h, size_t B, const int64_t* d_e, uint64_t* d_u, uint8_t* d_ok, void* stream) {
  if (!h || (B && (!d_e || !d_u || !d_ok))) return PSF_ERR_PARAM;
  if (!h->has_pub) return PSF_ERR_NO_KEY;                     // f_a needs the public matrix only (mp_perturbation.rs:366-369)
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  PSFP_QUIESCE(h);
  psf_status rc = ensure_batch(h, B);
  if (rc != PSF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t m = h->m, ld = h->ld;
  hipLaunchKernelGGL(k_check_domain, dim3((unsigned)B), dim3(256), 0, st, d_e, m, m, domain_bound(h), d_ok);   // :367
  hipLaunchKernelGGL(k_narrow_transpose, dim3((unsigned)(ld / 64), (unsigned)((m + 63) / 64)), dim3(256), 0, st, d_e, m, B, ld, h->dPf, d_ok);
  launch_zq_mfma(h, st, ZQ_FA, h->dPf, h->dP8f, B, nullptr, d_u, h->n);                                                    // :368
  HIP_TRY(hipGetLastError());
  h->last_stream = st;
  return PSF_OK;
}

psf_status psfp_f_a(psfp_handle* h, size_t B, const int64_t* e, uint64_t* u) {
  if (!h || (B && (!e || !u))) return PSF_ERR_PARAM;
  if (!h->has_pub) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  { const psf_status rw = psfp_wait(h); if (rw != PSF_OK) return rw; }      // the handle's dU / dE may belong to an asynchronous samp_p in flight
  psf_status rc = ensure_batch(h, B);
  if (rc != PSF_OK) return rc;
  HIP_TRY(hipMemcpy(h->dE, e, B * h->m * sizeof(int64_t), hipMemcpyHostToDevice));
  rc = psfp_f_a_dev(h, B, h->dE, h->dU, h->dOk, nullptr);
  if (rc != PSF_OK) return rc;
  std::vector<uint8_t> ok(B);
  HIP_TRY(hipMemcpy(u, h->dU, B * h->n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ok.data(), h->dOk, B, hipMemcpyDeviceToHost));
  for (uint8_t o : ok) if (!o) return PSF_ERR_DOMAIN;
  return PSF_OK;
}

psf_status psfp_uniform_targets_dev(psfp_handle* h, uint64_t seed, uint64_t first_index, size_t B, uint64_t* d_u, void* stream) {
  if (!h || (B && !d_u)) return PSF_ERR_PARAM;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(h->prm.device));
  hipLaunchKernelGGL(k_uniform_targets, dim3(grid_for(B * h->n)), dim3(256), 0, (hipStream_t)stream, seed, first_index, h->n, B, h->q, d_u);
  HIP_TRY(hipGetLastError());
  h->last_stream = (hipStream_t)stream;
  return PSF_OK;
}

psf_status psf_narrow_rows_dev(const int64_t* d_src, int32_t* d_dst, size_t count, int* d_overflow, int device, void* stream) {
  if (count && (!d_src || !d_dst || !d_overflow)) return PSF_ERR_PARAM;
  if (((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 7)) return PSF_ERR_PARAM;
  if (count == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_narrow_rows, dim3(grid_for(count / 2 + 1, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, d_src, d_dst, count, d_overflow);
  HIP_TRY(hipGetLastError());
  return PSF_OK;
}

psf_status psfp_enable_timing(psfp_handle* h, int on) {
  if (!h) return PSF_ERR_PARAM;
  h->timing = on != 0;
  if (!on) clear_slots(h);
  return PSF_OK;
}

psf_status psfp_get_timing(psfp_handle* h, char* names, size_t names_len, double* ms, size_t* count) {
  if (!h || !count) return PSF_ERR_PARAM;
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  HIP_TRY(hipStreamSynchronize(h->aux));
  HIP_TRY(hipStreamSynchronize(h->s1));
  std::string joined;
  std::vector<std::string> names_v;
  std::vector<double> sums;
  for (auto& s : h->slots) {          // launches of the same kernel (one per slice) are summed
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.e0, s.e1) != hipSuccess) continue;
    size_t j = 0;
    while (j < names_v.size() && names_v[j] != s.name) ++j;
    if (j == names_v.size()) { names_v.push_back(s.name); sums.push_back(0.0); }
    sums[j] += t;
  }
  size_t nout = 0;
  for (size_t j = 0; j < names_v.size(); ++j) {
    if (ms && nout < *count) ms[nout] = sums[j];
    if (!joined.empty()) joined += ';';
    joined += names_v[j];
    ++nout;
  }
  if (names && names_len) { std::strncpy(names, joined.c_str(), names_len - 1); names[names_len - 1] = 0; }
  *count = nout;
  return PSF_OK;
}

}  // extern "C"

#include "psfgpv_impl.hpp"
