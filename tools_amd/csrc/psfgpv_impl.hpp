// psfgpv_impl.hpp -- PSFGPV behind the C ABI (gpv.rs:53-57, impl PSF :59-225).  Included at the end of psfp.hip:
// the Z_q machinery (A, its digit planes, f_a, check_domain, samp_d) is shared with PSFPerturbation through an inner
// psfp_handle created with r = 1 (then s*r = s and the domain bound s^2 m r^2 = s^2 m, gpv.rs:113-116, :219-224).

#include <algorithm>
struct psfgpv_handle {
  psfp_handle* base = nullptr;
  double s = 0;
  size_t n = 0, m = 0, dim = 0;       // dim = lattice dimension walked by the nearest plane (= m here)
  int32_t* dSt = nullptr;             // dim x dim, row i = basis vector i
  double* dGt = nullptr;              // dim x dim, row i = b~_i
  double* dNorm2 = nullptr;
  SampleZParams* dSz = nullptr;
  uint64_t* dT = nullptr;             // n x n solve operator, transposed
  uint32_t* dPiv = nullptr;           // n pivot columns
  // blocked nearest plane (psf_np_kernels.hpp): per key
  size_t nblk = 0, dpad = 0, nrb = 0, nkc = 0, nkb = 0;   // 64-row blocks; dim padded to 128; 128-row blocks; K chunks of the pivots; K chunks of dim
  double* dGp = nullptr;              // bulk panels g[< 64 J][block J], fragment order
  double* dGin = nullptr;             // in-block triangles, packed, nblk x NP_TRI
  double* dGnx = nullptr;             // panels between neighbouring blocks, nblk x 64 x 64
  NpRow* dRows = nullptr;             // per-row constants of the sampler's fast path (1 / ||b~_i||^2, SampleZ tables)
  double* dBpiv = nullptr;            // b~_i on the pivot columns, fragment order
  int8_t* dB8 = nullptr;              // two digit planes of the basis, transposed, dpad x dpad each
  bool basis_hi = false, basis_generic = false;
  // per batch
  size_t bcap = 0, ld = 0;
  double* dTm = nullptr;              // running projections, dpad x ld
  double* dZf = nullptr;              // z as f64, chunk stream (ld / 128) x nkb
  int8_t* dZ8 = nullptr; size_t zplane = 0;          // three digit planes of z, [dpad / 16][ld][16] each
  unsigned char* dBocc = nullptr;     // which 128 x 128 tiles of the basis digit planes hold anything: [2][nrb][nrb] (k_np_occ_basis, per key)
  unsigned char* dZocc = nullptr;     // the same for z digits 1 and 2: [2][ld / 128][nrb] (k_np_occ_z, per call)
  int np_combine = 1;                 // PSF_NP_COMBINE: 1 / unset = one fused launch over the occupied tiles (k_np_combine8_fused), 0 = one launch per digit pair
  double* dC0p = nullptr;             // -sol on the pivots, chunk stream (ld / 128) x nkc
  uint64_t* dSol = nullptr;           // n x ld
  int* dFlags = nullptr;              // [0] sampler failure [1] second digit of some z in use [2] third digit [3] |z| beyond three digits; [4..7]: the same for the second pass
  // two-pass walk for large moduli (q sqrt(n) > 2^13 s): the first pass only finds a short coset representative e1, the second samples around it
  bool two_pass = false;
  size_t nkd = 0;                     // K chunks of all d coordinates
  double* dBfull = nullptr;           // b~_i on every coordinate, fragment order (A operand of the second projection)
  double* dC1 = nullptr;              // -e1, chunk stream (ld / 128) x nkd
  int64_t* dE1 = nullptr;             // e1, bcap x dim
  unsigned* dWalk = nullptr;          // k_np_walk: [group][block] counters of published z | [group][block] flags of completed rows | abort word
  size_t walk_words = 0;
  int np_walk = -1;                   // PSF_NP_WALK: 0 = one launch per block (k_np_step); 1 / unset = the whole walk in one launch (k_np_walk: updater workgroups, T in
                                      // registers) where it fits, launches otherwise; 3 = k_np_walk2 (helper waves update T in memory) for every batch that is resident
  int cus = 0;                        // compute units of the device
  int walk_slots[3] = {0, 0, 0};      // [G]: workgroups of k_np_walk<G> a compute unit holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor)
  unsigned walk_spins = 1u << 22;      // NpWalkSync::spin_limit
  int last_form = 0, last_G = 0;      // what the last call launched for the walk: 1 = k_np_walk<G> (one launch), 0 = k_np_step<G> per block, 3 = k_np_walk2<G>
  int last_parts = 1;                 // column ranges the last call walked side by side (np_split)
  int split_delay_us = 0;             // the second half starts this much behind the first
  int np_split = 0;                   // experiments build: 0 never, 1 whenever the shape allows, 2 for large batches (>= 3072) only
  hipStream_t sh[2] = {nullptr, nullptr};                  // the two halves of a large batch walk side by side on these (equal priority, non-blocking)
  hipEvent_t evFork = nullptr, evHalf[2] = {nullptr, nullptr};
  int np_g = 0;                       // PSF_NP_G: preimages per wave of the sampler (0 = by batch size)
  int np_immediate = -1;              // PSF_NP_IMMEDIATE: 1 = every block updates all the rows below it in the launch that follows, 0 = panel-deferred far update, -1 = by batch size
  bool has_key = false;
  bool timing = false;
  bool last_generic = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  hipStream_t last_stream = nullptr;
};

// Gram-Schmidt of the rows of dSt into dGt (MatQ::gso, gpv.rs:91), then ||b~_i||^2, the per-step sampler tables and the
// operands of the blocked nearest plane
static psf_status gpv_finish_basis(psfgpv_handle* g, bool compute_gso) {
  const size_t d = g->dim;
  KeygenClock kc("psfgpv");
  if (compute_gso) {
    hipLaunchKernelGGL(k_i32_to_f64, dim3(grid_for(d * d)), dim3(256), 0, 0, g->dSt, g->dGt, d * d);
    // blocked Gram-Schmidt with re-orthogonalisation on the FP64 matrix cores (psf_gemm_kernels.hpp)
    int* dinfo = nullptr;
    HIP_TRY(hipMalloc(&dinfo, sizeof(int)));
    HIP_TRY(hipMemset(dinfo, 0, sizeof(int)));
    const hipError_t ge = gso_blocked(nullptr, g->dGt, d, d, dinfo);
    int info = 0;
    if (ge == hipSuccess) hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost);
    hipFree(dinfo);
    if (ge != hipSuccess) return PSF_ERR_HIP;
    if (info != 0) return PSF_ERR_PARAM;                                 // linearly dependent "basis"
    kc.mark("  gso_blocked");
  }
  hipLaunchKernelGGL(k_row_norm2_chain, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, 0, g->dGt, d, g->dNorm2);
  HIP_TRY(hipGetLastError());
  std::vector<double> norm2(d);
  HIP_TRY(hipMemcpy(norm2.data(), g->dNorm2, d * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<SampleZParams> sz(d);
  for (size_t i = 0; i < d; ++i) {
    if (!(norm2[i] > 0.0)) return PSF_ERR_PARAM;                       // linearly dependent "basis"
    sz[i] = make_sample_z_params(g->s / std::sqrt(norm2[i]));
  }
  HIP_TRY(hipMemcpy(g->dSz, sz.data(), d * sizeof(SampleZParams), hipMemcpyHostToDevice));
  {
    std::vector<NpRow> rows(g->nblk * NP_NB, NpRow{0.0, 0.f, 0, 1, 0, 0, 16});
    for (size_t i = 0; i < d; ++i) {
      const SampleZParams& p = sz[i];
      const bool fast = p.n_int < (1u << 24) && p.c6 < (1ll << 24);     // candidate indices exact in fp32 (np_screen)
      rows[i] = NpRow{1.0 / norm2[i], (float)(p.inv_s * 2.1289340388624525), fast ? (int32_t)p.c6 : 0, p.n_int, p.thr_int, p.thr_frac, fast ? p.sh : 0u};
    }
    HIP_TRY(hipMemcpy(g->dRows, rows.data(), rows.size() * sizeof(NpRow), hipMemcpyHostToDevice));
  }
  kc.mark("  norms, SampleZ rows");
  {  // g[j][i] = <b_j, b~_i>, then its two packed forms
    double* dGd = nullptr;
    HIP_TRY(hipMalloc(&dGd, d * d * sizeof(double)));
    const unsigned tiles = (unsigned)((d + 63) / 64);
    hipLaunchKernelGGL(k_np_gram, dim3(tiles, tiles), dim3(256), 0, 0, g->dSt, g->dGt, d, dGd);
    if (np_panel_base(g->nblk))
      hipLaunchKernelGGL(k_np_pack_panels, dim3(grid_for(np_panel_base(g->nblk) * TR_CHUNK, 256, 256 * 64)), dim3(256), 0, 0, dGd, d, g->nblk, g->dGp);
    hipLaunchKernelGGL(k_np_pack_inblock, dim3(grid_for(g->nblk * NP_TRI)), dim3(256), 0, 0, dGd, d, g->nblk, g->dGin);
    hipLaunchKernelGGL(k_np_pack_next, dim3(grid_for(g->nblk * NP_NB * NP_NB)), dim3(256), 0, 0, dGd, d, g->nblk, g->dGnx);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    hipFree(dGd);
  }
  kc.mark("  gram + packing");
  {
    int* dinfo = nullptr;
    HIP_TRY(hipMalloc(&dinfo, 2 * sizeof(int)));
    HIP_TRY(hipMemset(dinfo, 0, 2 * sizeof(int)));
    hipLaunchKernelGGL(k_np_pack_basis8, dim3(grid_for(g->dpad * g->dpad, 256, 256 * 64)), dim3(256), 0, 0, g->dSt, d, g->dpad, g->dB8, dinfo);
    int info[2] = {0, 0};
    HIP_TRY(hipMemcpy(info, dinfo, sizeof(info), hipMemcpyDeviceToHost));
    hipFree(dinfo);
    g->basis_generic = info[0] != 0;
    g->basis_hi = info[1] != 0;
    hipLaunchKernelGGL(k_np_occ_basis, dim3((unsigned)(g->nrb * g->nrb), 2), dim3(256), 0, 0, g->dB8, g->dpad, (int)g->nrb, g->dBocc);
    HIP_TRY(hipGetLastError());
  }
  return PSF_OK;
}

static psf_status gpv_build_solver(psfgpv_handle* g) {
  psfp_handle* b = g->base;
  std::vector<uint64_t> A(b->n * b->m);
  HIP_TRY(hipMemcpy(A.data(), b->dA, A.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  std::vector<uint32_t> piv;
  std::vector<uint64_t> T;
  const psf_status rc = solve_precompute(A.data(), b->n, b->m, b->q, piv, T);
  if (rc != PSF_OK) return rc;                                         // gpv.rs:153-155: solve(...).unwrap() would panic
  {
    // The elimination finds its pivot columns row by row, and over a power-of-two modulus (a unit is needed) not always in ascending order.  The initial projection
    // <c0, b~_i> is an fma chain over the pivot coordinates, and the contract's chain runs in ascending COORDINATE order (the oracle walks all m columns and skips the
    // zeros): the pivots are therefore sorted, each with its row of the solve operator.  Unsorted, the two chains would differ in their last bits -- invisible while
    // |c0| is small, a different sample at every step once |t| reaches 2^60.
    const size_t nn = b->n;
    std::vector<size_t> order(nn);
    for (size_t r = 0; r < nn; ++r) order[r] = r;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return piv[x] < piv[y]; });
    std::vector<uint32_t> piv2(nn);
    std::vector<uint64_t> T2(T.size());
    for (size_t r = 0; r < nn; ++r) {
      piv2[r] = piv[order[r]];
      for (size_t t = 0; t < nn; ++t) T2[r * nn + t] = T[order[r] * nn + t];
    }
    piv.swap(piv2);
    T.swap(T2);
  }
  {  // stored transposed: lane r of the solve kernel reads Tt[t][r], consecutive lanes consecutive addresses
    const size_t nn = b->n;
    std::vector<uint64_t> Tt(T.size());
    for (size_t r = 0; r < nn; ++r)
      for (size_t t = 0; t < nn; ++t) Tt[t * nn + r] = T[r * nn + t];
    HIP_TRY(hipMemcpy(g->dT, Tt.data(), Tt.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy(g->dPiv, piv.data(), piv.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  // b~_i restricted to the pivot columns: the A operand of the initial projection T = B~ C0 (c0 = -sol is zero elsewhere)
  hipLaunchKernelGGL(k_np_pack_bpiv, dim3(grid_for(g->nrb * g->nkc * TR_CHUNK, 256, 256 * 64)), dim3(256), 0, 0, g->dGt, g->dPiv, g->dim, g->n, g->nrb, g->nkc, g->dBpiv);
  if (g->two_pass)
    hipLaunchKernelGGL(k_np_pack_bpiv, dim3(grid_for(g->nrb * g->nkd * TR_CHUNK, 256, 256 * 64)), dim3(256), 0, 0, g->dGt, (const uint32_t*)nullptr, g->dim, g->dim, g->nrb, g->nkd, g->dBfull);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return PSF_OK;
}

static void free_np_batch(psfgpv_handle* g) {
  hipFree(g->dTm); hipFree(g->dZf); hipFree(g->dZ8); hipFree(g->dC0p); hipFree(g->dSol); hipFree(g->dC1); hipFree(g->dE1); hipFree(g->dWalk); hipFree(g->dZocc);
  g->dZocc = nullptr;
  g->dTm = g->dZf = g->dC0p = g->dC1 = nullptr; g->dZ8 = nullptr; g->dSol = nullptr; g->dE1 = nullptr; g->dWalk = nullptr;
  g->bcap = 0;
}
static psf_status ensure_np_batch(psfgpv_handle* g, size_t B) {
  if (B <= g->bcap) return PSF_OK;
  HIP_TRY(hipDeviceSynchronize());
  free_np_batch(g);
  const size_t ld = round_up(B, TR_BN);
  g->ld = ld;
  HIP_TRY(hipMalloc(&g->dTm, g->dpad * ld * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dZf, ld * g->nkb * 16 * sizeof(double)));
  g->zplane = g->dpad * ld;
  HIP_TRY(hipMalloc(&g->dZ8, 3 * g->zplane));
  HIP_TRY(hipMalloc(&g->dZocc, 2 * (ld / 128) * g->nrb));
  HIP_TRY(hipMalloc(&g->dC0p, ld * g->nkc * 16 * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dSol, g->n * ld * sizeof(uint64_t)));
  if (g->two_pass) {
    HIP_TRY(hipMalloc(&g->dC1, ld * g->nkd * 16 * sizeof(double)));
    HIP_TRY(hipMalloc(&g->dE1, B * g->dim * sizeof(int64_t)));
  }
  g->walk_words = round_up(2 * (ld / NP_GW) * g->nblk + 4, 4);
  HIP_TRY(hipMalloc(&g->dWalk, g->walk_words * sizeof(unsigned)));
  HIP_TRY(hipMemset(g->dTm, 0, g->dpad * ld * sizeof(double)));
  HIP_TRY(hipMemset(g->dZf, 0, ld * g->nkb * 16 * sizeof(double)));      // padding rows / columns of the operands stay zero for good
  HIP_TRY(hipMemset(g->dZ8, 0, 3 * g->zplane));
  HIP_TRY(hipDeviceSynchronize());      // the clears run on the null stream, the walk possibly on a non-blocking one (see ensure_batch, psfp.hip)
  g->bcap = B;
  return PSF_OK;
}

static psf_status launch_np_recombination(psfgpv_handle* g, hipStream_t st, size_t B, int64_t* d_e, int pass, size_t col0 = 0);

// One one-launch walk at a time per device and process: the constructor makes `st` wait for the walk launched before (any handle, any stream), the destructor
// records the event the next one will wait for.  Held across the launch only; costs one hipStreamWaitEvent + one hipEventRecord per call.
struct WalkTurn {
  static constexpr int MAX_DEV = 64;
  static std::mutex& mu() { static std::mutex m; return m; }
  static hipEvent_t* events() { static hipEvent_t ev[MAX_DEV] = {}; return ev; }
  static bool* recorded() { static bool r[MAX_DEV] = {}; return r; }
  std::unique_lock<std::mutex> lk;
  int dev; hipStream_t st;
  WalkTurn(int device, hipStream_t s) : lk(mu()), dev(device), st(s) {
    if (dev < 0 || dev >= MAX_DEV) { dev = -1; return; }
    if (!events()[dev] && hipEventCreateWithFlags(&events()[dev], hipEventDisableTiming) != hipSuccess) { events()[dev] = nullptr; dev = -1; return; }
    if (recorded()[dev]) hipStreamWaitEvent(st, events()[dev], 0);
  }
  ~WalkTurn() {
    if (dev >= 0 && hipEventRecord(events()[dev], st) == hipSuccess) recorded()[dev] = true;
  }
};

// The per-batch buffers of the walk seen from preimage column `col0` (a multiple of 128) on: every one of them is laid out by columns or by 128-column blocks, so a
// column range of the batch is the same launch sequence on offset pointers -- what lets two halves of a large batch walk side by side (gpv_samp_p_enqueue).
struct NpCols {
  double* Tm; double* Zf; int8_t* Z8; unsigned char* Zocc; double* C0p; double* C1; uint64_t* Sol; int64_t* E1;
};
static NpCols np_cols(const psfgpv_handle* g, size_t col0) {
  const size_t cb = col0 / TR_BN;
  return NpCols{g->dTm + col0, g->dZf + cb * g->nkb * TR_CHUNK, g->dZ8 + col0 * 16, g->dZocc + cb * g->nrb, g->dC0p + cb * g->nkc * TR_CHUNK,
                g->dC1 ? g->dC1 + cb * g->nkd * TR_CHUNK : nullptr, g->dSol + col0, g->dE1 ? g->dE1 + col0 * g->dim : nullptr};
}

// does a batch of B preimages fit the one-launch walk?  (one sampler workgroup per CU at most beside the updaters that hold d/64 - 2 row blocks per column group)
static bool np_walk_fits(const psfgpv_handle* g, size_t B, int* Gw_out, unsigned* nSw_out, unsigned* ngroups_out, unsigned* ug_out) {
  if (g->np_walk == 0 || g->np_walk == 3 || g->cus <= 0 || g->nblk < 3) return false;      // (PSF_NP_WALK=3: k_np_walk2 for every batch, tests)
  // One preimage per sampler wave only: with two (k_np_walk<2>, 1025 ... 2048 preimages where it fits: 124 bytes of scratch per lane) the walk measured SLOWER than the
  // launch-per-block form with one preimage per wave (C4 shape, round 6, tools/np_batch_sweep.py: 2.71 / 2.82 / 2.87 ms at 1025 / 1536 / 1792 preimages against 2.14 / 2.40 /
  // 2.59); the experiments build keeps it behind PSF_NP_G=2
  if (!psf_experiments_build && g->np_g != 1 && B > 4 * (size_t)g->cus) return false;
  const int Gw = (g->np_g == 1 || g->np_g == 2) ? g->np_g : (B <= 4 * (size_t)g->cus ? 1 : 2);
  const unsigned nSw = (unsigned)((B + 4 * (size_t)Gw - 1) / (4 * (size_t)Gw));
  const unsigned per = (unsigned)(NP_GW / (4 * Gw)), ngroups = (nSw + per - 1) / per;
  // residency: the device holds walk_slots workgroups of this kernel per CU (occupancy API: registers and LDS, not an assumption about the chip being ours alone --
  // that part no API can promise, hence k_np_walk_solo); one sampler workgroup per CU at most, the updaters take what the samplers leave
  const long slots = (long)g->cus * (long)g->walk_slots[Gw];
  unsigned ug = 0;
  if (ngroups && nSw <= (unsigned)g->cus && slots > (long)nSw) ug = (unsigned)std::min<long>((slots - (long)nSw) / (long)ngroups, (long)g->cus / (long)ngroups);
  if (ug > g->nblk - 2) ug = (unsigned)(g->nblk - 2);
  if (!(nSw <= (unsigned)g->cus && ug >= 1 && (size_t)2 * NP_WALK_SLOTS * ug >= g->nblk - 2)) return false;
  if (Gw_out) { *Gw_out = Gw; *nSw_out = nSw; *ngroups_out = ngroups; *ug_out = ug; }
  return true;
}

// MatZ::sample_d_precomputed_gso for B preimages (gpv.rs:160), the columns col0 ... of the batch buffers: the launch sequence of psf_np_kernels.hpp on one stream
// pass 0: centre -sol on the pivot columns (K = n), e = sum z b + sol; pass 1 (two-pass mode): centre -e1 on every coordinate (K = d), e = sum z b + e1
// first_index / d_e: of the range's first preimage
static psf_status launch_nearest_plane(psfgpv_handle* g, hipStream_t st, uint64_t seed, uint32_t tag, uint64_t first_index, size_t B, int64_t* d_e, int pass = 0, size_t col0 = 0,
                                       bool may_walk = true) {
  const size_t ld = g->ld, nbj = round_up(B, TR_BN) / TR_BN;
  const size_t lds_gemm = 4 * TR_CHUNK * sizeof(double);
  const NpCols v = np_cols(g, col0);
  // T = B~[:, pivots] C0[pivots]
  int* const flags = g->dFlags + 4 * pass;
  if (pass == 0) hipLaunchKernelGGL(k_np_project, dim3((unsigned)nbj, (unsigned)g->nrb), dim3(256), lds_gemm, st, g->dBpiv, g->nkc, v.C0p, g->nkc, (int)g->nkc, v.Tm, ld);
  else hipLaunchKernelGGL(k_np_project, dim3((unsigned)nbj, (unsigned)g->nrb), dim3(256), lds_gemm, st, g->dBfull, g->nkd, v.C1, g->nkd, (int)g->nkd, v.Tm, ld);
  int G = g->np_g;
  if (G != 1 && G != 2) G = B <= 2048 ? 1 : 2;      // one wave pair per preimage while the sampler workgroups fit the chip's 512 slots at once (C4 shape: 2.45 / 2.80 ms at 1537 /
                                                    // 2048 preimages against 3.06 / 3.25 with two preimages per pair); two rounds of them lose to two preimages per pair
  NpSampleArgs a{v.Tm, ld, g->dGin, g->dGnx, g->dRows, g->dSz, v.Zf, g->nkb, v.Z8, g->zplane, ld, flags};
  // The whole walk in one launch (k_np_walk) where every workgroup can be resident at once: one sampler workgroup per CU at most (B <= 4 G CUs) beside one
  // updater workgroup per CU, and at most 2 * NP_WALK_SLOTS blocks of T per updater.  Otherwise one launch per block (k_np_step).
  {
    int Gw = 0; unsigned nSw = 0, ngroups = 0, ug = 0;
    if (may_walk && col0 == 0 && np_walk_fits(g, B, &Gw, &nSw, &ngroups, &ug)) {
      NpWalkSync sy{g->dWalk, g->dWalk + (size_t)ngroups * g->nblk, g->dWalk + (size_t)2 * ngroups * g->nblk, (unsigned)g->nblk, g->walk_spins};
      NpSampleArgs aw{g->dTm, ld, g->dGin, g->dGnx, g->dRows, g->dSz, g->dZf, g->nkb, g->dZ8, g->zplane, ld, flags};
      const unsigned ntot = nSw + ngroups * ug;
      unsigned long long* reruns = reinterpret_cast<unsigned long long*>(g->dFlags + 8);
      {
        // two walks at once on one device (two handles, two streams) could each hold half of the slots and wait for the other half for ever: walks of this process take
        // turns per device -- each launch waits for the event behind the previous one.  (Another process is not covered by this; k_np_walk_solo is.)
        WalkTurn turn(g->base->prm.device, st);
        hipMemsetAsync(g->dWalk, 0, g->walk_words * sizeof(unsigned), st);
        if (Gw == 1) hipLaunchKernelGGL((k_np_walk<1>), dim3(ntot), dim3(512), 65536, st, aw, g->dim, g->nblk, seed, tag, first_index, B, nSw, ngroups, ug, g->dGp, g->dTm, sy);
#ifdef PSF_EXPERIMENTS
        else hipLaunchKernelGGL((k_np_walk<2>), dim3(ntot), dim3(512), 65536, st, aw, g->dim, g->nblk, seed, tag, first_index, B, nSw, ngroups, ug, g->dGp, g->dTm, sy);
#endif
      }
      // a walk that gave up (the abort word) is walked again without waits between workgroups, from a fresh projection; both launches return at once otherwise
      if (pass == 0) hipLaunchKernelGGL(k_np_project, dim3((unsigned)nbj, (unsigned)g->nrb), dim3(256), lds_gemm, st, g->dBpiv, g->nkc, g->dC0p, g->nkc, (int)g->nkc, g->dTm, ld, (const unsigned*)sy.abort);
      else hipLaunchKernelGGL(k_np_project, dim3((unsigned)nbj, (unsigned)g->nrb), dim3(256), lds_gemm, st, g->dBfull, g->nkd, g->dC1, g->nkd, (int)g->nkd, g->dTm, ld, (const unsigned*)sy.abort);
      if (Gw == 1) hipLaunchKernelGGL((k_np_walk_solo<1>), dim3(nSw), dim3(512), 65536, st, aw, g->dim, g->nblk, seed, tag, first_index, B, g->dGp, g->dTm, (const unsigned*)sy.abort, reruns);
#ifdef PSF_EXPERIMENTS
      else hipLaunchKernelGGL((k_np_walk_solo<2>), dim3(nSw), dim3(512), 65536, st, aw, g->dim, g->nblk, seed, tag, first_index, B, g->dGp, g->dTm, (const unsigned*)sy.abort, reruns);
#endif
      g->last_form = 1; g->last_G = Gw;
      return launch_np_recombination(g, st, B, d_e, pass, col0);
    }
  }
  const unsigned nS = (unsigned)((B + 4 * (size_t)G - 1) / (4 * (size_t)G));
  // The sampler workgroups alone fill the chip (two per CU): the walk in one launch with the helper waves updating T in memory (k_np_walk2)
  // MEASURED SLOWER than one launch per block at C4 (6.9 against 4.4 ms, profiles/r05_notes.md: with two wave pairs per SIMD the vector pipe is the bound, and the
  // helpers' tiles re-read the block's z per 16 rows): a labelled opt-in (PSF_NP_WALK=3), kept bit-identical by tests/test_gpu_switch_matrix.py
#ifdef PSF_EXPERIMENTS
  if (g->np_walk == 3 && col0 == 0 && g->cus > 0 && g->nblk >= 3 && nS <= 2u * (unsigned)g->cus) {
    const unsigned per = (unsigned)(NP_GW / (4 * G)), ngroups = (nS + per - 1) / per;
    NpWalkSync sy{g->dWalk, g->dWalk + (size_t)ngroups * g->nblk, g->dWalk + (size_t)2 * ngroups * g->nblk, (unsigned)g->nblk, g->walk_spins};
    hipMemsetAsync(g->dWalk, 0, g->walk_words * sizeof(unsigned), st);
    if (G == 1) hipLaunchKernelGGL((k_np_walk2<1>), dim3(nS), dim3(512), 65536, st, a, g->dim, g->nblk, seed, tag, first_index, B, nS, g->dGp, g->dTm, sy);
    else hipLaunchKernelGGL((k_np_walk2<2>), dim3(nS), dim3(512), 65536, st, a, g->dim, g->nblk, seed, tag, first_index, B, nS, g->dGp, g->dTm, sy);
    g->last_form = 3; g->last_G = G;
    return launch_np_recombination(g, st, B, d_e, pass, col0);
  }
#endif
  const size_t W = NP_PANEL;
  // small batches: a rank-64 update of every row below costs less than the sampler's 64 steps, and the T matrix stays in the Infinity Cache (measured
  // at C2, 1024 preimages: 4.57 vs 4.80 ms); large batches: the panel-deferred update moves T an eighth as often (C4, 4096 preimages: 5.06 vs 5.20 ms)
  const bool immediate = g->np_immediate >= 0 ? g->np_immediate != 0 : B <= 2048;
  for (size_t J = g->nblk; J-- > 0;) {
    NpStepJobs jobs;
    for (int q = 0; q < 3; ++q) { jobs.job[q] = NpUpdateJob{0, 0, 0, 0, 0, 0}; jobs.ntiles[q] = 0; }
    auto set_job = [&](int q, size_t J_first, size_t nsub, size_t rb0, size_t rb1, size_t lo, size_t hi) {
      if (rb1 <= rb0 || hi <= lo) return;
      jobs.job[q] = NpUpdateJob{(int)J_first, (int)nsub, (int)rb0, (int)(rb1 - rb0), lo, hi};
      jobs.ntiles[q] = (unsigned)((rb1 - rb0) * nbj);
    };
    // window: block J + 1 into the rows from the start of the panel below its own up to block J (whose rows the sampler updates itself)
    if (J + 1 < g->nblk) {
      const size_t P = (J + 1) / W;
      const size_t lo = (P >= 1 && !immediate) ? (P - 1) * W * NP_NB : 0, hi = J * NP_NB;
      set_job(0, J + 1, 1, lo / 128, (hi + 127) / 128, lo, hi);
    }
    // far: block J belongs to panel Pj; the panel above it, Pj + 1, is complete
    const size_t Pj = J / W, P = Pj + 1;
    if (P * W < g->nblk && P >= 2 && !immediate) {
      const size_t top = std::min(g->nblk, (P + 1) * W) - 1, nsub = top - P * W + 1;
      const size_t near_lo = (P - 2) * W * NP_NB, near_hi = (P - 1) * W * NP_NB;      // rows of panel P - 2
      const size_t i = (P * W - 1) - J;                                               // 0 .. W-1: position of this launch inside panel P - 1
      if (i == 0) set_job(1, top, nsub, near_lo / 128, near_hi / 128, near_lo, near_hi);
      // the rows further down only need it before panel P - 2 starts; measured on MI355X: spreading these tiles over the launches of the panel
      // stretches every one of them (a tile's K loop over the whole panel is latency bound), so they all ride in the first launch as well
      if (i == 0) set_job(2, top, nsub, 0, near_lo / 128, 0, near_lo);
    }
    const unsigned ntot = nS + jobs.ntiles[0] + jobs.ntiles[1] + jobs.ntiles[2];
    if (G == 1) hipLaunchKernelGGL((k_np_step<1>), dim3(ntot), dim3(512), 65536, st, a, g->dim, J, seed, tag, first_index, B, nS, jobs, (int)nbj, g->dGp, v.Tm);
    else hipLaunchKernelGGL((k_np_step<2>), dim3(ntot), dim3(512), 65536, st, a, g->dim, J, seed, tag, first_index, B, nS, jobs, (int)nbj, g->dGp, v.Tm);
  }
  g->last_form = 0; g->last_G = G;
  return launch_np_recombination(g, st, B, d_e, pass, col0);
}

// e = sum_i z_i b_i + sol (pass 1 of the two-pass walk: + e1)
static psf_status launch_np_recombination(psfgpv_handle* g, hipStream_t st, size_t B, int64_t* d_e, int pass, size_t col0) {
  const size_t ld = g->ld;
  const NpCols v = np_cols(g, col0);
  int* const flags = g->dFlags + 4 * pass;
  const psfp_handle* b = g->base;
  const dim3 cgrid((unsigned)((B + 127) / 128), (unsigned)(g->dpad / 128));
  const int nk128 = (int)(g->dpad / 128);
  if (!g->basis_generic && g->np_combine != 0) {
    // z = z0 + 256 z1 + 65536 z2, b = b0 + 256 b1: every digit pair in one launch, over the tiles of the digit planes that hold anything
    hipLaunchKernelGGL(k_np_occ_z, dim3((unsigned)(cgrid.x * g->nrb), 2), dim3(256), 0, st, v.Z8, g->zplane, ld, nk128, v.Zocc);
    hipLaunchKernelGGL(k_np_combine8_fused, cgrid, dim3(256), 65536 + 768, st, g->dB8, g->dpad, g->dim, nk128, g->basis_hi ? 2 : 1, g->dBocc, v.Z8, g->zplane, ld, B, v.Zocc, d_e, g->dim);
    if (pass == 0) hipLaunchKernelGGL(k_np_add_sol, dim3(grid_for(g->n * B)), dim3(256), 0, st, v.Sol, g->dPiv, g->n, B, ld, d_e, g->dim);
  }
#ifdef PSF_EXPERIMENTS
  else if (!g->basis_generic) {
    // the same sum as one pass per digit pair in use (the z digits beyond the first are gated on the device): PSF_NP_COMBINE=0, kept for the switch matrix
    const size_t plane = g->dpad * g->dpad;
    const int8_t* zp[3] = {v.Z8, v.Z8 + g->zplane, v.Z8 + 2 * g->zplane};
    const int* gate[3] = {nullptr, flags + 1, flags + 2};
    bool first = true;
    for (int zi = 0; zi < 3; ++zi)
      for (int bi = 0; bi < (g->basis_hi ? 2 : 1); ++bi) {
        const long long scale = 1ll << (8 * (zi + bi));
        if (first) hipLaunchKernelGGL((k_np_combine8<false>), cgrid, dim3(256), 65536, st, g->dB8 + bi * plane, g->dpad, g->dim, nk128, zp[zi], ld, B, scale, gate[zi], d_e, g->dim);
        else hipLaunchKernelGGL((k_np_combine8<true>), cgrid, dim3(256), 65536, st, g->dB8 + bi * plane, g->dpad, g->dim, nk128, zp[zi], ld, B, scale, gate[zi], d_e, g->dim);
        first = false;
      }
    if (pass == 0) hipLaunchKernelGGL(k_np_add_sol, dim3(grid_for(g->n * B)), dim3(256), 0, st, v.Sol, g->dPiv, g->n, B, ld, d_e, g->dim);
  }
#endif
  // integer fallback: always for a basis beyond two int8 digits, otherwise only if a z left the three-digit range (decided on the device)
  hipLaunchKernelGGL(k_np_combine_generic, dim3(grid_for(g->dim * B, 256, 256 * 64)), dim3(256), 0, st, g->dSt, g->dim, v.Zf, g->nkb, v.Sol, g->dPiv, pass == 0 ? g->n : (size_t)0, B, ld,
                     g->basis_generic ? (const int*)nullptr : (const int*)(flags + 3), d_e, g->dim);
  if (pass == 1) hipLaunchKernelGGL(k_np_add_e1, dim3(grid_for(g->dim * B, 256, 256 * 64)), dim3(256), 0, st, v.E1, g->dim * B, d_e);
  (void)b;
  return PSF_OK;
}

#ifdef NP_PROFILE
extern "C" void psf_debug_np_prof(long long* out, int reset) {
  if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_np_prof), sizeof(long long) * 8);
  if (reset) { long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_np_prof), z, sizeof(z)); }
}
extern "C" void psf_debug_np_spread(unsigned long long* sum, unsigned long long* mx, int reset) {
  if (sum) hipMemcpyFromSymbol(sum, HIP_SYMBOL(g_np_sum), sizeof(unsigned long long) * 8);
  if (mx) hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_np_max), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_np_sum), z, sizeof(z)); hipMemcpyToSymbol(HIP_SYMBOL(g_np_max), z, sizeof(z)); }
}
#if NP_PROFILE == 3
extern "C" void psf_debug_np_single(unsigned long long* out, int reset) {
  if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_np_single), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_np_single), z, sizeof(z)); }
}
#endif
extern "C" void psf_debug_np_events(unsigned long long* out, int reset) {
  if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_np_events), sizeof(unsigned long long) * 4);
  if (reset) { unsigned long long z[4] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_np_events), z, sizeof(z)); }
}
#endif

extern "C" {

static psf_status psfgpv_init(psfgpv_handle* g);

psf_status psfgpv_create(const psfgpv_params* prm, psfgpv_handle** out) {
  if (!prm || !out || !(prm->s > 0.0)) return PSF_ERR_PARAM;
  psfp_params bp;
  bp.gp = prm->gp; bp.r = 1.0; bp.s = prm->s; bp.device = prm->device; bp.flags = PSFP_FLAG_NO_PERTURB;
  psfp_handle* b = nullptr;
  const psf_status rc = psfp_create(&bp, &b);
  if (rc != PSF_OK) return rc;
  psfgpv_handle* g = new psfgpv_handle();
  g->base = b; g->s = prm->s; g->n = b->n; g->m = b->m; g->dim = b->m;
  const psf_status rc2 = psfgpv_init(g);
  if (rc2 != PSF_OK) { psfgpv_destroy(g); return rc2; }     // releases the partial allocations and the inner handle
  *out = g;
  return PSF_OK;
}

static psf_status psfgpv_init(psfgpv_handle* g) {
  const size_t d = g->dim;
  g->nblk = (d + NP_NB - 1) / NP_NB;
  g->dpad = round_up(d, 128);
  g->nrb = g->dpad / 128;
  g->nkc = round_up(g->n, 16) / 16;
  g->nkb = round_up(d, NP_NB) / 16;
  HIP_TRY(hipMalloc(&g->dSt, d * d * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&g->dGt, d * d * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dNorm2, d * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dSz, d * sizeof(SampleZParams)));
  HIP_TRY(hipMalloc(&g->dT, g->n * g->n * sizeof(uint64_t)));
  HIP_TRY(hipMalloc(&g->dPiv, g->n * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&g->dGp, (np_panel_base(g->nblk) + 1) * TR_CHUNK * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dGin, g->nblk * NP_TRI * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dGnx, g->nblk * NP_NB * NP_NB * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dRows, g->nblk * NP_NB * sizeof(NpRow)));
  HIP_TRY(hipMalloc(&g->dBpiv, g->nrb * g->nkc * TR_CHUNK * sizeof(double)));
  HIP_TRY(hipMalloc(&g->dB8, 2 * g->dpad * g->dpad));
  HIP_TRY(hipMalloc(&g->dBocc, 2 * g->nrb * g->nrb));
  HIP_TRY(hipMalloc(&g->dFlags, 12 * sizeof(int)));                       // [0..7] per call (two passes), [8..9] a 64-bit count of walks re-run by k_np_walk_solo (never cleared)
  HIP_TRY(hipMemset(g->dFlags, 0, 12 * sizeof(int)));
  // large moduli: q sqrt(n) > 2^13 s (relative centre error of a single pass above 2^-40, see include/psf_mi355x.h "Precision of the centres"); PSF_NP_TWO_PASS=0/1 forces
  g->two_pass = (double)g->base->q * std::sqrt((double)g->n) > g->s * 8192.0;
  { const char* ev = psf_exp_env("PSF_NP_TWO_PASS"); if (ev) g->two_pass = atoi(ev) != 0; }
  g->nkd = round_up(d, 16) / 16;
  if (g->two_pass) HIP_TRY(hipMalloc(&g->dBfull, g->nrb * g->nkd * TR_CHUNK * sizeof(double)));
  { const char* ev = psf_exp_env("PSF_NP_G"); g->np_g = ev ? atoi(ev) : 0; }
  { const char* ev = psf_exp_env("PSF_NP_IMMEDIATE"); g->np_immediate = ev ? (atoi(ev) != 0 ? 1 : 0) : -1; }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_project), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TR_CHUNK * sizeof(double)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_step<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_step<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_walk<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#ifdef PSF_EXPERIMENTS
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_walk<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#endif
#ifdef PSF_EXPERIMENTS
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_walk2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_walk2<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_combine8<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_combine8<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#endif
  { int cu = 0; HIP_TRY(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, g->base->prm.device)); g->cus = cu; }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_walk_solo<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#ifdef PSF_EXPERIMENTS
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_walk_solo<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#endif
  // workgroups of the one-launch walk a compute unit really holds (registers, LDS): the residency test of launch_nearest_plane multiplies by the CU count
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&g->walk_slots[1], reinterpret_cast<const void*>(k_np_walk<1>), 512, 65536));
#ifdef PSF_EXPERIMENTS
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&g->walk_slots[2], reinterpret_cast<const void*>(k_np_walk<2>), 512, 65536));
#endif
  if (const char* e = psf_exp_env("PSF_NP_WALK")) g->np_walk = std::atoi(e);
  if (const char* e = psf_exp_env("PSF_NP_WALK_SPINS")) { const long v = std::atol(e); if (v >= 1) g->walk_spins = (unsigned)v; }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_np_combine8_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 768));
  if (const char* e = psf_exp_env("PSF_NP_COMBINE")) g->np_combine = std::atoi(e);
  for (auto& e : g->ev) HIP_TRY(hipEventCreate(&e));
#ifdef PSF_EXPERIMENTS      /* the two-halves walk (measured neutral: not in the release library) */
  for (auto& sx : g->sh) HIP_TRY(hipStreamCreateWithFlags(&sx, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&g->evFork, hipEventDisableTiming));
  for (auto& e : g->evHalf) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
#endif
  if (const char* e = psf_exp_env("PSF_NP_SPLIT")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) g->np_split = v; }
  if (const char* e = psf_exp_env("PSF_NP_SPLIT_DELAY")) { const int v = std::atoi(e); if (v >= 0 && v <= 1000) g->split_delay_us = v; }
  return PSF_OK;
}

void psfgpv_destroy(psfgpv_handle* g) {
  if (!g) return;
  hipSetDevice(g->base->prm.device);
  free_np_batch(g);
  hipFree(g->dSt); hipFree(g->dGt); hipFree(g->dNorm2); hipFree(g->dSz); hipFree(g->dT); hipFree(g->dPiv);
  hipFree(g->dGp); hipFree(g->dGin); hipFree(g->dGnx); hipFree(g->dRows); hipFree(g->dBpiv); hipFree(g->dB8); hipFree(g->dBocc); hipFree(g->dFlags); hipFree(g->dBfull);
  for (auto& e : g->ev) if (e) hipEventDestroy(e);
  for (auto& sx : g->sh) if (sx) hipStreamDestroy(sx);
  if (g->evFork) hipEventDestroy(g->evFork);
  for (auto& e : g->evHalf) if (e) hipEventDestroy(e);
  psfp_destroy(g->base);
  delete g;
}

size_t psfgpv_m(const psfgpv_handle* g) { return g ? g->m : 0; }

// gpv.rs:83-94
psf_status psfgpv_trap_gen(psfgpv_handle* g, uint64_t seed) {
  if (!g) return PSF_ERR_PARAM;
  psfp_handle* b = g->base;
  HIP_TRY(hipSetDevice(b->prm.device));
  PSFP_QUIESCE(b);
  g->has_key = false;
  KeygenClock kc("psfgpv");
  psf_status rc = gen_A_R(b, seed);                                          // :84-88
  if (rc != PSF_OK) return rc;
  kc.mark("A, R");
  // gen_short_basis_for_trapdoor (:90, short_basis_classical.rs:54-110), assembled transposed on the device
  int8_t* dBT = nullptr;
  const size_t ldw = b->ldr;
  HIP_TRY(hipMalloc(&dBT, b->m * ldw));
  const int reversed = is_power_of_base(b->prm.gp.base, b->k, b->q) ? 1 : 0;
  hipLaunchKernelGGL(k_gpv_bottom_t, dim3(grid_for(b->m * ldw, 256, 256 * 64)), dim3(256), 0, 0, b->dA, b->m, (uint32_t)b->n, (uint32_t)b->k, b->mb, b->w,
                     b->q, b->prm.gp.base, b->dSk, reversed, dBT, ldw);
  hipLaunchKernelGGL(k_gpv_basis_t, dim3((unsigned)((b->mb + 63) / 64), (unsigned)((b->m + 63) / 64)), dim3(256), 0, 0, dBT, ldw, b->dR, b->ldr, b->m,
                     b->mb, b->w, g->dSt);
  hipLaunchKernelGGL(k_gpv_basis_t_tail, dim3(grid_for(b->m * b->w, 256, 256 * 64)), dim3(256), 0, 0, dBT, ldw, b->m, b->mb, b->w, g->dSt);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  hipFree(dBT);
  kc.mark("short basis");
  rc = gpv_finish_basis(g, true);                                             // :91 gso
  if (rc != PSF_OK) return rc;
  kc.mark("gso + walk operands");
  rc = gpv_build_solver(g);
  if (rc != PSF_OK) return rc;
  kc.mark("solve operator");
  b->has_key = true; b->has_pub = true;
  g->has_key = true;
  return PSF_OK;
}

psf_status psfgpv_load_key(psfgpv_handle* g, const uint64_t* A, const int32_t* basis_t, const double* gso_t) {
  if (!g || !A || !basis_t || !gso_t) return PSF_ERR_PARAM;
  psfp_handle* b = g->base;
  HIP_TRY(hipSetDevice(b->prm.device));
  PSFP_QUIESCE(b);
  g->has_key = false;
  HIP_TRY(hipMemcpy(b->dA, A, b->n * b->m * sizeof(uint64_t), hipMemcpyHostToDevice));
  split_A(b);
  HIP_TRY(hipMemcpy(g->dSt, basis_t, g->dim * g->dim * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->dGt, gso_t, g->dim * g->dim * sizeof(double), hipMemcpyHostToDevice));
  psf_status rc = gpv_finish_basis(g, false);
  if (rc != PSF_OK) return rc;
  rc = gpv_build_solver(g);
  if (rc != PSF_OK) return rc;
  b->has_key = true; b->has_pub = true;
  g->has_key = true;
  return PSF_OK;
}

psf_status psfgpv_export_key(const psfgpv_handle* g, uint64_t* A, int8_t* R, int32_t* basis_t, double* gso_t) {
  if (!g) return PSF_ERR_PARAM;
  if (!g->has_key) return PSF_ERR_NO_KEY;
  const psfp_handle* b = g->base;
  HIP_TRY(hipSetDevice(b->prm.device));
  if (A) HIP_TRY(hipMemcpy(A, b->dA, b->n * b->m * sizeof(uint64_t), hipMemcpyDeviceToHost));
  if (R) HIP_TRY(hipMemcpy2D(R, b->w, b->dR, b->ldr, b->w, b->mb, hipMemcpyDeviceToHost));
  if (basis_t) HIP_TRY(hipMemcpy(basis_t, g->dSt, g->dim * g->dim * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (gso_t) HIP_TRY(hipMemcpy(gso_t, g->dGt, g->dim * g->dim * sizeof(double), hipMemcpyDeviceToHost));
  return PSF_OK;
}

// gpv.rs:152-161
static psf_status gpv_samp_p_enqueue(psfgpv_handle* g, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_e, void* stream);
psf_status psfgpv_samp_p_dev(psfgpv_handle* g, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_e, void* stream) {
  if (!g || (B && (!d_u || !d_e))) return PSF_ERR_PARAM;
  if (!g->has_key) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(g->base->prm.device));
  PSFP_QUIESCE(g->base);                // the walk's buffers are shared with the asynchronous calls in flight (which enqueue through gpv_samp_p_enqueue themselves)
  return gpv_samp_p_enqueue(g, seed, first_index, B, d_u, d_e, stream);
}
static psf_status gpv_samp_p_enqueue(psfgpv_handle* g, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_e, void* stream) {
  psfp_handle* b = g->base;
  HIP_TRY(hipSetDevice(b->prm.device));
  hipStream_t st = (hipStream_t)stream;
  psf_status rcb = ensure_np_batch(g, B);
  if (rcb != PSF_OK) return rcb;
  HIP_TRY(hipMemsetAsync(b->dFail, 0, 2 * sizeof(int), st));
  HIP_TRY(hipMemsetAsync(g->dFlags, 0, 8 * sizeof(int), st));
  if (g->timing) hipEventRecord(g->ev[0], st);
  // Two halves side by side (EXPERIMENTS build only: measured and not kept).  A batch that does not fit the one-launch walk runs one launch per 64-row block, and in
  // every such launch the FP64-MFMA update tiles wait for the sampler workgroups to leave their slots (59 us of vector work, then 12 us of matrix work at C4).  Cut into
  // two column ranges on two streams the halves could overlap those phases -- they do not: inside one call the halves run in lockstep and an FP64 MFMA holds its SIMD
  // against the other half's samplers; C4 4.416 -> 4.396 ms per call (3 x 60 steps each, tools/c4_split_ab.sh; a phase offset of 10 ... 100 us between the halves
  // changes nothing, tools/c4_delay_sweep.sh).  Two INDEPENDENT handles on two streams do reach 4.05 ms per 4096 preimages (tools/c4_split_probe.py): what
  // overlaps there is one call's solve / projection / recombination (0.4 ms) with the other call's walk, across calls -- not available inside one call, whose
  // result the caller's stream waits for.  profiles/r06_notes.md, "C4".  Rows are those of the undivided call bit for bit (tests/test_gpu_np_forms.py).
  size_t half0 = 0;
#ifdef PSF_EXPERIMENTS
  if (!g->two_pass && g->np_split > 0 && g->sh[0] && B >= 256) {
    const size_t h0 = round_up((B + 1) / 2, TR_BN);
    const bool shape_ok = h0 < B && !np_walk_fits(g, B, nullptr, nullptr, nullptr, nullptr) && !np_walk_fits(g, h0, nullptr, nullptr, nullptr, nullptr) &&
                          !np_walk_fits(g, B - h0, nullptr, nullptr, nullptr, nullptr);
    if (shape_ok && (g->np_split == 1 || B >= 3072)) half0 = h0;
  }
#endif
  // :153-158  sol = A.solve(u), centre = -sol  (the halves solve for their own columns on their own streams)
  auto solve = [&](hipStream_t sx, size_t col0, size_t cnt) {
    const NpCols v = np_cols(g, col0);
    const size_t cols = col0 + cnt == B ? g->ld - col0 : round_up(cnt, TR_BN);      // the last range also rewrites the padding columns of the batch buffers
    const size_t nk16 = g->nkc * 16;
    const dim3 tgrid((unsigned)((cols + 63) / 64), (unsigned)((nk16 + 63) / 64));
    const bool tiled = b->q < (1ull << 24) && g->n <= 65536 && !psf_exp_env("PSF_NP_SOLVE_PLAIN");
    const bool acc32 = tiled && (double)g->n * (double)(b->q - 1) * (double)(b->q - 1) < 4294967296.0;
    if (acc32) hipLaunchKernelGGL((k_np_solve_tiled<true>), tgrid, dim3(256), 0, sx, g->dT, g->n, nk16, b->q, d_u + col0 * g->n, cnt, g->ld, v.Sol, v.C0p, cols);
    else if (tiled) hipLaunchKernelGGL((k_np_solve_tiled<false>), tgrid, dim3(256), 0, sx, g->dT, g->n, nk16, b->q, d_u + col0 * g->n, cnt, g->ld, v.Sol, v.C0p, cols);
    else hipLaunchKernelGGL(k_np_solve, dim3(grid_for(nk16 * cols)), dim3(256), 0, sx, g->dT, g->n, nk16, b->q, b->two64, d_u + col0 * g->n, cnt, g->ld, v.Sol, v.C0p, cols);
  };
  if (!half0) solve(st, 0, B);
  if (g->timing) hipEventRecord(g->ev[1], st);
  // :160  sol + sample_d_precomputed_gso(basis, gso, centre, s)
  psf_status rc;
  if (half0) {
    HIP_TRY(hipEventRecord(g->evFork, st));                                    // what the caller enqueued before the call (u) is complete
    const size_t cnt[2] = {half0, B - half0}, off[2] = {0, half0};
    rc = PSF_OK;
    for (int i = 0; i < 2 && rc == PSF_OK; ++i) {
      HIP_TRY(hipStreamWaitEvent(g->sh[i], g->evFork, 0));
      solve(g->sh[i], off[i], cnt[i]);
      if (i == 1 && g->split_delay_us > 0) hipLaunchKernelGGL(k_np_delay, dim3(1), dim3(64), 0, g->sh[i], (unsigned long long)g->split_delay_us * 100ull);
      rc = launch_nearest_plane(g, g->sh[i], seed, TAG_GPV, first_index + off[i], cnt[i], d_e + off[i] * g->dim, 0, off[i], false);
      HIP_TRY(hipEventRecord(g->evHalf[i], g->sh[i]));
    }
    for (int i = 0; i < 2; ++i) HIP_TRY(hipStreamWaitEvent(st, g->evHalf[i], 0));      // the caller's stream continues behind both halves
    g->last_parts = 2;
  }
  else if (!g->two_pass) { rc = launch_nearest_plane(g, st, seed, TAG_GPV, first_index, B, d_e); g->last_parts = 1; }
  else {
    g->last_parts = 1;
    rc = launch_nearest_plane(g, st, seed, TAG_GPV, first_index, B, g->dE1, 0);          // a short representative e1 of the coset (A e1 = u)
    hipLaunchKernelGGL(k_np_center_from_e, dim3(grid_for(g->nkd * 16 * g->ld, 256, 256 * 64)), dim3(256), 0, st, g->dE1, g->dim, B, g->ld, g->nkd, g->dC1, g->dFlags);
    if (rc == PSF_OK) rc = launch_nearest_plane(g, st, seed, TAG_GPV2, first_index, B, d_e, 1);   // v ~ D_{Lambda, s, -e1}; e = e1 + v
  }
  if (g->timing) hipEventRecord(g->ev[2], st);
  if (rc != PSF_OK) return rc;
  HIP_TRY(hipGetLastError());
  g->last_stream = st;
  b->last_stream = st;
  return PSF_OK;
}

psf_status psfgpv_last_status(psfgpv_handle* g) {
  if (!g) return PSF_ERR_PARAM;
  psf_status rc = psfp_last_status(g->base);              // synchronises the stream of the last call
  if (rc != PSF_OK) return rc;
  int fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIP_TRY(hipMemcpy(fl, g->dFlags, sizeof(fl), hipMemcpyDeviceToHost));
  g->last_generic = g->basis_generic || fl[g->two_pass ? 7 : 3] != 0;
  return (fl[0] || fl[4]) ? PSF_ERR_SAMPLER : PSF_OK;
}

psf_status psfgpv_samp_p(psfgpv_handle* g, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e) {
  if (!g || (B && (!u || !e))) return PSF_ERR_PARAM;
  if (!g->has_key) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(g->base->prm.device));
  if (B * (g->n + g->m) * 8 <= SIO_MAX_BYTES && !psf_exp_env("PSF_HOST_STRAIGHT")) {
    // a small call (the reference's call is one preimage): cached device buffers, u / e / flags through one pinned buffer, one synchronisation
    psfp_handle* h = g->base;
    if (B * g->n > h->sio_du_cap) { hipFree(h->sio_du); h->sio_du = nullptr; h->sio_du_cap = 0; HIP_TRY(hipMalloc(&h->sio_du, B * g->n * sizeof(uint64_t))); h->sio_du_cap = B * g->n; }
    if (B * g->m > h->sio_de_cap) { hipFree(h->sio_de); h->sio_de = nullptr; h->sio_de_cap = 0; HIP_TRY(hipMalloc(&h->sio_de, B * g->m * sizeof(int64_t))); h->sio_de_cap = B * g->m; }
    int fl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    psf_status rc = sio_call(h, B * g->n, B * g->m, u, e, h->sio_du, h->sio_de, h->sets[0].dFail, h->sets[1].dFail, g->dFlags, 8, fl,
                             [&]() { return psfgpv_samp_p_dev(g, seed, first_index, B, h->sio_du, h->sio_de, nullptr); });
    if (rc != PSF_OK) return rc;
    if (fl[0]) return PSF_ERR_SAMPLER;
    g->last_generic = g->basis_generic || fl[1 + (g->two_pass ? 7 : 3)] != 0;
    return (fl[1 + 0] || fl[1 + 4]) ? PSF_ERR_SAMPLER : PSF_OK;
  }
  if (psf_exp_env("PSF_HOST_STRAIGHT")) {                      // the form of rounds 1-3 (comparison arm of the tests)
    uint64_t* du = nullptr; int64_t* de = nullptr;
    HIP_TRY(hipMalloc(&du, B * g->n * sizeof(uint64_t)));
    if (hipMalloc(&de, B * g->m * sizeof(int64_t)) != hipSuccess) { hipFree(du); return PSF_ERR_HIP; }
    psf_status rcs = hipMemcpy(du, u, B * g->n * sizeof(uint64_t), hipMemcpyHostToDevice) == hipSuccess ? PSF_OK : PSF_ERR_HIP;
    if (rcs == PSF_OK) rcs = psfgpv_samp_p_dev(g, seed, first_index, B, du, de, nullptr);
    if (rcs == PSF_OK) rcs = psfgpv_last_status(g);
    if (hipMemcpy(e, de, B * g->m * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess && rcs == PSF_OK) rcs = PSF_ERR_HIP;
    hipFree(du); hipFree(de);
    return rcs;
  }
  // A batch: cached device buffers (no hipMalloc / hipFree per call), u through the pinned buffer, the rows narrowed to int32 on the device (every entry of a preimage
  // fits by far; k_narrow_rows raises a flag otherwise and the int64 rows are copied as before), ONE copy into pinned memory, widened into e by four threads with
  // streaming stores.  The straight form (two pageable copies around two allocations) took 9.75 ms around 4.34 ms of kernels at C2.
  psfp_handle* h = g->base;
  const size_t nu = B * g->n, ne = B * g->m;
  if (nu > h->sio_du_cap) { hipFree(h->sio_du); h->sio_du = nullptr; h->sio_du_cap = 0; HIP_TRY(hipMalloc(&h->sio_du, nu * sizeof(uint64_t))); h->sio_du_cap = nu; }
  if (ne > h->sio_de_cap) { hipFree(h->sio_de); h->sio_de = nullptr; h->sio_de_cap = 0; HIP_TRY(hipMalloc(&h->sio_de, ne * sizeof(int64_t))); h->sio_de_cap = ne; }
  if (ne > h->sio_d32_cap) { hipFree(h->sio_d32); h->sio_d32 = nullptr; h->sio_d32_cap = 0; HIP_TRY(hipMalloc(&h->sio_d32, ne * sizeof(int32_t) + 2 * sizeof(int))); h->sio_d32_cap = ne; }
  const size_t ub = round_up(nu * 8, 64), eb = round_up(ne * 4, 64);
  psf_status rc = sio_ensure(h, ub + eb + 64);
  if (rc != PSF_OK) return rc;
  uint64_t* hu = reinterpret_cast<uint64_t*>(h->sio_pin);
  int32_t* he = reinterpret_cast<int32_t*>(h->sio_pin + ub);
  int* hf = reinterpret_cast<int*>(h->sio_pin + ub + eb);
  int* d_ovf = reinterpret_cast<int*>(h->sio_d32 + ne);                    // overflow word of the narrowing, behind the rows
  std::memcpy(hu, u, nu * 8);
  hipLaunchKernelGGL(k_copy_words, dim3(sio_grid(nu)), dim3(256), 0, nullptr, hu, h->sio_du, nu);
  HIP_TRY(hipMemsetAsync(d_ovf, 0, 2 * sizeof(int), nullptr));
  rc = psfgpv_samp_p_dev(g, seed, first_index, B, h->sio_du, h->sio_de, nullptr);
  if (rc != PSF_OK) { hipStreamSynchronize(nullptr); return rc; }
  hipLaunchKernelGGL(k_narrow_rows, dim3(grid_for(ne / 2 + 1, 256, 256 * 16)), dim3(256), 0, nullptr, h->sio_de, h->sio_d32, ne, d_ovf);
  // flags first (with the overflow word of the narrowing), then the rows in NT pieces, an event behind each: thread i widens piece i as soon as it has landed,
  // while the later pieces are still crossing PCIe
  hipLaunchKernelGGL(k_sio_flags, dim3(1), dim3(64), 0, nullptr, h->sets[0].dFail, h->sets[1].dFail, g->dFlags, 8, hf);
  HIP_TRY(hipMemcpyAsync(hf + 12, d_ovf, sizeof(int), hipMemcpyDeviceToHost, nullptr));
  constexpr int NT = 4;
  if (!h->sio_ev[0]) for (auto& ev : h->sio_ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(h->sio_ev[NT], nullptr));                       // flags and overflow word are in pinned memory
  const size_t per = round_up((ne + NT - 1) / NT, 16);
  for (int i = 0; i < NT; ++i) {
    const size_t b0 = (size_t)i * per, cnt = b0 >= ne ? 0 : (ne - b0 < per ? ne - b0 : per);
    if (cnt) HIP_TRY(hipMemcpyAsync(he + b0, h->sio_d32 + b0, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipEventRecord(h->sio_ev[i], nullptr));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventSynchronize(h->sio_ev[NT]));
  if (hf[12]) {                                                            // an entry beyond 32 bits: the int64 rows, as before
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(e, h->sio_de, ne * sizeof(int64_t), hipMemcpyDeviceToHost));
  } else {
    std::thread th[NT];
    std::atomic<int> bad{0};
    const int device = h->prm.device;
    auto piece = [&, device](int i, bool set_dev) {
      const size_t b0 = (size_t)i * per, cnt = b0 >= ne ? 0 : (ne - b0 < per ? ne - b0 : per);
      if (set_dev && hipSetDevice(device) != hipSuccess) { bad = 1; return; }
      if (hipEventSynchronize(h->sio_ev[i]) != hipSuccess) { bad = 1; return; }
      if (cnt) widen_rows(e + b0, he + b0, cnt);
    };
    int started = 0;
    try {
      for (; started < NT; ++started) th[started] = std::thread(piece, started, true);
    } catch (...) { }
    for (int i = started; i < NT; ++i) piece(i, false);                   // (no thread to be had: this one does the rest)
    for (int i = 0; i < started; ++i) th[i].join();
    if (bad) return PSF_ERR_HIP;
  }
  if (hf[0]) return PSF_ERR_SAMPLER;
  g->last_generic = g->basis_generic || hf[1 + (g->two_pass ? 7 : 3)] != 0;
  return (hf[1 + 0] || hf[1 + 4]) ? PSF_ERR_SAMPLER : PSF_OK;
}

// gpv.rs:152-161 on host buffers without waiting (the machinery of psfp_samp_p_async on the inner handle: int32 narrowing, per-slot pinned rings, chunk transfers by the
// DMA engines, widening workers): returns once the work is enqueued; e is complete when psfgpv_wait returns.  At most two calls in flight; the rows of call i cross
// PCIe while call i + 1 walks.  A row entry beyond 32 bits (the synchronous call copies 64-bit rows then) makes psfgpv_wait return PSF_ERR_UNSUPPORTED.
psf_status psfgpv_samp_p_async(psfgpv_handle* g, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* e) {
  if (!g || (B && (!u || !e))) return PSF_ERR_PARAM;
  if (!g->has_key) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  psfp_handle* h = g->base;
  HIP_TRY(hipSetDevice(h->prm.device));
  if (B > g->bcap) {                                        // ensure_np_batch reallocates the walk's buffers: nothing may be in flight
    const psf_status rw = psfp_wait(h);
    if (rw != PSF_OK) return rw;
    const psf_status rb = ensure_np_batch(g, B);
    if (rb != PSF_OK) return rb;
  }
  return hp_async(h, B, u, e, false, false, g->dFlags, [&](size_t off, size_t cnt, const uint64_t* d_u, int64_t* d_e, hipStream_t cs) -> psf_status {
    return gpv_samp_p_enqueue(g, seed, first_index + off, cnt, d_u, d_e, cs);
  });
}
psf_status psfgpv_wait(psfgpv_handle* g) {
  if (!g) return PSF_ERR_PARAM;
  return psfp_wait(g->base);
}
uint64_t psfgpv_async_next_ticket(const psfgpv_handle* g) { return g ? psfp_async_next_ticket(g->base) : 0; }
psf_status psfgpv_wait_ticket(psfgpv_handle* g, uint64_t ticket) { return g ? psfp_wait_ticket(g->base, ticket) : PSF_ERR_PARAM; }

psf_status psfgpv_samp_d(psfgpv_handle* g, uint64_t seed, uint64_t first_index, size_t B, int64_t* e) {
  return g ? psfp_samp_d(g->base, seed, first_index, B, e) : PSF_ERR_PARAM;
}
psf_status psfgpv_f_a(psfgpv_handle* g, size_t B, const int64_t* e, uint64_t* u) {
  if (!g) return PSF_ERR_PARAM;
  if (!g->has_key) return PSF_ERR_NO_KEY;
  return psfp_f_a(g->base, B, e, u);
}
psf_status psfgpv_f_a_dev(psfgpv_handle* g, size_t B, const int64_t* d_e, uint64_t* d_u, uint8_t* d_ok, void* stream) {
  if (!g) return PSF_ERR_PARAM;
  if (!g->has_key) return PSF_ERR_NO_KEY;
  return psfp_f_a_dev(g->base, B, d_e, d_u, d_ok, stream);
}
psf_status psfgpv_check_domain(psfgpv_handle* g, size_t B, const int64_t* e, size_t len, uint8_t* ok) {
  return g ? psfp_check_domain(g->base, B, e, len, ok) : PSF_ERR_PARAM;
}
psf_status psfgpv_uniform_targets_dev(psfgpv_handle* g, uint64_t seed, uint64_t first_index, size_t B, uint64_t* d_u, void* stream) {
  return g ? psfp_uniform_targets_dev(g->base, seed, first_index, B, d_u, stream) : PSF_ERR_PARAM;
}
psf_status psfgpv_enable_timing(psfgpv_handle* g, int on) {
  if (!g) return PSF_ERR_PARAM;
  g->timing = on != 0;
  return PSF_OK;
}
psf_status psfgpv_get_timing(psfgpv_handle* g, double* solve_ms, double* nearest_plane_ms) {
  if (!g) return PSF_ERR_PARAM;
  HIP_TRY(hipStreamSynchronize(g->last_stream));
  float a = 0.f, c = 0.f;
  if (g->timing) { hipEventElapsedTime(&a, g->ev[0], g->ev[1]); hipEventElapsedTime(&c, g->ev[1], g->ev[2]); }
  if (solve_ms) *solve_ms = a;
  if (nearest_plane_ms) *nearest_plane_ms = c;
  return PSF_OK;
}
int psfgpv_two_pass(const psfgpv_handle* g) { return (g && g->two_pass) ? 1 : 0; }
// debugging / tests: the coefficients z_i the LAST walk of the last call drew for preimage b (d doubles; the second pass's in two-pass mode)
extern "C" psf_status psfgpv_debug_last_z(psfgpv_handle* g, size_t b, double* z_out) {
  if (!g || !z_out || !g->dZf || b >= g->ld) return PSF_ERR_PARAM;
  HIP_TRY(hipSetDevice(g->base->prm.device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<double> zf(g->ld * g->nkb * 16);
  HIP_TRY(hipMemcpy(zf.data(), g->dZf, zf.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < g->dim; ++i) z_out[i] = zf[((b / TR_BN) * g->nkb + i / 16) * TR_CHUNK + tr_chunk_pos((int)(b % TR_BN), (int)(i % 16))];
  return PSF_OK;
}

// which kernels walked the last call (bench.py's roofline label; tests): form 1 = k_np_walk<G>, one launch; 0 = one k_np_step<G> launch per 64-row block; `reruns` = walks of
// this handle since its creation that gave up waiting and were walked again by k_np_walk_solo (0 on a GPU that is not shared)
psf_status psfgpv_get_nearest_plane_form(psfgpv_handle* g, int* form, int* preimages_per_wave, size_t* blocks, uint64_t* reruns) {
  if (!g) return PSF_ERR_PARAM;
  HIP_TRY(hipSetDevice(g->base->prm.device));
  HIP_TRY(hipStreamSynchronize(g->last_stream));
  unsigned long long r = 0;
  HIP_TRY(hipMemcpy(&r, g->dFlags + 8, sizeof(r), hipMemcpyDeviceToHost));
  if (form) *form = (g->last_form == 0 && g->last_parts == 2) ? 2 : g->last_form;
  if (preimages_per_wave) *preimages_per_wave = g->last_G;
  if (blocks) *blocks = g->nblk;
  if (reruns) *reruns = (uint64_t)r;
  return PSF_OK;
}
// tests: force the form of the walk (-1 = by batch size, 0 = one launch per block, 1 = one launch where it fits) and the number of polls after which a wait of the
// one-launch walk gives up (0 = keep; 1 makes every call take the k_np_walk_solo route)
psf_status psfgpv_debug_set_walk(psfgpv_handle* g, int form, unsigned spins) {
  if (!g || form < -1 || form > 1) return PSF_ERR_PARAM;
  g->np_walk = form;
  if (spins) g->walk_spins = spins;
  return PSF_OK;
}
// tests (experiments build): the two-halves form of launch-per-block batches: 0 = never, 1 = whenever the shape allows (from 256 preimages on), 2 = from 3072 preimages on
psf_status psfgpv_debug_set_split(psfgpv_handle* g, int split) {
  if (!g || split < 0 || split > 2) return PSF_ERR_PARAM;
  if (!psf_experiments_build) return split == 0 ? PSF_OK : PSF_ERR_UNSUPPORTED;
  g->np_split = split;
  return PSF_OK;
}
// column ranges the last call walked side by side (1 or 2)
int psfgpv_debug_last_parts(const psfgpv_handle* g) { return g ? g->last_parts : 0; }

psf_status psfgpv_get_nearest_plane_stats(psfgpv_handle* g, size_t* blocks, size_t* generic_recombination) {
  if (!g) return PSF_ERR_PARAM;
  HIP_TRY(hipStreamSynchronize(g->last_stream));
  int fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIP_TRY(hipMemcpy(fl, g->dFlags, sizeof(fl), hipMemcpyDeviceToHost));
  if (blocks) *blocks = g->nblk;
  if (generic_recombination) *generic_recombination = (g->basis_generic || fl[g->two_pass ? 7 : 3]) ? 1 : 0;
  return PSF_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// PSFGPVRing (gpv_ring.rs:62-67, impl PSF :69-284): ring key material on the host, then the embedded PSFGPV machinery.
// ---------------------------------------------------------------------------------------------------------------------
struct psfring_handle {
  psfgpv_handle* g = nullptr;          // n x d system with d = n(k+2) (gadget parameters n, k, m_bar' = 2n)
  psf_gadget_params gp;                // ring parameters (m_bar = k + 2)
  double s_td = 0;
  std::vector<uint64_t> a;             // (k+2) x n
  std::vector<int64_t> r, e;           // k x n
  uint32_t* dHat = nullptr;            // NTT images of the k+2 polynomials of a (psf_ntt_api.hpp), when (q, n) has a wave kernel
  bool fa_ntt = false;                 // f_a as k+2 R_q products (gpv_ring.rs:243-247) instead of the embedded matrix product
};

static psf_status ring_install(psfring_handle* h) {
  const size_t n = h->gp.n, K = h->gp.k + 2;
  std::vector<int32_t> bt;
  const psf_status rc = ring_short_basis_t(h->gp, h->a.data(), h->r.data(), h->e.data(), bt);   // gpv_ring.rs:169
  if (rc != PSF_OK) return rc;
  std::vector<uint64_t> A_emb;
  ring_embed_a(h->a.data(), n, K, h->gp.q, A_emb);                                               // gpv_ring.rs:172-178
  psfgpv_handle* g = h->g;
  psfp_handle* b = g->base;
  HIP_TRY(hipSetDevice(b->prm.device));
  PSFP_QUIESCE(b);
  g->has_key = false;
  HIP_TRY(hipMemcpy(b->dA, A_emb.data(), A_emb.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  split_A(b);
  HIP_TRY(hipMemcpy(g->dSt, bt.data(), bt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  psf_status rc2 = gpv_finish_basis(g, true);           // GSO of the embedded basis (inside MatPolyOverZ::sample_d, gpv_ring.rs:205)
  if (rc2 != PSF_OK) return rc2;
  rc2 = gpv_build_solver(g);                            // gpv_ring.rs:180-185
  if (rc2 != PSF_OK) return rc2;
  b->has_key = true; b->has_pub = true;
  g->has_key = true;
  // the key side of f_a is transformed once: a -> the images of its k+2 polynomials
  h->fa_ntt = false;
  if (ntt_route(h->gp.q, n) == 2 && ((size_t)K * n + 4 * n) * sizeof(uint32_t) <= 64 * 1024) {
    // (the key is installed and usable at this point: a failure below only means that f_a keeps the matrix-product route, it is not an error of the call)
    uint64_t* da = nullptr;
    psf_status rf = (h->dHat || hipMalloc(&h->dHat, K * n * sizeof(uint32_t)) == hipSuccess) ? PSF_OK : PSF_ERR_HIP;
    if (rf == PSF_OK && hipMalloc(&da, K * n * sizeof(uint64_t)) != hipSuccess) rf = PSF_ERR_HIP;
    if (rf == PSF_OK && hipMemcpy(da, h->a.data(), K * n * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) rf = PSF_ERR_HIP;
    if (rf == PSF_OK) rf = ntt_forward_dev(b->prm.device, h->gp.q, n, K, da, 64, h->dHat, nullptr);
    if (rf == PSF_OK && hipDeviceSynchronize() != hipSuccess) rf = PSF_ERR_HIP;
    hipFree(da);
    if (rf != PSF_OK) (void)hipGetLastError();
    h->fa_ntt = rf == PSF_OK;
  }
  return PSF_OK;
}

// one R_q product kernel launch on device buffers: the NTT when (q, n) has one, the exact schoolbook kernel otherwise (64-bit layout only)
static psf_status polymul_dev_any(int device, uint64_t q, size_t n, size_t count, const void* da, const void* db, void* dout, int io_bits, int method, hipStream_t st) {
  if (method != 0) {
    const psf_status rc = ntt_polymul_dev(device, q, n, count, da, db, dout, io_bits, st);
    if (rc != PSF_ERR_UNSUPPORTED || method == 1) return rc;
  }
  if (io_bits != 64) return PSF_ERR_UNSUPPORTED;
  if (count == 0) return PSF_OK;
  HIP_TRY(hipSetDevice(device));
  const uint64_t two64 = (uint64_t)((((u128)1) << 64) % q);
  hipLaunchKernelGGL(k_polymul_negacyclic, dim3((unsigned)count), dim3(256), 2 * n * sizeof(uint64_t), st, q, two64, (uint32_t)n, (const uint64_t*)da, n, (const int64_t*)db, n,
                     (uint64_t*)dout, n);
  HIP_TRY(hipGetLastError());
  return PSF_OK;
}

extern "C" {

psf_status psfring_create(const psfring_params* prm, psfring_handle** out) {
  if (!prm || !out || !(prm->s > 0.0) || !(prm->s_td > 0.0)) return PSF_ERR_PARAM;
  const psf_gadget_params& gp = prm->gp;
  if (gp.n < 1 || gp.k < 1 || gp.q <= 1 || gp.q >= (1ull << 62)) return gp.q >= (1ull << 62) ? PSF_ERR_UNSUPPORTED : PSF_ERR_PARAM;
  psfgpv_params gpvp;
  gpvp.gp = psf_gadget_params{gp.n, gp.k, 2 * gp.n, gp.base, gp.q};
  gpvp.s = prm->s; gpvp.device = prm->device; gpvp.flags = 0;
  psfgpv_handle* g = nullptr;
  const psf_status rc = psfgpv_create(&gpvp, &g);
  if (rc != PSF_OK) return rc;
  psfring_handle* h = new psfring_handle();
  h->g = g; h->gp = gp; h->s_td = prm->s_td;
  *out = h;
  return PSF_OK;
}

void psfring_destroy(psfring_handle* h) {
  if (!h) return;
  if (h->dHat) hipFree(h->dHat);
  psfgpv_destroy(h->g);
  delete h;
}

// method: 0 = schoolbook kernel, 1 = NTT (PSF_ERR_UNSUPPORTED without a plan), -1 = NTT when there is one
psf_status psf_poly_mul_negacyclic_method(int device, uint64_t q, size_t n, size_t count, const uint64_t* a, const int64_t* b, uint64_t* out, int method) {
  if (q <= 1 || q >= (1ull << 62) || n < 1 || n > 8192 || (count && (!a || !b || !out))) return PSF_ERR_PARAM;
  if (method == 1 && ntt_route(q, n) == 0) return PSF_ERR_UNSUPPORTED;
  if (count == 0) return PSF_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return PSF_ERR_HIP;
  HIP_TRY(hipSetDevice(device));
  uint64_t *da = nullptr, *dout = nullptr; int64_t* db = nullptr;
  auto done = [&](psf_status st) { hipFree(da); hipFree(db); hipFree(dout); return st; };
  if (hipMalloc(&da, count * n * sizeof(uint64_t)) != hipSuccess || hipMalloc(&db, count * n * sizeof(int64_t)) != hipSuccess ||
      hipMalloc(&dout, count * n * sizeof(uint64_t)) != hipSuccess) return done(PSF_ERR_HIP);
  if (hipMemcpy(da, a, count * n * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db, b, count * n * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) return done(PSF_ERR_HIP);
  const psf_status rc = polymul_dev_any(device, q, n, count, da, db, dout, 64, method, nullptr);
  if (rc != PSF_OK) return done(rc);
  if (hipMemcpy(out, dout, count * n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return done(PSF_ERR_HIP);
  return done(PSF_OK);
}

psf_status psf_poly_mul_negacyclic(int device, uint64_t q, size_t n, size_t count, const uint64_t* a, const int64_t* b, uint64_t* out) {
  return psf_poly_mul_negacyclic_method(device, q, n, count, a, b, out, -1);
}

psf_status psf_poly_mul_negacyclic_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, const void* d_b, void* d_out, int io_bits, void* stream) {
  if (q <= 1 || q >= (1ull << 62) || n < 1 || n > 8192 || (io_bits != 16 && io_bits != 64) || (count && (!d_a || !d_b || !d_out))) return PSF_ERR_PARAM;
  return polymul_dev_any(device, q, n, count, d_a, d_b, d_out, io_bits, -1, (hipStream_t)stream);
}
psf_status psf_ntt_forward_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, int io_bits, uint32_t* d_hat, void* stream) {
  return ntt_forward_dev(device, q, n, count, d_a, io_bits, d_hat, (hipStream_t)stream);
}
psf_status psf_poly_mul_hat_dev(int device, uint64_t q, size_t n, size_t count, const uint32_t* d_hat, size_t hat_stride, const void* d_b, void* d_out, int io_bits,
                                void* stream) {
  return ntt_mul_hat_dev(device, q, n, count, d_hat, hat_stride, d_b, d_out, io_bits, (hipStream_t)stream);
}

// MatQ::gso (gpv.rs:88-91) as a free function: rows of an integer matrix -> their Gram-Schmidt vectors
psf_status psf_gso_rows(int device, const int32_t* basis_t, size_t rows, size_t width, double* out) {
  if (!basis_t || !out || rows < 1 || width < 1) return PSF_ERR_PARAM;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return PSF_ERR_HIP;
  HIP_TRY(hipSetDevice(device));
  int32_t* dS = nullptr; double* dG = nullptr; int* dinfo = nullptr;
  auto done = [&](psf_status st) { hipFree(dS); hipFree(dG); hipFree(dinfo); return st; };
  if (hipMalloc(&dS, rows * width * sizeof(int32_t)) != hipSuccess || hipMalloc(&dG, rows * width * sizeof(double)) != hipSuccess ||
      hipMalloc(&dinfo, sizeof(int)) != hipSuccess) return done(PSF_ERR_HIP);
  if (hipMemcpy(dS, basis_t, rows * width * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess || hipMemset(dinfo, 0, sizeof(int)) != hipSuccess) return done(PSF_ERR_HIP);
  hipLaunchKernelGGL(k_i32_to_f64, dim3(grid_for(rows * width)), dim3(256), 0, 0, dS, dG, rows * width);
  if (gso_blocked(nullptr, dG, rows, width, dinfo) != hipSuccess) return done(PSF_ERR_HIP);
  int info = 0;
  if (hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return done(PSF_ERR_HIP);
  if (info != 0) return done(PSF_ERR_PARAM);
  if (hipMemcpy(out, dG, rows * width * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return done(PSF_ERR_HIP);
  return done(PSF_OK);
}

// gen_trapdoor_ring_lwe (gadget_ring.rs:62-81): A = [1 | a_bar | g_j - (a_bar r_j + e_j)] mod (X^n + 1, q); the k products a_bar * r_j run on the
// device (NTT kernel when q allows, the exact schoolbook kernel for every other q < 2^62).  r, e: SampleZ(s) from `seed`
// (trapdoor_distribution.rs:112-122), or the caller's own draw (`params.distribution.sample(...)`, gadget_ring.rs:69-70).
static psf_status ring_lwe_assemble(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const int64_t* r, const int64_t* e, uint64_t* a) {
  const size_t n = gp->n, k = gp->k;
  const uint64_t q = gp->q;
  std::vector<uint64_t> prod(k * n);
  psf_status rc;
  if (ntt_route(q, n) == 2) {                                            // a_bar is transformed ONCE, then k products image x r_j
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return PSF_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    uint64_t *da = nullptr, *dout = nullptr; int64_t* dr = nullptr; uint32_t* dhat = nullptr;
    auto done = [&](psf_status st) { hipFree(da); hipFree(dr); hipFree(dout); hipFree(dhat); return st; };
    if (hipMalloc(&da, n * sizeof(uint64_t)) != hipSuccess || hipMalloc(&dr, k * n * sizeof(int64_t)) != hipSuccess ||
        hipMalloc(&dout, k * n * sizeof(uint64_t)) != hipSuccess || hipMalloc(&dhat, n * sizeof(uint32_t)) != hipSuccess) return done(PSF_ERR_HIP);
    if (hipMemcpy(da, a_bar, n * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dr, r, k * n * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) return done(PSF_ERR_HIP);
    rc = ntt_forward_dev(device, q, n, 1, da, 64, dhat, nullptr);
    if (rc == PSF_OK) rc = ntt_mul_hat_dev(device, q, n, k, dhat, 0, dr, dout, 64, nullptr);                 // a_bar * r (:78)
    if (rc == PSF_OK && hipMemcpy(prod.data(), dout, k * n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = PSF_ERR_HIP;
    if (done(rc) != PSF_OK) return rc;
  } else {
    std::vector<uint64_t> abar_rep(k * n);
    for (size_t j = 0; j < k; ++j)
      for (size_t c = 0; c < n; ++c) abar_rep[j * n + c] = a_bar[c] % q;
    rc = psf_poly_mul_negacyclic(device, q, n, k, abar_rep.data(), r, prod.data());                 // a_bar * r (:78)
    if (rc != PSF_OK) return rc;
  }
  for (size_t c = 0; c < (k + 2) * n; ++c) a[c] = 0;
  a[0] = 1 % q;                                                                                 // :74-76
  for (size_t c = 0; c < n; ++c) a[n + c] = a_bar[c] % q;
  uint64_t gpow = 1 % q;
  for (size_t j = 0; j < k; ++j) {                                                              // g^t - (a_bar r + e), :77-78
    for (size_t c = 0; c < n; ++c) {
      i128 v = (c == 0 ? (i128)gpow : (i128)0) - ((i128)prod[j * n + c] + e[j * n + c]);
      v %= (i128)q;
      if (v < 0) v += q;
      a[(2 + j) * n + c] = (uint64_t)v;
    }
    gpow = mulmod_u64(gpow, gp->base % q, q);
  }
  return PSF_OK;
}
psf_status psf_gen_trapdoor_ring_lwe(int device, const psf_gadget_params* gp, const uint64_t* a_bar, double s, uint64_t seed, uint64_t* a, int64_t* r, int64_t* e) {
  if (!gp || !a_bar || !a || !r || !e || !(s > 0.0) || gp->n < 1 || gp->k < 1 || gp->q <= 1) return PSF_ERR_PARAM;
  if (gp->q >= (1ull << 62)) return PSF_ERR_UNSUPPORTED;
  const size_t n = gp->n, k = gp->k;
  const SampleZParams sp = make_sample_z_params(s);
  int fail = 0;
  for (size_t j = 0; j < k; ++j)
    for (size_t c = 0; c < n; ++c) {
      r[j * n + c] = sample_z(seed, TAG_RING_R, 0, (uint32_t)(j * n + c), 0.0, sp, &fail);      // :69
      e[j * n + c] = sample_z(seed, TAG_RING_E, 0, (uint32_t)(j * n + c), 0.0, sp, &fail);      // :70
    }
  if (fail) return PSF_ERR_SAMPLER;
  return ring_lwe_assemble(device, gp, a_bar, r, e, a);
}
psf_status psf_gen_trapdoor_ring_lwe_with(int device, const psf_gadget_params* gp, const uint64_t* a_bar, const int64_t* r, const int64_t* e, uint64_t* a) {
  if (!gp || !a_bar || !a || !r || !e || gp->n < 1 || gp->k < 1 || gp->q <= 1) return PSF_ERR_PARAM;
  if (gp->q >= (1ull << 62)) return PSF_ERR_UNSUPPORTED;
  for (size_t i = 0; i < gp->k * gp->n; ++i)
    if (r[i] > (1ll << 30) || r[i] < -(1ll << 30) || e[i] > (1ll << 30) || e[i] < -(1ll << 30)) return PSF_ERR_UNSUPPORTED;   // the embedded short basis is int32
  return ring_lwe_assemble(device, gp, a_bar, r, e, a);
}

// gen_gadget_ring (gadget_ring.rs:103-109): k constant polynomials base^j; out[j] = the constant term
psf_status psf_gen_gadget_ring(uint64_t k, uint64_t base, int64_t* out) { return psf_gen_gadget_vec(k, base, out); }

// find_solution_gadget_ring (gadget_ring.rs:145-166): u in R_q as n coefficients -> k polynomials, polynomial i = the i-th base-`base` digit of
// every coefficient (index i + j k of the classical solution, :160).  The digits come from the device kernel.
psf_status psf_find_solution_gadget_ring(int device, const uint64_t* u, size_t n, uint64_t q, uint64_t k, uint64_t base, int64_t* out) {
  if (!u || !out || n < 1) return PSF_ERR_PARAM;
  std::vector<int64_t> classical(k * n);
  const psf_status rc = psf_find_solution_gadget_mat(device, u, n, 1, q, k, base, classical.data());   // out[k j + i] = digit i of u_j
  if (rc != PSF_OK) return rc;
  for (size_t i = 0; i < k; ++i)
    for (size_t j = 0; j < n; ++j) out[i * n + j] = classical[i + j * k];
  return PSF_OK;
}

// gen_short_basis_for_trapdoor_ring (short_basis_ring.rs:64-79): the (k+2) x n(k+2) matrix of polynomials sa_l * sa_r reduced by X^n + 1,
// out[(row * n(k+2) + col) * n + coefficient]
psf_status psf_gen_short_basis_for_trapdoor_ring(const psf_gadget_params* gp, const uint64_t* a, const int64_t* r, const int64_t* e, int64_t* out) {
  if (!gp || !a || !r || !e || !out) return PSF_ERR_PARAM;
  std::vector<int32_t> bt;
  const psf_status rc = ring_short_basis_t(*gp, a, r, e, bt);
  if (rc != PSF_OK) return rc;
  const size_t n = gp->n, K = gp->k + 2, d = n * K;
  for (size_t col = 0; col < d; ++col)
    for (size_t row = 0; row < K; ++row)
      for (size_t c = 0; c < n; ++c) out[(row * d + col) * n + c] = bt[col * d + row * n + c];
  return PSF_OK;
}

// gpv_ring.rs:91-98: a_bar uniform (:92-94), then gen_trapdoor_ring_lwe
psf_status psfring_trap_gen(psfring_handle* h, uint64_t seed) {
  if (!h) return PSF_ERR_PARAM;
  const size_t n = h->gp.n, k = h->gp.k;
  h->r.assign(k * n, 0); h->e.assign(k * n, 0); h->a.assign((k + 2) * n, 0);
  std::vector<uint64_t> a_bar(n);
  for (size_t c = 0; c < n; ++c) a_bar[c] = uniform_mod(seed, TAG_RING_A, (uint32_t)c, 0, h->gp.q);
  const psf_status rc = psf_gen_trapdoor_ring_lwe(h->g->base->prm.device, &h->gp, a_bar.data(), h->s_td, seed, h->a.data(), h->r.data(), h->e.data());
  if (rc != PSF_OK) return rc;
  return ring_install(h);
}

psf_status psfring_load_key(psfring_handle* h, const uint64_t* a, const int64_t* r, const int64_t* e) {
  if (!h || !a || !r || !e) return PSF_ERR_PARAM;
  const size_t n = h->gp.n, k = h->gp.k;
  h->a.assign(a, a + (k + 2) * n);
  h->r.assign(r, r + k * n);
  h->e.assign(e, e + k * n);
  return ring_install(h);
}

psf_status psfring_export_key(const psfring_handle* h, uint64_t* a, int64_t* r, int64_t* e, int32_t* basis_t, double* gso_t) {
  if (!h) return PSF_ERR_PARAM;
  if (!h->g->has_key) return PSF_ERR_NO_KEY;
  if (a) std::memcpy(a, h->a.data(), h->a.size() * sizeof(uint64_t));
  if (r) std::memcpy(r, h->r.data(), h->r.size() * sizeof(int64_t));
  if (e) std::memcpy(e, h->e.data(), h->e.size() * sizeof(int64_t));
  return psfgpv_export_key(h->g, nullptr, nullptr, basis_t, gso_t);
}

psf_status psfring_samp_d(psfring_handle* h, uint64_t seed, uint64_t first_index, size_t B, int64_t* sigma) {
  return h ? psfgpv_samp_d(h->g, seed, first_index, B, sigma) : PSF_ERR_PARAM;
}
psf_status psfring_samp_p(psfring_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* sigma) {
  return h ? psfgpv_samp_p(h->g, seed, first_index, B, u, sigma) : PSF_ERR_PARAM;
}
psf_status psfring_samp_p_async(psfring_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* u, int64_t* sigma) {
  return h ? psfgpv_samp_p_async(h->g, seed, first_index, B, u, sigma) : PSF_ERR_PARAM;
}
psf_status psfring_wait(psfring_handle* h) { return h ? psfgpv_wait(h->g) : PSF_ERR_PARAM; }
uint64_t psfring_async_next_ticket(const psfring_handle* h) { return h ? psfgpv_async_next_ticket(h->g) : 0; }
psf_status psfring_wait_ticket(psfring_handle* h, uint64_t ticket) { return h ? psfgpv_wait_ticket(h->g, ticket) : PSF_ERR_PARAM; }
psf_status psfring_samp_p_dev(psfring_handle* h, uint64_t seed, uint64_t first_index, size_t B, const uint64_t* d_u, int64_t* d_sigma, void* stream) {
  return h ? psfgpv_samp_p_dev(h->g, seed, first_index, B, d_u, d_sigma, stream) : PSF_ERR_PARAM;
}
// gpv_ring.rs:243-247: the domain check, then u = sum_j a_j * sigma_j in R_q.  With the images of a (ring_install) that is k+2 forward
// transforms, k+2 leaf products and one inverse transform per preimage (k_ring_fa); PSF_RING_FA=matmul keeps the product with rot^-(iota(a)) on
// the int8 matrix cores (same residues; also the route of every (q, n) without a wave kernel).
static bool ring_fa_by_ntt(const psfring_handle* h) {
  static const int forced = [] { const char* e = psf_exp_env("PSF_RING_FA"); return !e ? 0 : (std::strcmp(e, "matmul") == 0 ? 1 : 2); }();
  return h->fa_ntt && forced != 1;
}
psf_status psfring_f_a_dev(psfring_handle* h, size_t B, const int64_t* d_sigma, uint64_t* d_u, uint8_t* d_ok, void* stream) {
  if (!h) return PSF_ERR_PARAM;
  if (!ring_fa_by_ntt(h)) return psfgpv_f_a_dev(h->g, B, d_sigma, d_u, d_ok, stream);
  if (B && (!d_sigma || !d_u || !d_ok)) return PSF_ERR_PARAM;
  if (!h->g->has_key) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  psfp_handle* b = h->g->base;
  HIP_TRY(hipSetDevice(b->prm.device));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_check_domain, dim3((unsigned)B), dim3(256), 0, st, d_sigma, b->m, b->m, domain_bound(b), d_ok);      // :244
  HIP_TRY(hipGetLastError());
  const psf_status rc = ntt_ring_fa_dev(b->prm.device, h->gp.q, h->gp.n, (uint32_t)(h->gp.k + 2), h->dHat, d_sigma, d_u, B, st);   // :245-246
  b->last_stream = st;
  return rc;
}
psf_status psfring_f_a(psfring_handle* h, size_t B, const int64_t* sigma, uint64_t* u) {
  if (!h) return PSF_ERR_PARAM;
  if (!ring_fa_by_ntt(h)) return psfgpv_f_a(h->g, B, sigma, u);
  if (B && (!sigma || !u)) return PSF_ERR_PARAM;
  if (!h->g->has_key) return PSF_ERR_NO_KEY;
  if (B == 0) return PSF_OK;
  psfp_handle* b = h->g->base;
  HIP_TRY(hipSetDevice(b->prm.device));
  { const psf_status rw = psfp_wait(b); if (rw != PSF_OK) return rw; }
  psf_status rc = ensure_batch(b, B);
  if (rc != PSF_OK) return rc;
  HIP_TRY(hipMemcpy(b->dE, sigma, B * b->m * sizeof(int64_t), hipMemcpyHostToDevice));
  rc = psfring_f_a_dev(h, B, b->dE, b->dU, b->dOk, nullptr);
  if (rc != PSF_OK) return rc;
  std::vector<uint8_t> ok(B);
  HIP_TRY(hipMemcpy(u, b->dU, B * b->n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ok.data(), b->dOk, B, hipMemcpyDeviceToHost));
  for (uint8_t o : ok) if (!o) return PSF_ERR_DOMAIN;
  return PSF_OK;
}
psf_status psfring_check_domain(psfring_handle* h, size_t B, const int64_t* sigma, size_t len, uint8_t* ok) {
  return h ? psfgpv_check_domain(h->g, B, sigma, len, ok) : PSF_ERR_PARAM;
}
psf_status psfring_uniform_targets_dev(psfring_handle* h, uint64_t seed, uint64_t first_index, size_t B, uint64_t* d_u, void* stream) {
  return h ? psfgpv_uniform_targets_dev(h->g, seed, first_index, B, d_u, stream) : PSF_ERR_PARAM;
}
psf_status psfring_last_status(psfring_handle* h) { return h ? psfgpv_last_status(h->g) : PSF_ERR_PARAM; }
psf_status psfring_enable_timing(psfring_handle* h, int on) { return h ? psfgpv_enable_timing(h->g, on) : PSF_ERR_PARAM; }
psf_status psfring_get_timing(psfring_handle* h, double* a, double* b) { return h ? psfgpv_get_timing(h->g, a, b) : PSF_ERR_PARAM; }
psf_status psfring_get_nearest_plane_form(psfring_handle* h, int* form, int* preimages_per_wave, size_t* blocks, uint64_t* reruns) {
  return h ? psfgpv_get_nearest_plane_form(h->g, form, preimages_per_wave, blocks, reruns) : PSF_ERR_PARAM;
}
psf_status psfring_debug_set_walk(psfring_handle* h, int form, unsigned spins) { return h ? psfgpv_debug_set_walk(h->g, form, spins) : PSF_ERR_PARAM; }
psf_status psfring_debug_set_split(psfring_handle* h, int split) { return h ? psfgpv_debug_set_split(h->g, split) : PSF_ERR_PARAM; }
int psfring_debug_last_parts(const psfring_handle* h) { return h ? psfgpv_debug_last_parts(h->g) : 0; }

}  // extern "C"
