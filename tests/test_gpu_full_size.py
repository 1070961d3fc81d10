"""BASELINE.json's full-size configurations through the size-independent properties the domain offers: every preimage of
the batch satisfies A e = u (mod q) and the norm bound of check_domain (mp_perturbation.rs:366-369, :396-402; gpv.rs:190-193,
:219-224; gpv_ring.rs:243-247, :274-283), and a bounded sample of the same batch is bitwise equal to the CPU oracle.
Runs bench.py (the measured path itself) in a subprocess and reads its JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("config,sample", [("c3", 256), ("c3prime", 256), ("c2", 64), ("c2s240", 64), ("c4", 128)])
def test_full_size_batch_is_valid_and_matches_the_oracle_on_a_sample(config, sample):
    d = run_bench("--config", config, "--steps", "1", "--warmup", "0", "--cpu-sample", str(sample))
    assert d["valid"] is True                                   # A e = u and check_domain for every row of the batch
    assert d["cpu_baseline"]["matches_gpu_bitwise"] is True     # the sampled rows equal the oracle's
    assert d["n_gpus"] == 1 and d["data"] == "synthetic"
    if config in ("c3", "c3prime"):          # c3prime: SURVEY 8d's second C3 point, q = 1073741789 (digit column in S_k, non-mask reduction mod q)
        assert d["roofline"]["bound"] == "mfma" and 0.5 < d["roofline"]["frac"] <= 1.0


@pytest.mark.timeout(2400)
def test_c5_per_gpu_shape_is_valid_and_matches_the_oracle_on_a_sample(oracle):
    """BASELINE.json configs[4] as one of its eight ranks sees it: n = 1024, q = 2^60 (k = 60, m = 122 980), 8192 preimages,
    60.5 GB factor.  Every row: A e = u and check_domain.  A sample of 16 rows: every stage bit for bit against the oracle --
    normals, centres x = sqrt(Sigma_2) d (the factor is streamed back in row blocks and pushed through the oracle's
    ascending fma chain), perturbation, syndrome, gadget preimage, e."""
    import numpy as np
    import torch
    import tools_amd as T
    n, q, r, s, B = 1024, 2**60, 10.0, 1024.0, 8192
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    psf.trap_gen(3, export=False)
    m = psf.m
    assert (psf.k, m) == (60, 122980)
    dev = torch.device("cuda:0")
    first = 5 * B                                               # the index range rank 5 of 8 owns
    u = torch.empty((B, n), dtype=torch.int64, device=dev)
    e = torch.empty((B, m), dtype=torch.int64, device=dev)
    psf.uniform_targets_dev(u.data_ptr(), B, seed=7, first_index=first)
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=1000, first_index=first)
    torch.cuda.synchronize()
    assert psf.last_status() == 0
    u2 = torch.empty_like(u)
    ok = torch.empty((B,), dtype=torch.uint8, device=dev)
    psf.f_a_dev(e.data_ptr(), u2.data_ptr(), ok.data_ptr(), B)
    torch.cuda.synchronize()
    assert bool((u2 == u).all().item()) and bool(ok.all().item())
    # ---- 16 rows spread over the batch, recomputed as a batch of their own (rows do not depend on the batch they ride in)
    rows = [0, 1, 127, 128, 1000, 4095, 4096, 8191]
    S = 16
    lo = 4000
    uh = u[lo:lo + S].cpu().numpy().astype(np.uint64)
    st = psf.samp_p_stages(uh, seed=1000, first_index=first + lo)
    assert (st["e"] == e[lo:lo + S].cpu().numpy()).all()
    for rr in rows:                                             # and single rows anywhere in the batch
        one = psf.samp_p(u[rr].cpu().numpy().astype(np.uint64), seed=1000, first_index=first + rr)
        assert (one == e[rr].cpu().numpy()).all()
    d_ref = np.array([oracle.normals(1000, first + lo + b, m) for b in range(S)])
    assert (st["d"].view(np.uint64) == d_ref.view(np.uint64)).all(), "normals differ"
    step = 2048
    for row0 in range(0, m, step):
        nr = min(step, m - row0)
        Lr = psf.export_sqrt_sigma2_rows(row0, nr)
        x_ref = oracle.centres_rows(Lr, row0, nr, m, d_ref)
        assert (st["x"][:, row0:row0 + nr].view(np.uint64) == x_ref.view(np.uint64)).all(), f"centres differ in rows {row0}.."
    A, R = psf.export_A_R()
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s, with_L=False)
    orc.load_key(A, R)
    for b in range(S):
        fx = orc.samp_p_from_x(1000, first + lo + b, uh[b], st["x"][b])
        assert (fx["p"] == st["p"][b]).all(), "perturbation differs"
        assert (fx["v"] == st["v"][b]).all(), "syndrome differs"
        assert (fx["z"] == st["z"][b]).all(), "gadget preimage differs"
        assert (fx["e"] == st["e"][b]).all(), "preimage differs"
    psf.close()


@pytest.mark.timeout(1200)
def test_c3_single_calls_match_the_oracle_in_every_stage(oracle):
    """The streaming product of a SINGLE call (k_trmm_stream: 1 and 16 preimages, psf.rs:48-80 -- one reference call is one preimage; round 6: 24 and 40 preimages on
    the LDS-shared tiles, with the recombination of k_recombine_wg) at the C3 shape
    (n = 512, q = 2^30, m = 30 801): every stage bit for bit against the oracle, the centres through the oracle's ascending fma chain over the factor streamed
    back in row blocks.  (The batch kernels at this shape are covered by the bench runs above, the single call at the C5 shape by the test above.)"""
    import numpy as np
    import tools_amd as T
    n, q, r, s = 512, 2**30, 9.0, 512.0
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    psf.trap_gen(11, export=False)
    m = psf.m
    A, R = psf.export_A_R()
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s, with_L=False)
    orc.load_key(A, R)
    for S, first in ((1, 123456), (16, 7), (24, 900), (40, 2**33 + 5), (72, 31)):      # k_trmm_stream (dense normals stream), k_trmm_stream_wg32, k_trmm_stream_wg, both over six fragments; k_gadget_wave / quad<., 16> / quad<., 4>
        uh = oracle.uniform_targets(5, S, n, q)
        st = psf.samp_p_stages(uh, seed=42, first_index=first)
        assert (psf.samp_p(uh, seed=42, first_index=first) == st["e"]).all()
        d_ref = np.array([oracle.normals(42, first + b, m) for b in range(S)])
        assert (st["d"].view(np.uint64) == d_ref.view(np.uint64)).all(), "normals differ"
        for row0 in (0, 4096, 16384, m - 2048):               # a few row blocks of the factor, the last one included
            nr = min(2048, m - row0)
            Lr = psf.export_sqrt_sigma2_rows(row0, nr)
            x_ref = oracle.centres_rows(Lr, row0, nr, m, d_ref)
            assert (st["x"][:, row0:row0 + nr].view(np.uint64) == x_ref.view(np.uint64)).all(), f"centres differ in rows {row0}.. at {S} preimages"
        for b in range(S):
            fx = orc.samp_p_from_x(42, first + b, uh[b], st["x"][b])
            assert (fx["p"] == st["p"][b]).all() and (fx["v"] == st["v"][b]).all() and (fx["z"] == st["z"][b]).all() and (fx["e"] == st["e"][b]).all()
    psf.close()


@pytest.mark.timeout(900)
def test_c5_rank_zero_of_eight_fits_beside_the_key():
    """BASELINE.json configs[4] as RANK 0 of its eight ranks sees it (the part of the 8-GPU job that can be rehearsed on one GPU): the 60.6 GB key, the batch
    buffers of 8192 preimages and the gather ring sized for eight ranks (`--emulate-world 8`: 36.3 GB per level on the receiving rank, tools_amd/shard.py) all
    resident, through bench.py's own launcher and a one-rank RCCL group.  The ring must get at least one level (AsyncRowGather.fit_depth), nothing may run out
    of memory, every row must pass A e = u and check_domain."""
    d = run_bench("--config", "c5", "--force-dist", "--emulate-world", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-latency", timeout=850)
    assert d["valid"] is True and d["ranks_seen"] == 1 and "self-spawned" in d["launcher"]
    g = d["gather"]
    assert g["buffers_sized_for_ranks"] == 8 and g["depth"] >= 1
    assert g["bytes_per_depth"] == 8192 * 122980 * 4 * 9                 # its own staging block + one receive block per rank
    assert g["hbm_free_GB_after_allocation"] > 10.0
