"""Seeded random configurations, GPU against the oracle: the parametrised parity tests pin the reference's own sizes and the shapes a kernel was written for;
this file draws the rest -- odd n, prime / power-of-two / arbitrary moduli up to 2^61, gadget bases 2..7, Gaussians from barely positive definite to wide,
ragged batch sizes on both sides of every tile boundary (64, 128, 256) -- so that a dispatch rule or a padding assumption nobody thought of shows up as a
bit difference.  Every draw is reproducible (numpy Generator with a fixed seed per case); whole batches are compared, and the reference's invariants
(f_a(samp_p(u)) == u, check_domain; mp_perturbation.rs:433-448, gpv.rs:253-268, gpv_ring.rs:318-334) are asserted on the GPU rows."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRIMES = [257, 3329, 7681, 12289, 65537, 1073741789, 2**31 - 1, 2**61 - 1]
# the menus the draws choose from (tools/fuzz_configs.py --wide swaps in broader ones)
R_MENU = [1.5, 2.0, 3.0, 4.5, 30.0]
S_FACTOR_MENU = [1.1, 1.5, 3.0]
BATCH_MENU = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300]
GPV_S_MENU = [8.0, 30.0, 240.0, 1000.0]


def draw_modulus(rng):
    kind = rng.integers(0, 3)
    if kind == 0:
        return int(2 ** rng.integers(4, 61))
    if kind == 1:
        return int(PRIMES[rng.integers(0, len(PRIMES))])
    return int(rng.integers(17, 2**20)) | 1


def draw_batch(rng):
    return int(rng.choice(BATCH_MENU))


@pytest.mark.parametrize("case", range(64))
def test_perturbation_random_configuration(oracle, case):
    import tools_amd as T
    rng = np.random.default_rng(1000 + case)
    while True:                                              # a draw outside the documented domain bound (s r sqrt(m) < 2^23) is redrawn, not skipped
        n = int(rng.integers(1, 13))
        q = draw_modulus(rng)
        base = int(rng.choice([2, 2, 2, 3, 5, 7]))
        k = 1
        while base**k < q:
            k += 1
        if k > 64:
            base, k = 2, int(math.ceil(math.log2(q)))
        m_bar = n * int(math.ceil(math.log2(q))) + int(rng.integers(0, 40))
        r = float(rng.choice(R_MENU))
        # sigma_max(R) <= sqrt(m_bar) + sqrt(n k) + a few; s from 1.1x the positive-definiteness bound upwards
        bound = r * math.sqrt(base * base + 1) * (math.sqrt(m_bar) + math.sqrt(n * k) + 4.0)
        s = bound * float(rng.choice(S_FACTOR_MENU))
        B = draw_batch(rng)
        if s * r * math.sqrt(m_bar + n * k) < 2**23 * 0.9:
            break
    gp = T.GadgetParameters(n, k, m_bar, base, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = psf.trap_gen(100 + case)
    orc = oracle.PSFPerturbation(oracle.GadgetParams(n, k, m_bar, base, q), r, s)
    assert orc.trap_gen(100 + case) == 0                                  # key generation from the same seed: A, R bitwise, the factor within rounding
    assert (A == orc.A).all() and (R == orc.R).all(), (n, q, base, k, m_bar)
    np.testing.assert_allclose(Lp, orc.L_packed, rtol=0, atol=1e-9 * np.abs(orc.L_packed).max())
    orc.load_key(A, R, Lp)
    u = oracle.uniform_targets(case, B, n, q)
    first = int(rng.integers(0, 2**40))
    e = psf.samp_p(u, seed=7 + case, first_index=first)
    assert psf.last_status() == 0
    assert (e == orc.samp_p(7 + case, u, first_index=first)).all(), (n, q, base, k, m_bar, r, s, B)
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all()
    d = psf.samp_d(seed=3 + case, B=min(B, 9), first_index=first)          # mp_perturbation.rs:264-267
    assert (d == orc.samp_d(3 + case, B=min(B, 9), first_index=first)).all()
    assert psf.check_domain(d).all() and (psf.f_a(d) == orc.f_a(d)).all()
    psf.close()


# 752, 786, 988, 1174: found by tools/fuzz_configs.py (3 600 draws) -- moduli of 2^57 .. 2^60 over n >= 6: centres beyond 2^62 (now PSF_ERR_SAMPLER on both sides, it was
# undefined behaviour in the oracle) and first-pass coefficients beyond 2^53 (the oracle recombined them exactly, the device as the doubles the walk uses: both now the latter)
@pytest.mark.parametrize("case", list(range(32)) + [752, 786, 988, 1174])
def test_gpv_random_configuration(oracle, case):
    import tools_amd as T
    from tools_amd import _ffi
    rng = np.random.default_rng(2000 + case)
    q = draw_modulus(rng)
    n = int(rng.integers(2, 40 if q < 2**24 else 12))
    s = float(rng.choice(GPV_S_MENU)) * (1.0 if q < 2**30 else 4.0)
    B = int(rng.choice([1, 3, 4, 5, 8, 9, 64, 130]))
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    A, (bt, gt) = psf.trap_gen(200 + case)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.load_key(A, bt, gt) == 0
    assert orc.two_pass == psf.two_pass                 # the same rule on both sides (q sqrt(n) > 2^13 s)
    d = psf.samp_d(seed=4 + case, B=3, first_index=5)   # gpv.rs:113-116
    assert (d == orc.samp_d(4 + case, B=3, first_index=5)).all()
    if psf.check_domain(d).all():
        assert (psf.f_a(d) == orc.f_a(d)).all()
    u = oracle.uniform_targets(case, B, n, q)
    first = int(rng.integers(0, 2**40))
    try:
        e = psf.samp_p(u, seed=9 + case, first_index=first)
    except T.PsfError as err:
        # Two corners of the parameter space end in PSF_ERR_SAMPLER by contract, and the oracle must end there on the same inputs: a Gaussian far below the
        # smoothing parameter (s / ||b~_i|| ~ 0.25: a half-integral centre is accepted with probability e^-25, the attempt cap of 65 536 ends the draw
        # the reference would spend 10^11 iterations on), and a modulus near 2^61 over a basis with a Gram-Schmidt vector of norm ~0.05, whose coefficient
        # q sqrt(n) / ||b~_i|| leaves the 2^53 range of the walk (the reference carries it as a big integer).
        assert err.status == _ffi.ERR_SAMPLER, (n, q, s, B)
        with pytest.raises(RuntimeError, match="oracle status 6"):
            orc.samp_p(9 + case, u, first_index=first)
        psf.close()
        return
    assert (e == orc.samp_p(9 + case, u, first_index=first)).all(), (n, q, s, B)
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()
    psf.close()


@pytest.mark.parametrize("case", range(16))
def test_ring_random_configuration(oracle, case):
    import tools_amd as T
    rng = np.random.default_rng(3000 + case)
    n = int(2 ** rng.integers(2, 7))
    q = int(rng.choice([257, 3329, 7681, 12289, 2**16 + 1, 1073741789]))
    s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * float(rng.choice([4.0, 8.0]))
    B = int(rng.choice([1, 2, 5, 8, 17, 70]))
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
    a, (r, e0) = psf.trap_gen(300 + case)
    _, _, _, bt, gt = psf.export_key()
    orc = oracle.PSFGPVRing(oracle.gadget_params_ring_default(n, q), s, 1.005)
    assert orc.load_key(a, r, e0, gso_t=gt) == 0
    u = oracle.uniform_targets(case, B, n, q)
    first = int(rng.integers(0, 2**40))
    sg = psf.samp_p(u, seed=11 + case, first_index=first)
    assert (sg == orc.samp_p(11 + case, u, first_index=first)).all(), (n, q, s, B)
    if psf.check_domain(sg).all():
        assert (psf.f_a(sg) == u).all()
    psf.close()


LEGACY = {"PSF_TRMM_STREAM_MAX": "0", "PSF_FUSED_MAX": "0", "PSF_GADGET_WAVE": "0", "PSF_GADGET_WAVE16": "0", "PSF_COMPACT_D": "0", "PSF_ZQ_FOLD128": "0"}


@pytest.mark.parametrize("case", range(0, 64, 3))
def test_perturbation_random_configuration_on_the_batch_kernels(oracle, monkeypatch, exp_lib, case):
    """Since round 4 every batch of this file's menus (<= 300 preimages) takes the single-call kernels (k_samp_p_small, k_trmm_stream, k_gadget_wave*, the
    128-bit Z_q fold).  The same seeded draws with those switched off keep the batch kernels (k_trmm_f64_big, k_gadget_queue, the per-class fold, the
    chunk-stream normals) under the random shapes they were found correct on."""
    for k, v in LEGACY.items():
        monkeypatch.setenv(k, v)
    test_perturbation_random_configuration(oracle, case)
