"""The labelled opt-in PSFP_FLAG_STRUCTURED_SQRT: x = B d with the block factor B = [[L_1, -g R], [0, h I]] of Sigma_2 instead of its dense Cholesky
factor (include/psf_mi355x.h).  Same distribution as mp_perturbation.rs:315 (any square root of Sigma_2 will do), a different algorithm: so the
tests are (1) bitwise parity of every stage against the oracle's restatement of the structured contract, (2) B B^t = Sigma_2 checked from R alone,
(3) the end-to-end distribution checks the dense path has to pass (spherical output, multi-panel), (4) the PSF invariants A e = u / check_domain."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = [(8, 64, 3.0, 25.0), (15, 157, float(np.log2(15)), 40.0), (4, 2**60, 2.0, 70.0), (8, 64, 100.0, 25.0), (32, 256, 5.0, 120.0)]


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


@pytest.mark.parametrize("n,q,r,s", CONFIGS)
def test_structured_stage_parity(T, oracle, n, q, r, s):
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s, structured=True)
    A, (R, L1, (Sk, gso)) = psf.trap_gen(11)
    mb = psf.m_bar
    assert L1.size == mb * (mb + 1) // 2
    dense = T.PSFPerturbation(gp, r, s)
    A2, (R2, _, _) = dense.trap_gen(11)
    assert (A == A2).all() and (R == R2).all()                       # the same (A, R) as the dense handle: only the perturbation sampler differs
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s, with_L=False)
    orc.load_key(A, R)
    rc, L1_ref = orc.structured_sqrt(R, s)
    assert rc == 0
    np.testing.assert_allclose(L1, L1_ref, rtol=0, atol=1e-9 * np.abs(L1_ref).max())     # blocked device Cholesky vs unblocked recurrence
    B = 5
    u = oracle.uniform_targets(3, B, n, q)
    st = psf.samp_p_stages(u, seed=77, first_index=1000)
    for b in range(B):
        tr = orc.samp_p_structured_trace(L1, s, 77, 1000 + b, u[b])             # the device's own L_1: same key on both sides
        assert (st["d"][b].view(np.uint64) == tr["d"].view(np.uint64)).all(), "normals / fixed-point normals differ"
        assert (st["x"][b].view(np.uint64) == tr["x"].view(np.uint64)).all(), "centres differ"
        for key in ("p", "v", "z", "e"):
            assert (st[key][b] == tr[key]).all(), key
    e = psf.samp_p(u, seed=5)
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all()
    # key round trip in structured form
    again = T.PSFPerturbation(gp, r, s, structured=True)
    again.load_key(A, R, L1)
    assert (again.samp_p(u, seed=5) == e).all()


def test_structured_factor_squares_to_sigma2(T):
    """B B^t y = Sigma_2 y for probe vectors, Sigma_2 y evaluated from R alone (mp_perturbation.rs:125-135)."""
    n, q, r, s = 32, 256, 5.0, 120.0
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s, structured=True)
    A, (R, L1p, _) = psf.trap_gen(5)
    mb, w, m = psf.m_bar, psf.w, psf.m
    L1 = np.zeros((mb, mb))
    L1[np.tril_indices(mb)] = L1p
    c = r * r / (2 * math.pi)
    kappa, alpha = 5.0, s * s - 1.0
    beta = alpha - kappa
    g, h = math.sqrt(c) * kappa / math.sqrt(beta), math.sqrt(c * beta)
    Bm = np.zeros((m, m))
    Bm[:mb, :mb] = L1
    Bm[:mb, mb:] = -g * R
    Bm[mb:, mb:] = h * np.eye(w)
    Tm = np.vstack([R.astype(np.float64), np.eye(w)])
    sigma2 = c * ((s * s - 1.0) * np.eye(m) - kappa * (Tm @ Tm.T))
    err = np.abs(Bm @ Bm.T - sigma2).max() / np.abs(sigma2).max()
    assert err < 1e-12, err


def test_structured_preimages_are_spherical(T):
    n, q, r, s, B = 32, 256, 5.0, 120.0, 30000
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s, structured=True)
    psf.trap_gen(5, export=False)
    u = np.tile((np.arange(n, dtype=np.uint64) * 37 + 11) % q, (B, 1))
    e = psf.samp_p(u, seed=13).astype(np.float64)
    sigma = s * r / math.sqrt(2 * math.pi)
    std = e.std(axis=0)
    assert np.abs(std / sigma - 1).max() < 0.03, (std.min(), std.max(), sigma)
    for p0 in range(0, psf.m, 128):
        pan = (std[p0:p0 + 128] ** 2).mean() / sigma**2
        assert abs(pan - 1) < 0.006, (p0, pan)
    corr = np.corrcoef(e.T)
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() < 6.2 / math.sqrt(B), np.abs(corr).max()
    nrm2 = (e**2).sum(axis=1)
    assert abs(nrm2.mean() / (psf.m * sigma**2) - 1) < 0.005


def test_structured_not_pd_is_reported(T):
    # beta = s^2 - 1 - (b^2 + 1) <= 0
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(4, 16), 2.0, 2.0, structured=True)
    with pytest.raises(T.PsfError) as ei:
        psf.trap_gen(1)
    assert ei.value.status == 2


def test_structured_factor_for_another_covariance_is_refused(T):
    """The structured factor's constants g, h follow from s and are rebuilt from the handle's s when a key is loaded (an exported key carries only L_1), so a
    factor for another s_cov could not survive an export / load round trip: compute_sqrt_sigma_2 refuses it instead of pairing L_1(s_cov) with g, h(s)."""
    from tools_amd import _ffi
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(8, 64), 3.0, 25.0, structured=True)
    psf.trap_gen(3)
    psf.compute_sqrt_sigma_2(25.0)                       # the handle's own s: allowed, same factor
    with pytest.raises(T.PsfError) as ei:
        psf.compute_sqrt_sigma_2(35.0)
    assert ei.value.status == _ffi.ERR_UNSUPPORTED
    u = np.zeros((3, 8), dtype=np.int64)
    e = psf.samp_p(u, seed=4)                            # the key in the handle is still the one for s
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all()
    psf.close()


@pytest.mark.parametrize("case", range(64))
def test_structured_random_configuration(T, oracle, case):
    """seeded random shapes for the opt-in structured square root: odd n, moduli of every kind, widths from just above the positive-definiteness bound, ragged batches --
    every stage of a few rows bitwise against the oracle's restatement, the whole batch through the invariants"""
    rng = np.random.default_rng(8000 + case)
    while True:                                              # a draw outside the documented domain bound (s r sqrt(m) < 2^23) is redrawn, not skipped: 64 cases run
        n = int(rng.integers(2, 40))
        kind = int(rng.integers(0, 3))
        q = int(2 ** rng.integers(4, 61)) if kind == 0 else (int(rng.choice([257, 3329, 12289, 1073741789, 2**61 - 1])) if kind == 1 else int(rng.integers(17, 2**20)) | 1)
        k = int(math.ceil(math.log2(q)))
        r = float(rng.choice([2.0, 3.0, 4.5, 30.0]))
        m_bar = n * k + int(rng.integers(0, 40))
        s = r * math.sqrt(5.0) * (math.sqrt(m_bar) + math.sqrt(n * k) + 4.0) * float(rng.choice([1.2, 2.0]))
        if s * r * math.sqrt(m_bar + n * k) < 2**23 * 0.9:
            break
    B = int(rng.choice([1, 5, 127, 129, 256, 300]))
    gp = T.GadgetParameters(n, k, m_bar, 2, q)
    psf = T.PSFPerturbation(gp, r, s, structured=True)
    A, (R, L1, _) = psf.trap_gen(300 + case)
    orc = oracle.PSFPerturbation(oracle.GadgetParams(n, k, m_bar, 2, q), r, s, with_L=False)
    orc.load_key(A, R)
    u = oracle.uniform_targets(case, B, n, q)
    first = int(rng.integers(0, 2**40))
    st = psf.samp_p_stages(u, seed=21 + case, first_index=first)
    for b in sorted(set([0, B // 2, B - 1])):
        tr = orc.samp_p_structured_trace(L1, s, 21 + case, first + b, u[b])
        assert (st["x"][b].view(np.uint64) == tr["x"].view(np.uint64)).all(), (n, q, m_bar, r, s, B, b)
        for key in ("p", "v", "z", "e"):
            assert (st[key][b] == tr[key]).all(), (key, n, q, m_bar, r, s, B, b)
    assert (psf.f_a(st["e"]) == u).all() and psf.check_domain(st["e"]).all()
    psf.close()
