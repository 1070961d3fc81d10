"""The reference's invariant tests and distribution checks, run on the CPU oracle."""
import math

import numpy as np
import pytest


def test_trapdoor_relation(oracle):
    # gadget_classical.rs:363-385 / gadget_default.rs:137-152: A [R; I] = G mod q
    for n, q in [(5, 32), (10, 1024), (7, 97)]:
        gp = oracle.gadget_params_default(n, q)
        w = gp.n * gp.k
        a_bar = oracle.sample_a_bar(1, n, gp.m_bar, q)
        R = oracle.sample_r(1, gp.m_bar, w)
        assert set(np.unique(R)) <= {-1, 0, 1}                      # trapdoor_distribution.rs:134-169
        A = oracle.gen_trapdoor(gp, a_bar, R)
        T = np.vstack([R.astype(object), np.eye(w, dtype=object)])
        G = oracle.gen_gadget_mat(n, gp.k, 2).astype(object)
        assert (((A.astype(object) @ T) - G) % q == 0).all()
    # with an invertible (unit upper triangular) tag, gadget_classical.rs:389-414
    n, q = 6, 32
    gp = oracle.gadget_params_default(n, q)
    rng = np.random.default_rng(0)
    tag = np.triu(rng.integers(0, q, size=(n, n)), 1) + np.eye(n, dtype=np.int64)
    a_bar = oracle.sample_a_bar(2, n, gp.m_bar, q)
    R = oracle.sample_r(2, gp.m_bar, n * gp.k)
    A = oracle.gen_trapdoor(gp, a_bar, R, tag=tag.astype(np.uint64))
    T = np.vstack([R.astype(object), np.eye(n * gp.k, dtype=object)])
    G = oracle.gen_gadget_mat(n, gp.k, 2).astype(object)
    assert (((A.astype(object) @ T) - tag.astype(object) @ G) % q == 0).all()


def test_short_basis_is_in_lattice(oracle):
    # short_basis_classical.rs:128-188: A S_A = 0 mod q, S_A full rank
    for n, q in [(3, 16), (4, 23)]:
        gp = oracle.gadget_params_default(n, q)
        w = gp.n * gp.k
        a_bar = oracle.sample_a_bar(5, n, gp.m_bar, q)
        R = oracle.sample_r(5, gp.m_bar, w)
        A = oracle.gen_trapdoor(gp, a_bar, R)
        S = oracle.gen_short_basis_for_trapdoor(gp, A, R)
        assert ((A.astype(object) @ S.astype(object)) % q == 0).all()
        assert abs(np.linalg.det(S.astype(float))) > 0.5


@pytest.mark.parametrize("n,q,r,s", [(5, 256, math.log2(5), 25.0), (6, 128, math.log2(6), 25.0), (8, 64, 3.0, 25.0)])
def test_psf_perturbation_invariants(oracle, n, q, r, s):
    # mp_perturbation.rs:416-464
    psf = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    assert psf.trap_gen(3) == 0
    for i in range(5):
        assert psf.check_domain(psf.samp_d(10 + i)).all()
    ds = psf.samp_d(2, B=3)
    u = psf.f_a(ds)
    A = psf.A.astype(object)
    assert (u.astype(object) == (ds.astype(object) @ A.T) % q).all()          # f_a == a * sigma
    e = psf.samp_p(4, u)
    assert psf.check_domain(e).all()
    assert (psf.f_a(e) == u).all()
    # batch (grouped) path == single-preimage trace path
    for b in range(3):
        assert (psf.samp_p_trace(4, b, u[b])["e"] == e[b]).all()


def test_not_positive_definite(oracle):
    psf = oracle.PSFPerturbation(oracle.gadget_params_default(8, 64), 3.0, 3.0)
    assert psf.trap_gen(1) == oracle.ERR_NOT_PD                          # mp_perturbation.rs:109-110


def test_check_domain_cases(oracle):
    # mp_perturbation.rs:517-554
    psf = oracle.PSFPerturbation(oracle.gadget_params_default(8, 128), 3.0, 25.0)
    m = psf.m
    assert psf.check_domain(np.zeros(m, dtype=np.int64)).all()
    assert psf.check_domain(np.full(m, 25, dtype=np.int64)).all()
    assert not psf.check_domain(np.zeros(m - 1, dtype=np.int64)).any()
    assert not psf.check_domain(np.zeros(m + 1, dtype=np.int64)).any()
    big = np.zeros(m, dtype=np.int64)
    big[0] = 25 * m
    assert not psf.check_domain(big).any()
    with pytest.raises(AssertionError):
        psf.f_a(big)


@pytest.mark.parametrize("s,c", [(4.5, 0.3),          # narrow words, fractional centre
                                 (4.5, 2.0),          # integer centre: one more candidate
                                 (300.0, -7.25),      # narrow words near the 4096-candidate limit (Lemire rejections ~1 %)
                                 (400.0, 0.5)])       # wide (32-bit) words
def test_sample_z_distribution(oracle, s, c):
    # chi-square of D_{Z,s,c} against rho_s(x-c), binned so that every cell expects >= 8 draws -- the reference has no such
    # test
    N = 60000
    xs = np.array([oracle.sample_z(99, oracle.TAG_PERTURB, i, 5, c, s) for i in range(N)])
    lo, hi = math.ceil(c) - math.ceil(6 * s), math.floor(c) + math.floor(6 * s)
    assert xs.min() >= lo and xs.max() <= hi
    support = np.arange(lo, hi + 1)
    prob = np.exp(-math.pi * (support - c) ** 2 / s**2)
    prob /= prob.sum()
    width = max(1, int(s / 6))
    edges = np.arange(lo, hi + 1 + width, width)
    obs, _ = np.histogram(xs, bins=edges)
    exp = np.array([prob[(support >= a) & (support < b)].sum() for a, b in zip(edges[:-1], edges[1:])]) * N
    keep = exp >= 8
    chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
    dof = keep.sum() - 1
    assert obs[~keep].sum() <= 8 * (~keep).sum() + 30
    assert chi2 < dof + 5 * math.sqrt(2 * dof), (chi2, dof)
    assert abs(xs.mean() - c) < 5 * s / math.sqrt(2 * math.pi * N) + 0.02


def test_normal_distribution(oracle):
    from scipy import stats
    d = np.array([oracle.sample_normal(7, i, 3) for i in range(40000)])
    assert stats.kstest(d, "norm").pvalue > 1e-3
    assert abs(d.mean()) < 0.03 and abs(d.std() - 1) < 0.02


def test_streams_are_independent_of_batching(oracle):
    psf = oracle.PSFPerturbation(oracle.gadget_params_default(5, 32), 2.5, 25.0)
    assert psf.trap_gen(8) == 0
    u = oracle.uniform_targets(1, 40, 5, 32)
    full = psf.samp_p(77, u, first_index=100)
    part = psf.samp_p(77, u[17:23], first_index=117)
    assert (part == full[17:23]).all()


def test_a_draw_that_ends_at_the_attempt_cap_is_reported(oracle):
    """The reference's sample_z loops until it accepts; the contract caps a draw at 65 536 attempts (psf_rng.hpp kMaxAttempts), ends it with the nearest integer
    and REPORTS it: the device raises PSF_ERR_SAMPLER, and so must the oracle's sampling entry points (found by tests/test_gpu_random_configs.py: the oracle
    used to return the same fallback value with status 0).  A width of 0.2 around a half-integer accepts with probability e^-19 per attempt."""
    import ctypes as C
    from oracle.oracle import lib
    lib().orc_sample_z_cap_hits.restype = C.c_ulong
    before = lib().orc_sample_z_cap_hits()
    z = oracle.sample_z(1, 3, 0, 0, 0.5, 0.2)
    assert z in (0, 1) and lib().orc_sample_z_cap_hits() == before + 1
    assert oracle.sample_z(1, 3, 0, 0, 0.5, 3.0) is not None and lib().orc_sample_z_cap_hits() == before + 1      # an ordinary draw does not count
    # through an entry point: a PSFGPV whose Gaussian is far below the smoothing parameter
    n, q, s = 6, 257, 0.3
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.trap_gen(1) == 0
    u = oracle.uniform_targets(1, 2, n, q)
    import pytest
    with pytest.raises(RuntimeError, match="oracle status 6"):
        orc.samp_p(5, u)
