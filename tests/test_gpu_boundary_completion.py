"""Round 3 boundary completion (VERDICT r02 "What's missing" #2-#4):
  * compute_sqrt_sigma_2 with a general covariance (mp_perturbation.rs:111 takes any `mat_sigma: &MatQ`);
  * trapdoors drawn by the caller's own TrapdoorDistribution (gadget_classical.rs:62-64, gadget_ring.rs:69-70: `params.distribution.sample`);
  * PSFGPVRing over a modulus beyond 2^31 (GadgetParametersRing carries any ModulusPolynomialRingZq, gadget_parameters.rs:73-81);
  * a verifier's handle: f_a with the public matrix alone (PSF::f_a takes `a`, mp_perturbation.rs:366)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


def test_general_covariance_matches_the_oracle_and_shapes_the_preimages(T, oracle):
    n, q, r, s = 6, 128, 3.0, 30.0
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = psf.trap_gen(3)
    m = psf.m
    # a non-spherical target: two variances, alternating, plus a weak correlation between neighbours
    d = np.where(np.arange(m) % 2 == 0, 30.0**2, 42.0**2)
    Sigma = np.diag(d)
    for i in range(m - 1):
        Sigma[i, i + 1] = Sigma[i + 1, i] = 60.0
    psf.compute_sqrt_sigma_2(sigma=Sigma)
    _, (_, L_dev, _) = psf.export_key()
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    rc, L_ref = orc.compute_sqrt_sigma_2_dense(R, Sigma[np.tril_indices(m)])
    assert rc == 0
    np.testing.assert_allclose(L_dev, L_ref, rtol=0, atol=1e-10 * np.abs(L_ref).max())
    # L L^t = Sigma_2 = (r^2 / 2 pi) (Sigma - (b^2 + 1) T T^t - I)
    Lfull = np.zeros((m, m)); Lfull[np.tril_indices(m)] = L_dev
    Tm = np.vstack([R.astype(np.float64), np.eye(n * gp.k)])
    S2 = (r * r / (2 * math.pi)) * (Sigma - 5.0 * Tm @ Tm.T - np.eye(m))
    np.testing.assert_allclose(Lfull @ Lfull.T, S2, rtol=0, atol=1e-9 * np.abs(S2).max())
    # sampling with the new factor: bitwise the oracle's, valid, and the preimages carry the requested covariance r^2 Sigma / 2 pi
    orc.load_key(A, R, L_dev)
    u = oracle.uniform_targets(4, 6, n, q)
    e = psf.samp_p(u, seed=8)
    assert (e == orc.samp_p(8, u)).all()
    B = 20000
    ub = np.tile(u[:1], (B, 1))
    eb = psf.samp_p(ub, seed=21).astype(np.float64)
    assert ((A.astype(object) @ eb[:50].astype(np.int64).astype(object).T).T % q == ub[:50].astype(object)).all()
    var = eb.var(axis=0)
    want = (r * r / (2 * math.pi)) * d
    assert np.abs(var / want - 1).max() < 0.07, (var / want).min()          # sqrt(2 / B) = 1 %
    assert abs((var[0::2] / want[0::2]).mean() - 1) < 0.01 and abs((var[1::2] / want[1::2]).mean() - 1) < 0.01
    cov01 = np.mean([np.cov(eb[:, i], eb[:, i + 1])[0, 1] for i in range(0, m - 1)])
    assert abs(cov01 / ((r * r / (2 * math.pi)) * 60.0) - 1) < 0.1
    # the spherical entry point still does what it did, and a covariance that is too small is refused as the reference panics (:109-110)
    psf.compute_sqrt_sigma_2(35.0)
    with pytest.raises(T.PsfError):
        psf.compute_sqrt_sigma_2(sigma=np.eye(m) * 4.0)


def test_load_key_variants(T, oracle):
    n, q, r, s = 8, 64, 3.0, 25.0
    gp = T.GadgetParameters.init_default(n, q)
    full = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = full.trap_gen(2)
    u = oracle.uniform_targets(1, 5, n, q)
    e = full.samp_p(u, seed=4)
    # (A, R): the factor is recomputed exactly as trap_gen computes it
    again = T.PSFPerturbation(gp, r, s)
    again.load_key(A, R)
    assert (again.samp_p(u, seed=4) == e).all()
    _, (_, L2, _) = again.export_key()
    assert (L2 == Lp).all()
    # (A,): the verifier's handle
    ver = T.PSFPerturbation(gp, r, s)
    with pytest.raises(T.PsfError):
        ver.f_a(e)                                        # nothing installed yet
    ver.load_key(A)
    assert (ver.f_a(e) == u).all() and ver.check_domain(e).all()
    with pytest.raises(T.PsfError) as ei:
        ver.samp_p(u, seed=4)
    assert ei.value.status == T._ffi.ERR_NO_KEY
    with pytest.raises(T.PsfError):
        ver.compute_sqrt_sigma_2(30.0)                    # no R to build Sigma_2 from


def test_load_trapdoor_installs_a_trapdoor_without_computing_a_factor(T, oracle):
    """psfp_load_trapdoor: (A, R) with no factor and no Cholesky (compute_sqrt_sigma_2 of the reference is a pure function of mat_r and mat_sigma,
    mp_perturbation.rs:111; ADVICE r03: psfp_load_key(A, R, NULL) ran a full factorisation with the handle's own s first, and refused an R whose
    Sigma_2(s) is not positive definite although the caller's covariance is fine)."""
    n, q, r, s = 8, 64, 3.0, 25.0
    gp = T.GadgetParameters.init_default(n, q)
    full = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = full.trap_gen(2)
    u = oracle.uniform_targets(1, 5, n, q)
    psf = T.PSFPerturbation(gp, r, s)
    psf.load_trapdoor(R, A)
    with pytest.raises(T.PsfError) as ei:
        psf.samp_p(u, seed=4)                               # no factor yet
    assert ei.value.status == T._ffi.ERR_NO_KEY
    assert (psf.f_a(full.samp_p(u, seed=4)) == u).all()     # the public matrix works
    psf.compute_sqrt_sigma_2(s)                             # completes the key: the factor trap_gen computes
    assert (psf.samp_p(u, seed=4) == full.samp_p(u, seed=4)).all()
    # a handle whose own s is too small for this R is no obstacle: nothing is factored at install time
    tiny = T.PSFPerturbation(gp, r, 1.5)
    with pytest.raises(T.PsfError):
        tiny.load_key(A, R)                                 # load_key(A, R) factors Sigma_2(1.5): not positive definite
    tiny.load_trapdoor(R, A)
    tiny.compute_sqrt_sigma_2(s)
    _, (_, L3, _) = tiny.export_key()
    assert (L3 == Lp).all()
    # R alone (A = NULL): compute_sqrt_sigma_2 needs no public matrix; samp_p still does
    bare = T.PSFPerturbation(gp, r, s)
    bare.load_trapdoor(R)
    bare.compute_sqrt_sigma_2(s)
    assert (bare.export_sqrt_sigma2_rows(0, bare.m) == Lp).all()
    with pytest.raises(T.PsfError) as ei:
        bare.samp_p(u, seed=4)
    assert ei.value.status == T._ffi.ERR_NO_KEY


def test_trapdoor_from_the_callers_own_distribution(T, oracle):
    """R drawn by the caller (here: a sparse {-2..2} distribution, not PlusMinusOneZero): A = [A_bar | H G - A_bar R] from the device equals the
    oracle's, A [R; I] = H G, and the pair works as a key."""
    n, q, r, s = 6, 157, 4.0, 60.0
    gp = T.GadgetParameters.init_default(n, q)
    rng = np.random.default_rng(3)
    w = n * gp.k
    a_bar = rng.integers(0, q, size=(n, gp.m_bar), dtype=np.uint64)
    tag = rng.integers(0, q, size=(n, n), dtype=np.uint64)
    R = rng.choice(np.array([-2, -1, 0, 0, 0, 1, 2]), size=(gp.m_bar, w)).astype(np.int64)
    A = T.gadget.gen_trapdoor_with_r(gp, a_bar, R, tag=tag)
    ogp = oracle.gadget_params_default(n, q)
    assert (A == oracle.gen_trapdoor(ogp, a_bar, R.astype(np.int8), tag=tag)).all()
    Tm = np.vstack([R.astype(object), np.eye(w, dtype=object)])
    G = np.zeros((n, w), dtype=object)
    for i in range(n):
        for t in range(gp.k):
            G[i, i * gp.k + t] = pow(int(gp.base), t, q)
    assert ((A.astype(object) @ Tm) % q == (tag.astype(object) @ G) % q).all()
    with pytest.raises(T.PsfError):
        T.gadget.gen_trapdoor_with_r(gp, a_bar, R * 100, tag=tag)        # |R_ij| > 127: not an int8 trapdoor
    # identity tag: a usable PSFPerturbation key
    A1 = T.gadget.gen_trapdoor_with_r(gp, a_bar, R)
    psf = T.PSFPerturbation(gp, r, s)
    psf.load_key(A1, R.astype(np.int8))
    u = oracle.uniform_targets(5, 9, n, q)
    e = psf.samp_p(u, seed=12)
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all()
    orc = oracle.PSFPerturbation(ogp, r, s)
    _, (_, Lp, _) = psf.export_key()
    orc.load_key(A1, R.astype(np.int8), Lp)
    assert (e == orc.samp_p(12, u)).all()


def _polymul(x, y, n, q):
    acc = [0] * n
    for i in range(n):
        for j in range(n):
            if i + j >= n:
                acc[i + j - n] -= int(x[i]) * int(y[j])
            else:
                acc[i + j] += int(x[i]) * int(y[j])
    return [v % q for v in acc]


@pytest.mark.parametrize("n,q", [(4, 2**45), (8, 2**61 - 1), (4, 2**31 + 11), (16, 2**40 + 15)])
def test_ring_psf_over_large_moduli(T, oracle, n, q):
    s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
    gp = T.GadgetParametersRing.init_default(n, q)
    psf = T.PSFGPVRing(gp, s, 1.005)
    a, (r, e) = psf.trap_gen(9)
    ogp = oracle.gadget_params_ring_default(n, q)
    oa, orr, oe = oracle.ring_trap_gen(ogp, 1.005, 9)
    assert (a == oa).all() and (r == orr).all() and (e == oe).all()
    for j in range(gp.k):                                  # a_0 e_j + a_1 r_j + a_{2+j} = base^j in R_q (gadget_ring.rs:190-211)
        lhs = [(x + y + int(z)) % q for x, y, z in zip(_polymul(a[0], e[j], n, q), _polymul(a[1], r[j], n, q), a[2 + j])]
        assert lhs == [pow(2, j, q)] + [0] * (n - 1)
    _, _, _, bt, gt = psf.export_key()
    assert (bt == oracle.ring_short_basis_t(ogp, oa, orr, oe)).all()
    orc = oracle.PSFGPVRing(ogp, s, 1.005)
    orc.load_key(a, r, e, gso_t=gt)
    u = oracle.uniform_targets(3, 4, n, q)
    sg = psf.samp_p(u, seed=5)
    assert (sg == orc.samp_p(5, u)).all()
    assert psf.check_domain(sg).all() and (psf.f_a(sg) == u).all()
    for b in range(2):                                     # f_a in R_q with big integers: sum_j a_j sigma_j = u
        acc = [0] * n
        for j in range(gp.k + 2):
            acc = [(x + y) % q for x, y in zip(acc, _polymul(a[j], sg[b, j], n, q))]
        assert acc == [int(v) for v in u[b]]


def test_ring_trapdoor_from_the_callers_own_r_e(T, oracle):
    n, q = 8, 3329
    gp = T.GadgetParametersRing.init_default(n, q)
    rng = np.random.default_rng(8)
    a_bar = rng.integers(0, q, size=n, dtype=np.uint64)
    r = rng.integers(-3, 4, size=(gp.k, n)).astype(np.int64)
    e = rng.integers(-3, 4, size=(gp.k, n)).astype(np.int64)
    a = T.gadget.gen_trapdoor_ring_lwe_with(gp, a_bar, r, e)
    assert int(a[0][0]) == 1 and (a[0][1:] == 0).all() and (a[1] == a_bar).all()
    for j in range(gp.k):
        lhs = [(x + y + int(z)) % q for x, y, z in zip(_polymul(a[0], e[j], n, q), _polymul(a[1], r[j], n, q), a[2 + j])]
        assert lhs == [pow(2, j, q)] + [0] * (n - 1)
    s = 4 * ((2 * 2 * 3.0 * math.sqrt(n) + 1) * 2) * 4        # a wider trapdoor needs a wider s (compute_s with s_td = 3)
    psf = T.PSFGPVRing(gp, s, 3.0)
    psf.load_key(a, r, e)
    u = oracle.uniform_targets(3, 6, n, q)
    sg = psf.samp_p(u, seed=2)
    assert psf.check_domain(sg).all() and (psf.f_a(sg) == u).all()
