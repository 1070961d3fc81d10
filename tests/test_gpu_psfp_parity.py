"""GPU parity of the PSFPerturbation path (HIP, through the C ABI) against the CPU oracle.

Bit-exact for every integer stage and, because the summation orders are part of the contract, also for the
f64 centres x = sqrt(Sigma_2) d.  The Cholesky factor itself (setup, computed by a different blocked
algorithm on the device) is compared with a stated tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = [  # (n, q, r, s) -- reference test / doc sizes: mp_perturbation.rs:43-47, :417, :434 ; benches/psf.rs:52,79
    (8, 64, 3.0, 25.0),
    (5, 256, np.log2(5), 25.0),
    (6, 128, np.log2(6), 25.0),
    (10, 128, np.log2(10), 40.0),
    (15, 157, np.log2(15), 40.0),   # prime modulus: S_k carries the digit column of q
    (8, 128, 3.0, 30.0),
    (4, 2**60, 2.0, 70.0),          # C5's modulus: two-limb / 8-digit Z_q arithmetic, k = 60 gadget
    (3, 2**61 - 1, 2.0, 70.0),      # large prime modulus: digit column in S_k, wide non-power-of-two reduction
    (4, 1073741789, 2.0, 50.0),     # C3's prime alternative
    (8, 64, 100.0, 25.0),           # wide gadget Gaussian: |z| > 127 occurs, exercising the hi byte plane of z
    (8, 64, 400.0, 25.0),           # > 4096 candidates per SampleZ: the 32-bit attempt words ("wide")
]


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


def make_pair(T, oracle, n, q, r, s, seed=11):
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, (Sk, gso)) = psf.trap_gen(seed)
    ogp = oracle.gadget_params_default(n, q)
    orc = oracle.PSFPerturbation(ogp, r, s)
    return psf, orc, (A, R, Lp, Sk, gso)


@pytest.mark.parametrize("n,q,r,s", CONFIGS)
def test_trap_gen_parity(T, oracle, n, q, r, s):
    psf, orc, (A, R, Lp, Sk, gso) = make_pair(T, oracle, n, q, r, s)
    assert orc.trap_gen(11) == 0
    assert (A == orc.A).all(), "A = [A_bar | G - A_bar R] differs"
    assert (R == orc.R).all(), "R differs"
    assert (Sk == orc.Sk).all()
    np.testing.assert_allclose(gso, orc.Sk_gso, rtol=0, atol=1e-13)
    # Cholesky factor: different (blocked) algorithm on the device -> tolerance, relative to the largest entry
    scale = np.abs(orc.L_packed).max()
    np.testing.assert_allclose(Lp, orc.L_packed, rtol=0, atol=1e-9 * scale)
    # trapdoor relation A [R; I] = G mod q (gadget_classical.rs:363-385)
    Tm = np.vstack([R.astype(object), np.eye(psf.w, dtype=object)])
    G = oracle.gen_gadget_mat(n, psf.k, 2).astype(object)
    assert (((A.astype(object) @ Tm) - G) % q == 0).all()


@pytest.mark.parametrize("n,q,r,s", CONFIGS)
def test_samp_p_stage_parity(T, oracle, n, q, r, s):
    psf, orc, (A, R, Lp, Sk, gso) = make_pair(T, oracle, n, q, r, s)
    orc.load_key(A, R, Lp)            # same key material on both sides
    B = 5
    u = oracle.uniform_targets(3, B, n, q)
    st = psf.samp_p_stages(u, seed=77, first_index=1000)
    for b in range(B):
        tr = orc.samp_p_trace(77, 1000 + b, u[b])
        assert (st["d"][b].view(np.uint64) == tr["d"].view(np.uint64)).all(), "normals differ"
        assert (st["x"][b].view(np.uint64) == tr["x"].view(np.uint64)).all(), "centres x = sqrt(Sigma_2) d differ"
        assert (st["p"][b] == tr["p"]).all(), "perturbation differs"
        assert (st["v"][b] == tr["v"]).all(), "syndrome v = u - A p differs"
        assert (st["z"][b] == tr["z"]).all(), "gadget preimage differs"
        assert (st["e"][b] == tr["e"]).all(), "preimage differs"
    if r >= 100:
        assert np.abs(st["z"]).max() > 127, "this configuration is meant to exercise the hi plane"


@pytest.mark.parametrize("n,q,r,s,B", [(8, 64, 3.0, 25.0, 1), (8, 64, 3.0, 25.0, 300), (15, 157, np.log2(15), 40.0, 130),
                                       (8, 64, 100.0, 25.0, 140)])
def test_samp_p_batch_parity_and_invariants(T, oracle, n, q, r, s, B):
    psf, orc, (A, R, Lp, Sk, gso) = make_pair(T, oracle, n, q, r, s)
    orc.load_key(A, R, Lp)
    u = oracle.uniform_targets(5, B, n, q)
    e = psf.samp_p(u, seed=123, first_index=7)
    e_ref = orc.samp_p(123, u, first_index=7)
    assert (e == e_ref).all()
    # the reference's own invariants (mp_perturbation.rs:433-448): f_a(a, samp_p(a, td, u)) == u and check_domain
    assert psf.check_domain(e).all()
    assert (psf.f_a(e) == u).all()
    # sharding independence: rows [40, 60) computed alone equal the same rows of the full batch
    if B >= 60:
        e_part = psf.samp_p(u[40:60], seed=123, first_index=7 + 40)
        assert (e_part == e[40:60]).all()


def test_readme_flow(T, oracle):
    # README.md:62-77 / mp_perturbation.rs:43-56
    gp = T.GadgetParameters.init_default(8, 64)
    psf = T.PSFPerturbation(gp, 3, 25)
    psf.trap_gen(1)
    domain_sample = psf.samp_d(seed=2)
    assert psf.check_domain(domain_sample)
    range_fa = psf.f_a(domain_sample)
    preimage = psf.samp_p(range_fa, seed=3)
    assert psf.check_domain(preimage)
    assert (psf.f_a(preimage) == range_fa).all()
    # alternate covariance, mp_perturbation.rs:89-107
    psf.compute_sqrt_sigma_2(35.0)
    preimage = psf.samp_p(range_fa, seed=4)
    assert psf.check_domain(preimage)
    assert (psf.f_a(preimage) == range_fa).all()


@pytest.mark.parametrize("n,q,r,s", CONFIGS[:3] + [(8, 64, 9.0, 512.0)])   # last: C3's s r = 4608, wide attempt words
def test_samp_d_f_a_check_domain_parity(T, oracle, n, q, r, s):
    psf, orc, (A, R, Lp, Sk, gso) = make_pair(T, oracle, n, q, r, s)
    orc.load_key(A, R, Lp)
    e = psf.samp_d(seed=9, B=7, first_index=3)
    assert (e == orc.samp_d(9, B=7, first_index=3)).all()
    assert psf.check_domain(e).all() and orc.check_domain(e).all()   # mp_perturbation.rs:416-428
    assert (psf.f_a(e) == orc.f_a(e)).all()                           # mp_perturbation.rs:451-464


def test_domain_violations(T, oracle):
    # mp_perturbation.rs:466-554
    gp = T.GadgetParameters.init_default(8, 128)
    psf = T.PSFPerturbation(gp, 3.0, 25.0)
    psf.trap_gen(4)
    m = psf.m
    with pytest.raises(T.PsfError):                       # sigma is a matrix (:470-480)
        psf.f_a(np.zeros((2, m, 2), dtype=np.int64))
    with pytest.raises(T.PsfError):                       # wrong length (:486-496)
        psf.f_a(np.zeros(m - 1, dtype=np.int64))
    too_long = np.zeros(m, dtype=np.int64)
    too_long[0] = 25 * m                                   # (:502-513)
    with pytest.raises(T.PsfError) as ei:
        psf.f_a(too_long)
    assert ei.value.status == 3
    in_domain = np.full(m, 25, dtype=np.int64)             # (:517-532)
    assert psf.check_domain(np.zeros(m, dtype=np.int64))
    assert psf.check_domain(in_domain)
    assert not psf.check_domain(np.zeros(m - 1, dtype=np.int64))   # (:536-554)
    assert not psf.check_domain(np.zeros(m + 1, dtype=np.int64))
    assert not psf.check_domain(too_long)


def test_not_positive_definite(T):
    # mp_perturbation.rs:109-110: s below sqrt(b^2+1)(s_1(R)+1) -> Sigma_2 not PD -> panic
    gp = T.GadgetParameters.init_default(8, 64)
    psf = T.PSFPerturbation(gp, 3, 3.0)
    with pytest.raises(T.PsfError) as ei:
        psf.trap_gen(1)
    assert ei.value.status == 2


def test_no_key(T):
    gp = T.GadgetParameters.init_default(8, 64)
    psf = T.PSFPerturbation(gp, 3, 25)
    with pytest.raises(T.PsfError) as ei:
        psf.samp_p(np.zeros(8, dtype=np.uint64))
    assert ei.value.status == 6
