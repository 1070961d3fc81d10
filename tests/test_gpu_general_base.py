"""Gadget parameters set by hand (public fields of GadgetParameters, gadget_parameters.rs:44-52): base 3 and 5, k not the
default -- the shapes the reference's short_basis_gadget tests use (gadget_classical.rs:533-572)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # (n, q, base, k, m_bar, r, s)
    (4, 625, 5, 4, 4 * 4 + 4, 2.0, 60.0),         # q = base^k: bidiagonal S_k with 5 on the diagonal
    (3, 538, 5, 4, 3 * 4 + 4, 2.0, 60.0),         # q = 4123_5: digit column
    (5, 125, 3, 5, 5 * 5 + 9, 2.0, 40.0),         # base 3, 3^5 = 243 >= q
]


@pytest.mark.parametrize("n,q,base,k,m_bar,r,s", CASES)
def test_perturbation_general_base(oracle, n, q, base, k, m_bar, r, s):
    import tools_amd as T
    gp = T.GadgetParameters(n, k, m_bar, base, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, (Sk, gso)) = psf.trap_gen(2)
    ogp = oracle.GadgetParams(n, k, m_bar, base, q)
    orc = oracle.PSFPerturbation(ogp, r, s)
    assert orc.trap_gen(2) == 0
    assert (A == orc.A).all() and (R == orc.R).all() and (Sk == orc.Sk).all()
    np.testing.assert_allclose(Lp, orc.L_packed, rtol=0, atol=1e-9 * np.abs(orc.L_packed).max())
    G = oracle.gen_gadget_mat(n, k, base).astype(object)
    Tm = np.vstack([R.astype(object), np.eye(n * k, dtype=object)])
    assert (((A.astype(object) @ Tm) - G) % q == 0).all()
    orc.load_key(A, R, Lp)
    u = oracle.uniform_targets(4, 7, n, q)
    st = psf.samp_p_stages(u, seed=5)
    for b in range(7):
        tr = orc.samp_p_trace(5, b, u[b])
        for key in ("p", "v", "z", "e"):
            assert (st[key][b] == tr[key]).all(), key
    assert (psf.f_a(st["e"]) == u).all() and psf.check_domain(st["e"]).all()


@pytest.mark.parametrize("n,q,base,k,m_bar,r,s", CASES[:2])
def test_gpv_general_base(oracle, n, q, base, k, m_bar, r, s):
    import tools_amd as T
    gp = T.GadgetParameters(n, k, m_bar, base, q)
    psf = T.PSFGPV(gp, s)
    A, R, (bt, gt) = (lambda t: t)(psf.trap_gen(3, export=False) or psf.export_key(with_R=True))
    assert (bt.T == T.gadget.gen_short_basis_for_trapdoor(gp, A, R)).all()
    assert ((A.astype(object) @ bt.astype(object).T) % q == 0).all()
    orc = oracle.PSFGPV(oracle.GadgetParams(n, k, m_bar, base, q), s)
    assert orc.load_key(A, bt, gt) == 0
    u = oracle.uniform_targets(4, 5, n, q)
    e = psf.samp_p(u, seed=8)
    assert (e == orc.samp_p(8, u)).all()
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()


def test_documented_limits_are_reported_not_crashed():
    """DESIGN.md section 8 'Limits': every one of them comes back as a status code (the mirror raises PsfError)."""
    import tools_amd as T
    from tools_amd import _ffi
    GP = T.GadgetParameters
    # a Gaussian so wide that an in-domain coordinate would not fit the three int8 digits of the Z_q products (s r sqrt(m) >= 2^23)
    with pytest.raises(T.PsfError) as ei:
        T.PSFPerturbation(GP.init_default(8, 64), 1000.0, 1000.0)
    assert ei.value.status == _ffi.ERR_UNSUPPORTED
    # more than 64 gadget digits
    with pytest.raises(T.PsfError) as ei:
        T.PSFPerturbation(GP(4, 65, 4 * 65 + 4, 2, 2**61), 3.0, 50.0)
    assert ei.value.status == _ffi.ERR_UNSUPPORTED
    # modulus at or above 2^62, degenerate parameters
    for bad in (GP(4, 62, 252, 2, 2**62), GP(0, 6, 48, 2, 64), GP(4, 6, 28, 1, 64), GP(4, 6, 28, 2, 1)):
        with pytest.raises(T.PsfError) as ei:
            T.PSFPerturbation(bad, 3.0, 50.0)
        assert ei.value.status == _ffi.ERR_PARAM
    for r, s in ((0.0, 50.0), (3.0, -1.0), (float("nan"), 50.0)):
        with pytest.raises(T.PsfError) as ei:
            T.PSFPerturbation(GP.init_default(4, 64), r, s)
        assert ei.value.status == _ffi.ERR_PARAM
    # ring modulus at or above 2^62 (round 3 lifted the 2^31 limit: tests/test_gpu_boundary_completion.py runs 2^31 + 11 ... 2^61 - 1)
    with pytest.raises(T.PsfError) as ei:
        T.PSFGPVRing(T.GadgetParametersRing(8, 62, 64, 2, 2**62), 100.0, 1.005)
    assert ei.value.status == _ffi.ERR_UNSUPPORTED
    # a gadget too short for q (gadget_classical.rs:170-172 panics there)
    psf = T.PSFPerturbation(GP(4, 5, 5 * 4 + 4, 2, 64), 3.0, 50.0)
    with pytest.raises(T.PsfError):
        psf.trap_gen(1)
