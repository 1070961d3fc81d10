"""GPU parity of PSFGPV (gpv.rs) against the CPU oracle: A, R and the short basis bit-exact; the Gram-Schmidt vectors
within tolerance (different summation order on the device); samp_p bit-exact once both sides hold the same key."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = [(5, 256, 10.0), (6, 128, 10.0), (8, 128, 90.0), (4, 23, 12.0), (10, 157, 30.0)]   # gpv.rs:239,255 ; benches/psf.rs:27-33


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


@pytest.mark.parametrize("n,q,s", CONFIGS)
def test_trap_gen_parity(T, oracle, n, q, s):
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    A, R, (bt, gt) = (lambda t: (t[0], t[1], t[2]))(psf.trap_gen(5, export=False) or psf.export_key(with_R=True))
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.trap_gen(5) == 0
    assert (A == orc.A).all() and (R == orc.R).all()
    assert (bt == orc.basis_t).all(), "short basis S_A differs"
    # the basis also equals the host-side restatement of gen_short_basis_for_trapdoor (columns = basis vectors)
    S_host = T.gadget.gen_short_basis_for_trapdoor(psf.gp, A, R)
    assert (bt == S_host.T).all()
    assert ((A.astype(object) @ bt.astype(object).T) % q == 0).all()          # short_basis_classical.rs:128-188
    scale = np.abs(orc.gso_t).max()
    np.testing.assert_allclose(gt, orc.gso_t, rtol=0, atol=1e-9 * scale)


@pytest.mark.parametrize("n,q,s", CONFIGS)
def test_samp_p_parity_and_invariants(T, oracle, n, q, s):
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    A, (bt, gt) = psf.trap_gen(5)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.load_key(A, bt, gt) == 0
    B = 9
    u = oracle.uniform_targets(2, B, n, q)
    e = psf.samp_p(u, seed=31, first_index=4)
    assert (e == orc.samp_p(31, u, first_index=4)).all()
    assert (e[:3] == orc.samp_p(31, u[:3], first_index=4, percall=True)).all()     # per-call elimination, gpv.rs:153-156
    assert (psf.f_a(e) == u).all() if psf.check_domain(e).all() else True
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()
    # sharding independence
    assert (psf.samp_p(u[5:8], seed=31, first_index=9) == e[5:8]).all()
    # single call
    assert (psf.samp_p(u[0], seed=31, first_index=4) == e[0]).all()


def test_reference_flow(T):
    # gpv.rs:40-51 doc example and :253-268
    for n, q in [(8, 64), (5, 256), (6, 128)]:
        psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), 12 if q == 64 else 10)
        psf.trap_gen(1, export=False)
        domain_sample = psf.samp_d(seed=2)
        assert psf.check_domain(domain_sample)
        range_fa = psf.f_a(domain_sample)
        preimage = psf.samp_p(range_fa, seed=3)
        assert psf.check_domain(preimage)
        assert (psf.f_a(preimage) == range_fa).all()


def test_samp_d_f_a_domain(T, oracle):
    n, q, s = 6, 128, 10.0
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    A, (bt, gt) = psf.trap_gen(7)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    orc.load_key(A, bt, gt)
    e = psf.samp_d(seed=3, B=6, first_index=2)
    assert (e == orc.samp_d(3, B=6, first_index=2)).all()
    assert psf.check_domain(e).all()
    assert (psf.f_a(e) == orc.f_a(e)).all()
    m = psf.m
    with pytest.raises(T.PsfError):                      # gpv.rs:290-340 should_panic cases
        psf.f_a(np.zeros(m - 1, dtype=np.int64))
    big = np.zeros(m, dtype=np.int64)
    big[0] = 10 * m
    with pytest.raises(T.PsfError):
        psf.f_a(big)
    assert not psf.check_domain(big) and not psf.check_domain(np.zeros(m + 1, dtype=np.int64))
    assert psf.check_domain(np.full(m, 10, dtype=np.int64))


@pytest.mark.parametrize("g", [1, 2, 4])
def test_sampler_gives_the_same_bits_for_every_lane_split(T, oracle, monkeypatch, exp_lib, g):
    """k_np_sample<G> evaluates the attempts of one draw on 64 / G lanes; the value of a draw is its first accepted attempt,
    so G = 1, 2, 4 must agree with each other and with the oracle (PSF_NP_G forces the variant)."""
    n, q, s = 10, 157, 30.0
    monkeypatch.setenv("PSF_NP_G", str(g))
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFGPV(gp, s)
    A, (bt, gt) = psf.trap_gen(5)
    u = oracle.uniform_targets(2, 11, n, q)          # 11: the last wave / subgroup is partly empty for every G
    e = psf.samp_p(u, seed=31, first_index=4)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.load_key(A, bt, gt) == 0
    assert (e == orc.samp_p(31, u, first_index=4)).all()
    assert psf.nearest_plane_stats() == ((psf.m + 63) // 64, 0)


@pytest.mark.parametrize("n,q,s", [(3, 2**45, 60.0), (2, 2**60, 50.0), (2, 2**61 - 1, 50.0), (3, 2**31 - 1, 40.0)])
def test_large_modulus(T, oracle, n, q, s):
    """q sqrt(n) > 2^13 s: |c0| up to q would leave the centres of a single walk with an error of 2^-53 q sqrt(n) / s draw widths (2^-9 at
    2^45, garbage at 2^60: tests/test_oracle_centre_precision.py), so both sides sample in two passes -- a short coset representative first, the
    preimage around it second.  Bit-exact against the oracle, A e = u, check_domain, and short (a merely valid e could be as long as q)."""
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFGPV(gp, s)
    assert psf.two_pass
    A, (bt, gt) = psf.trap_gen(9)
    u = oracle.uniform_targets(3, 5, n, q)
    e = psf.samp_p(u, seed=11, first_index=0)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.two_pass
    assert orc.load_key(A, bt, gt) == 0
    assert (e == orc.samp_p(11, u, first_index=0)).all()
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()
    assert psf.check_domain(e).all()
    assert ((e.astype(np.float64) ** 2).sum(axis=1) < 2.0 * s * s / (2 * np.pi) * psf.m).all()
    assert (psf.f_a(e) == u).all()
    assert (psf.samp_p(u[2:4], seed=11, first_index=2) == e[2:4]).all()               # sharding independence holds for both passes


@pytest.mark.parametrize("force", [0, 1])
def test_both_forms_of_the_walk_stay_reachable(T, oracle, monkeypatch, exp_lib, force):
    """PSF_NP_TWO_PASS forces one form: two passes at a small modulus (the second projection runs over all d coordinates, K = d, several
    blocks) and one pass at a large one (the pre-round-3 behaviour, kept only as the comparison arm); bitwise against the oracle forced alike."""
    n, q, s = (40, 256, 300.0) if force else (3, 2**45, 60.0)
    monkeypatch.setenv("PSF_NP_TWO_PASS", str(force))
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    assert psf.two_pass == bool(force)
    A, (bt, gt) = psf.trap_gen(4)
    u = oracle.uniform_targets(6, 7, n, q)
    e = psf.samp_p(u, seed=5, first_index=1)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    orc.set_two_pass(force)
    assert orc.load_key(A, bt, gt) == 0
    assert (e == orc.samp_p(5, u, first_index=1)).all()
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()


def test_recombination_in_64_bit_integers_when_the_digit_planes_do_not_fit(T, oracle):
    """e = sum z_i b_i runs on the int8 matrix cores when basis entries and z fit two balanced base-256 digits; otherwise the
    64-bit integer kernel takes over: (a) a basis with an entry beyond 32639 (a unimodular transform of the device's basis, so
    still a basis of the same lattice), decided at load_key; (b) a Gaussian so wide that some |z_i| > 32639, decided on the
    device per call.  Both must give the oracle's bits and A e = u."""
    n, q, s = 6, 128, 40.0
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFGPV(gp, s)
    A, (bt, gt) = psf.trap_gen(5)
    u = oracle.uniform_targets(2, 7, n, q)
    psf.samp_p(u, seed=3)
    assert psf.nearest_plane_stats()[1] == 0
    bt2 = bt.copy()
    bt2[-1] = bt[-1] + 40000 * bt[-2]                # the last vector: its Gram-Schmidt component (and every norm) is unchanged
    assert np.abs(bt2).max() > 32639
    gt2 = oracle.gso_rows(bt2)
    big = T.PSFGPV(gp, s)
    big.load_key(A, bt2, gt2)
    e = big.samp_p(u, seed=3)
    assert big.nearest_plane_stats()[1] == 1
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.load_key(A, bt2, gt2) == 0
    assert (e == orc.samp_p(3, u)).all()
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()
    wide = T.PSFGPV(gp, 400000.0)
    wide.load_key(A, bt, gt)
    e = wide.samp_p(u, seed=4)
    assert wide.nearest_plane_stats()[1] == 1
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), 400000.0)
    assert orc.load_key(A, bt, gt) == 0
    assert (e == orc.samp_p(4, u)).all()
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()


@pytest.mark.parametrize("n,q,immediate", [(32, 256, 0), (64, 256, 1), (128, 2**15, 0), (128, 2**15, 1), (300, 2**15, 0)])
def test_samp_p_parity_across_block_counts(T, oracle, monkeypatch, exp_lib, n, q, immediate):
    """Lattice dimensions of 537 ... 9081 rows: 9 to 142 blocks of 64 with a short top block, the last one beyond the 8192 rows
    the register-resident walk of round 1 was limited to.  Key from the device, three preimages (the last wave is partly empty);
    s = 1000 makes |z| > 127 common, so the hi digit plane of z is exercised.  Both schedules of the updates below a block (the
    library picks by batch size; PSF_NP_IMMEDIATE forces one): every block into every row below it in the next launch, or only into
    the rows of its own and the next panel of 8 blocks with the rest deferred until its panel is complete -- same fma chains, same bits."""
    s = 1000.0
    monkeypatch.setenv("PSF_NP_IMMEDIATE", str(immediate))
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFGPV(gp, s)
    A, (bt, gt) = psf.trap_gen(3)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    assert orc.load_key(A, bt, gt) == 0
    u = oracle.uniform_targets(5, 3, n, q)
    e = psf.samp_p(u, seed=17, first_index=2)
    assert psf.nearest_plane_stats() == ((psf.m + 63) // 64, 0)
    assert (e == orc.samp_p(17, u, first_index=2)).all()
    assert np.abs(e).max() > 127
    assert ((A.astype(object) @ e.astype(object).T).T % q == u.astype(object)).all()
