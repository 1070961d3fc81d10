"""GPU parity of PSFGPVRing (gpv_ring.rs) against the CPU oracle: ring key material and the embedded short basis
bit-exact (the product builds it in closed form, the oracle by the literal polynomial matrix product), Gram-Schmidt
within tolerance, samp_p bit-exact with a shared key."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def compute_s(n):   # gpv_ring.rs:296-298
    return ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4


CONFIGS = [(5, 2**31 - 58, None), (6, 2**31 - 1, None), (8, 512, 100.0), (4, 16, 60.0), (16, 3329, None)]


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


def polymul_negacyclic(x, y, n, q):
    acc = [0] * n
    for i in range(n):
        for j in range(n):
            if i + j >= n:
                acc[i + j - n] -= int(x[i]) * int(y[j])
            else:
                acc[i + j] += int(x[i]) * int(y[j])
    return [v % q for v in acc]


@pytest.mark.parametrize("n,q,s", CONFIGS)
def test_ring_trap_gen_and_basis_parity(T, oracle, n, q, s):
    s = s or compute_s(n)
    gp = T.GadgetParametersRing.init_default(n, q)
    psf = T.PSFGPVRing(gp, s, 1.005)
    a, (r, e) = psf.trap_gen(9)
    ogp = oracle.gadget_params_ring_default(n, q)
    oa, orr, oe = oracle.ring_trap_gen(ogp, 1.005, 9)
    assert (a == oa).all() and (r == orr).all() and (e == oe).all()
    _, _, _, bt, gt = psf.export_key()
    assert (bt == oracle.ring_short_basis_t(ogp, oa, orr, oe)).all(), "embedded short basis differs"
    g_ref = oracle.gso_rows(bt)
    np.testing.assert_allclose(gt, g_ref, rtol=0, atol=1e-9 * np.abs(g_ref).max())
    # trapdoor relation in R_q (gadget_ring.rs:190-211): a_0 e_j + a_1 r_j + a_{2+j} = base^j
    for j in range(gp.k):
        lhs = [(x + y + int(z)) % q for x, y, z in zip(polymul_negacyclic(a[0], e[j], n, q), polymul_negacyclic(a[1], r[j], n, q), a[2 + j])]
        assert lhs == [pow(2, j, q)] + [0] * (n - 1)


@pytest.mark.parametrize("n,q,s", CONFIGS)
def test_ring_samp_p_parity_and_invariants(T, oracle, n, q, s):
    s = s or compute_s(n)
    gp = T.GadgetParametersRing.init_default(n, q)
    psf = T.PSFGPVRing(gp, s, 1.005)
    a, (r, e) = psf.trap_gen(9)
    _, _, _, bt, gt = psf.export_key()
    orc = oracle.PSFGPVRing(oracle.gadget_params_ring_default(n, q), s, 1.005)
    assert orc.load_key(a, r, e, gso_t=gt) == 0
    B = 6
    u = oracle.uniform_targets(3, B, n, q)
    sg = psf.samp_p(u, seed=21, first_index=2)
    assert (sg == orc.samp_p(21, u, first_index=2)).all()
    assert (sg[:2] == orc.samp_p(21, u[:2], first_index=2, percall=True)).all()
    # f_a(a, samp_p(a, td, u)) == u as an R_q identity (gpv_ring.rs:318-334): sum_j a_j * sigma_j
    for b in range(B):
        acc = [0] * n
        for j in range(gp.k + 2):
            acc = [(x + y) % q for x, y in zip(acc, polymul_negacyclic(a[j], sg[b, j], n, q))]
        assert acc == [int(v) for v in u[b]]
    if psf.check_domain(sg).all():
        assert (psf.f_a(sg) == u).all()
    assert (psf.samp_p(u[4], seed=21, first_index=6) == sg[4]).all()


def test_ring_reference_flow(T):
    # gpv_ring.rs:44-60 doc example; :302-334
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(8, 512), 100, 1.005)
    psf.trap_gen(1)
    ds = psf.samp_d(seed=2)
    assert psf.check_domain(ds)
    rng_fa = psf.f_a(ds)
    pre = psf.samp_p(rng_fa, seed=3)
    assert psf.check_domain(pre)
    assert (psf.f_a(pre) == rng_fa).all()
    for n, q in [(5, 2**31 - 58), (6, 2**31 - 1)]:
        psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), compute_s(n), 1.005)
        psf.trap_gen(4)
        ds = psf.samp_d(seed=5)
        rng_fa = psf.f_a(ds)
        pre = psf.samp_p(rng_fa, seed=6)
        assert (psf.f_a(pre) == rng_fa).all() and psf.check_domain(pre)
    # domain violations (gpv_ring.rs:352-445)
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(8, 512), 100, 1.005)
    psf.trap_gen(1)
    with pytest.raises(T.PsfError):
        psf.f_a(np.zeros((psf.K - 1, psf.n), dtype=np.int64))
    big = np.zeros((psf.K, psf.n), dtype=np.int64)
    big[0, 0] = 100 * psf.d
    with pytest.raises(T.PsfError):
        psf.f_a(big)
    assert not psf.check_domain(big)
    assert psf.check_domain(np.zeros((psf.K, psf.n), dtype=np.int64))


def test_polynomial_ring_product_kernel(T):
    rng = np.random.default_rng(5)
    for n, q in [(4, 16), (5, 2**31 - 58), (256, 3329), (64, 2**40 + 15), (1, 7)]:
        a = rng.integers(0, q, size=(3, n), dtype=np.uint64)
        b = rng.integers(-50, 50, size=(3, n), dtype=np.int64)
        got = T.gadget.poly_mul_negacyclic(a, b, q)
        for c in range(3):
            assert got[c].tolist() == polymul_negacyclic(a[c], b[c], n, q)
    assert T.gadget.poly_mul_negacyclic(np.zeros((0, 8), dtype=np.uint64), np.zeros((0, 8), dtype=np.int64), 17).shape == (0, 8)


def test_ring_f_a_equals_sum_of_ring_products(T):
    """f_a (gpv_ring.rs:243-247) evaluated on the embedding must equal sum_j a_j * sigma_j computed with the R_q kernel."""
    n, q = 16, 3329
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), compute_s(n), 1.005)
    a, _ = psf.trap_gen(2)
    sg = psf.samp_d(seed=3, B=4)
    u = psf.f_a(sg)
    for b in range(4):
        prods = T.gadget.poly_mul_negacyclic(a, sg[b], q)
        assert ((prods.astype(object).sum(axis=0)) % q == u[b].astype(object)).all()


def test_ntt_product_equals_schoolbook(T):
    """(incomplete) negacyclic NTT kernel == schoolbook kernel == big-integer reference, for NTT-friendly primes."""
    rng = np.random.default_rng(9)
    cases = [(3329, 256), (3329, 128), (3329, 512), (7681, 256), (12289, 1024), (257, 64), (17, 8), (5, 2), (2**31 - 2**27 + 1, 256),
             (1073479681, 512)]
    for q, n in cases:
        a = rng.integers(0, q, size=(2, n), dtype=np.uint64)
        b = rng.integers(-q + 1, q, size=(2, n), dtype=np.int64)
        ntt = T.gadget.poly_mul_negacyclic(a, b, q, method=1)
        sb = T.gadget.poly_mul_negacyclic(a, b, q, method=0)
        assert (ntt == sb).all(), (q, n)
        assert ntt[0].tolist() == polymul_negacyclic(a[0], b[0], n, q)
    for q, n in [(16, 8), (3329 * 3, 8), (3329, 6), (7, 4), (2**31 + 11, 8)]:      # not prime / 4 does not divide q-1 / n not 2^a / q too large
        with pytest.raises(T.PsfError) as ei:
            T.gadget.poly_mul_negacyclic(np.zeros((1, n), dtype=np.uint64), np.zeros((1, n), dtype=np.int64), q, method=1)
        assert ei.value.status == 8
