"""Seeded random inputs for the free functions of the replaced subsystems (sample::g_trapdoor::{gadget_classical, gadget_ring, short_basis_*}), device against the oracle /
against their defining identities: the known-answer files pin the reference's own vectors, this file draws the shapes nobody wrote down -- bases 2..9, moduli of every kind up to
2^61, tags H != I, trapdoor matrices R from a caller's own distribution (entries beyond {-1, 0, 1}), ring degrees 4..64 with wide moduli."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRIMES = [257, 3329, 12289, 65537, 1073741789, 2**31 - 1, 2**61 - 1]


def draw(rng):
    kind = int(rng.integers(0, 3))
    q = int(2 ** rng.integers(3, 61)) if kind == 0 else (int(PRIMES[rng.integers(0, len(PRIMES))]) if kind == 1 else int(rng.integers(9, 2**24)) | 1)
    base = int(rng.choice([2, 2, 3, 4, 5, 7, 9]))
    k = 1
    while base**k < q:
        k += 1
    if k > 64:
        base, k = 2, int(math.ceil(math.log2(q)))
    n = int(rng.integers(1, 9))
    return n, q, base, k


def unit_tag(rng, n, q):
    """an invertible n x n tag over Z_q: unit upper triangular times unit lower triangular (determinant 1)"""
    U = np.triu(rng.integers(0, min(q, 2**30), size=(n, n)).astype(object), 1) + np.eye(n, dtype=object)
    L = np.tril(rng.integers(0, min(q, 2**30), size=(n, n)).astype(object), -1) + np.eye(n, dtype=object)
    return np.array((U @ L) % q, dtype=np.uint64)


@pytest.mark.parametrize("case", range(96))
def test_gadget_functions_on_random_inputs(oracle, case):
    import tools_amd as T
    from tools_amd import gadget as G
    rng = np.random.default_rng(5000 + case)
    n, q, base, k = draw(rng)
    m_bar = n * int(math.ceil(math.log2(q))) + int(rng.integers(0, 12))
    gp = T.GadgetParameters(n, k, m_bar, base, q)
    ogp = oracle.GadgetParams(n, k, m_bar, base, q)
    # gadget vector / matrix, solutions of G x = v (gadget_classical.rs:91-287)
    assert (G.gen_gadget_mat(n, k, base) == oracle.gen_gadget_mat(n, k, base)).all()
    vals = (rng.integers(0, 2**62, size=(n, 5)).astype(object) % q).astype(np.uint64)
    sol = G.find_solution_gadget_mat(vals, q, k, base)
    assert (sol == oracle.find_solution_gadget_mat(vals, q, k, base)).all()
    Gm = oracle.gen_gadget_mat(n, k, base).astype(object)
    assert ((Gm @ sol.astype(object)) % q == vals.astype(object)).all()
    assert (G.short_basis_gadget(gp) == oracle.short_basis_gadget(ogp)).all()
    # gen_trapdoor with a tag and with the caller's own R (entries in [-3, 3]: not PlusMinusOneZero), gadget_classical.rs:56-68
    a_bar = oracle.sample_a_bar(case, n, m_bar, q)
    tag = unit_tag(rng, n, q) if case % 2 else None
    A, R = G.gen_trapdoor(gp, a_bar, tag=tag, seed=77 + case)
    assert (R == oracle.sample_r(77 + case, m_bar, n * k)).all()
    assert (A == oracle.gen_trapdoor(ogp, a_bar, R, tag=tag)).all()
    R2 = rng.integers(-3, 4, size=(m_bar, n * k)).astype(np.int64)
    A2 = G.gen_trapdoor_with_r(gp, a_bar, R2, tag=tag)
    Tm = np.vstack([R2.astype(object), np.eye(n * k, dtype=object)])
    H = np.eye(n, dtype=object) if tag is None else tag.astype(object)
    assert (((A2.astype(object) @ Tm) - H @ Gm) % q == 0).all()          # A [R; I] = H G (gadget_classical.rs:363-385)
    # short basis of Lambda^perp(A) from the trapdoor (short_basis_classical.rs:54-110): bitwise against the oracle, and A S = 0 mod q
    try:
        S = G.gen_short_basis_for_trapdoor(gp, A, R, tag=tag)
    except T.PsfError as err:
        # H^-1 is found by Gauss-Jordan with unit pivots (short_basis_classical.rs:106 calls tag.inverse()): over a modulus with two distinct prime factors an
        # invertible H can present a column without a unit entry (a unit mod 3 here, mod 5 there).  Reported as PSF_ERR_PARAM -- by the oracle too.
        assert err.status == 1 and tag is not None and sum(1 for p_ in (2, 3, 5, 7, 11, 13) if q % p_ == 0) + (1 if q > 1 else 0) >= 2, (n, q, base)
        with pytest.raises(RuntimeError, match="oracle status 1"):
            oracle.gen_short_basis_for_trapdoor(ogp, A, R, tag=tag)
        return
    assert (S == oracle.gen_short_basis_for_trapdoor(ogp, A, R, tag=tag)).all()
    assert ((A.astype(object) @ S.astype(object)) % q == 0).all()


def negacyclic(x, y, n, q):
    out = [0] * n
    for i in range(n):
        for j in range(n):
            t = int(x[i]) * int(y[j])
            if i + j < n:
                out[i + j] += t
            else:
                out[i + j - n] -= t
    return [v % q for v in out]


@pytest.mark.parametrize("case", range(40))
def test_ring_functions_on_random_inputs(oracle, case):
    import tools_amd as T
    from tools_amd import gadget as G
    rng = np.random.default_rng(6000 + case)
    n = int(2 ** rng.integers(2, 7))
    q = int(rng.choice([257, 12289, 2**16 + 1, 1073741789, 2**31 + 11, 2**45 - 55, 2**61 - 1]))
    gp = T.GadgetParametersRing.init_default(n, q)
    k = gp.k
    # negacyclic products in R_q, every method the library has (schoolbook / NTT where q allows), against the definition
    a = (rng.integers(0, 2**62, size=n).astype(object) % q).astype(np.uint64)
    b = rng.integers(-50, 51, size=n).astype(np.int64)
    want = negacyclic(a, b, n, q)
    assert [int(v) for v in G.poly_mul_negacyclic(a, b, q)] == want
    # the trapdoor relation with the caller's own r, e (gadget_ring.rs:62-81, :190-211): a_0 e_j + a_1 r_j + a_{2+j} = base^j
    a_bar = (rng.integers(0, 2**62, size=n).astype(object) % q).astype(np.uint64)
    r = rng.integers(-2, 3, size=(k, n)).astype(np.int64)
    e = rng.integers(-2, 3, size=(k, n)).astype(np.int64)
    av = G.gen_trapdoor_ring_lwe_with(gp, a_bar, r, e)
    for j in range(k):
        lhs = [(x + y + int(z)) % q for x, y, z in zip(negacyclic(av[0], e[j], n, q), negacyclic(av[1], r[j], n, q), av[2 + j])]
        assert lhs == [pow(2, j, q)] + [0] * (n - 1)
    # the embedded short basis against the oracle (short_basis_ring.rs:64-79) when its entries fit the library's int32 form
    ogp = oracle.gadget_params_ring_default(n, q)
    try:
        bt = G.gen_short_basis_for_trapdoor_ring(gp, av, r, e)
    except T.PsfError:
        return                                                               # wide moduli: entries beyond int32 are reported, not truncated
    K, d = k + 2, n * (k + 2)
    ref = oracle.ring_short_basis_t(ogp, av, r, e)                           # d x d, row c = embedding of column c
    assert bt.shape == (K, d, n)
    assert (bt.transpose(1, 0, 2).reshape(d, d) == ref).all()
