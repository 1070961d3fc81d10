"""The free functions of the replaced subsystems (sample::g_trapdoor::{gadget_classical, gadget_ring, short_basis_ring}) through the C ABI:
gen_trapdoor with a caller's A_bar and a non-identity tag (gadget_classical.rs:56-68), gen_trapdoor_ring_lwe (gadget_ring.rs:62-81),
gen_gadget_ring (:103-109), find_solution_gadget_ring (:145-166), gen_short_basis_for_trapdoor_ring (short_basis_ring.rs:64-79), and the
single-process multi-handle entry psfp_samp_p_multi.  Known answers from the reference's tests where it has them, the oracle and the
reference's invariants elsewhere."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


@pytest.mark.parametrize("n,q", [(6, 128), (10, 157), (4, 2**45), (3, 2**61 - 1)])
def test_gen_trapdoor_with_tag(T, oracle, n, q):
    gp = T.GadgetParameters.init_default(n, q)
    rng = np.random.default_rng(n)
    a_bar = rng.integers(0, q, size=(n, gp.m_bar), dtype=np.uint64)
    tag = rng.integers(0, q, size=(n, n), dtype=np.uint64)
    A, R = T.gadget.gen_trapdoor(gp, a_bar, tag, seed=9)
    assert (R == oracle.sample_r(9, gp.m_bar, n * gp.k)).all()                  # the PlusMinusOneZero stream of trap_gen
    ogp = oracle.gadget_params_default(n, q)
    assert (A == oracle.gen_trapdoor(ogp, a_bar, R, tag=tag)).all()
    # the trapdoor relation A [R; I] = H G (gadget_classical.rs:363-414)
    Tm = np.vstack([R.astype(object), np.eye(n * gp.k, dtype=object)])
    G = oracle.gen_gadget_mat(n, gp.k, 2).astype(object)
    assert (((A.astype(object) @ Tm) - tag.astype(object) @ G) % q == 0).all()
    # tag = NULL is the identity, and equals what trap_gen builds from the same A_bar
    A_id, R_id = T.gadget.gen_trapdoor(gp, a_bar, None, seed=9)
    assert (R_id == R).all() and (A_id == oracle.gen_trapdoor(ogp, a_bar, R)).all()
    assert (A_id[:, :gp.m_bar] == a_bar).all()


def test_gen_trapdoor_modulus_too_large(T):
    gp = T.GadgetParameters(4, 5, 24, 2, 100)          # 2^5 < 100: gadget_classical.rs:170-172
    with pytest.raises(T.PsfError) as ei:
        T.gadget.gen_trapdoor(gp, np.zeros((4, 24), dtype=np.uint64))
    assert ei.value.status == 4


def test_ring_gadget_kats_and_solutions(T, kats, oracle):
    G = T.gadget
    assert G.gen_gadget_ring(4, 2).tolist() == [1, 2, 4, 8]                     # gadget_ring.rs:103-109
    kv = kats["gen_gadget_vec/correctness_base_5"]
    assert G.gen_gadget_ring(kv["k"], kv["base"]).tolist() == kv["expect"]
    # the doctest at gadget_ring.rs:128-143: u = 5 124 12 14 14 1 2 4 1 5 in Z_128[X]/(X^10 + 1)
    u = np.array([5, 124, 12, 14, 14, 1, 2, 4, 1, 5], dtype=np.uint64)
    gp = T.GadgetParametersRing.init_default(10, 128)
    sol = G.find_solution_gadget_ring(u, 128, gp.k, 2)
    assert sol.shape == (gp.k, 10) and sol.min() >= 0 and sol.max() <= 1
    assert ((G.gen_gadget_ring(gp.k, 2) @ sol) % 128 == u.astype(np.int64)).all()   # <g^t, x> = u
    assert (sol == oracle.find_solution_gadget_ring(u, 128, gp.k, 2)).all()
    with pytest.raises(T.PsfError) as ei:                                       # "panics if the modulus of the value is greater than base^k"
        G.find_solution_gadget_ring(u, 1000, 5, 3)
    assert ei.value.status == 4


def _polys(rows, n):
    return np.array([[(p + [0] * n)[:n] for p in row] for row in rows], dtype=np.int64)


def test_short_basis_ring_fixture_and_invariants(T, kats, oracle):
    """The reference's fixed (a, r, e), n = 4, q = 16 (short_basis_ring.rs:358-444): the ABI's polynomial matrix equals the literal
    product sa_l * sa_r mod X^n + 1 of the KAT-pinned factors; and for a real trapdoor the basis is in the kernel of a over R_q
    (short_basis_ring.rs:183-199)."""
    kl = kats["short_basis_ring/working_sa_l"]
    n, q = kl["n"], kl["q"]
    gp = T.GadgetParametersRing.init_default(n, q)
    a = _polys([kl["a"]], n)[0].astype(np.uint64)
    r = _polys([kl["r"]], n)[0]
    e = _polys([kl["e"]], n)[0]
    basis = T.gadget.gen_short_basis_for_trapdoor_ring(gp, a, r, e)            # (k+2) x n(k+2) x n
    K, d = gp.k + 2, n * (gp.k + 2)
    assert basis.shape == (K, d, n)
    ogp = oracle.gadget_params_ring_default(n, q)
    bt = oracle.ring_short_basis_t(ogp, a, r, e)                                # d x d, row c = embedding of column c (literal product)
    for col in range(d):
        for row in range(K):
            assert (basis[row, col] == bt[col, row * n:(row + 1) * n]).all()
    # a real trapdoor from the ABI: a * basis == 0 in R_q for every column
    n2, q2 = 8, 257
    gp2 = T.GadgetParametersRing.init_default(n2, q2)
    a_bar = np.random.default_rng(3).integers(0, q2, size=n2, dtype=np.uint64)
    a2, r2, e2 = T.gadget.gen_trapdoor_ring_lwe(gp2, a_bar, 2.5, seed=4)
    oa, orr, oe = oracle.ring_trap_gen(oracle.gadget_params_ring_default(n2, q2), 2.5, 4)
    assert (r2 == orr).all() and (e2 == oe).all()                               # same SampleZ streams as the oracle's trap_gen
    assert a2[0].tolist() == [1] + [0] * (n2 - 1) and (a2[1] == a_bar).all()
    basis2 = T.gadget.gen_short_basis_for_trapdoor_ring(gp2, a2, r2, e2)
    K2 = gp2.k + 2
    for col in range(basis2.shape[1]):
        acc = np.zeros(n2, dtype=object)
        for j in range(K2):
            x, y = a2[j].astype(object), basis2[j, col].astype(object)
            for i in range(n2):
                for t in range(n2):
                    if i + t >= n2:
                        acc[i + t - n2] -= x[i] * y[t]
                    else:
                        acc[i + t] += x[i] * y[t]
        assert (acc % q2 == 0).all()
    # the trapdoor relation of gen_trapdoor_ring_lwe: a_{2+j} = g_j - (a_bar r_j + e_j)  (gadget_ring.rs:190-211)
    for j in range(gp2.k):
        prod = T.gadget.poly_mul_negacyclic(a_bar, r2[j], q2).astype(np.int64)
        want = (-(prod + e2[j])) % q2
        want[0] = (want[0] + pow(2, j, q2)) % q2
        assert (a2[2 + j].astype(np.int64) == want).all()


def test_samp_p_multi_equals_single_handle(T, oracle):
    n, q, r, s, B = 8, 64, 3.0, 25.0, 301
    gp = T.GadgetParameters.init_default(n, q)
    handles = [T.PSFPerturbation(gp, r, s) for _ in range(3)]
    for h in handles:
        h.trap_gen(5, export=False)                       # same seed: same key on every handle (every GPU of a node)
    u = oracle.uniform_targets(2, B, n, q)
    single = handles[0].samp_p(u, seed=77, first_index=1000)
    from tools_amd.psf import samp_p_multi
    for cnt in (1, 2, 3):
        multi = samp_p_multi(handles[:cnt], u, seed=77, first_index=1000)
        assert (multi == single).all()
    assert (handles[0].f_a(single) == u).all()
    # fewer rows than handles: the empty shares are skipped
    few = samp_p_multi(handles, u[:2], seed=77, first_index=1000)
    assert (few == single[:2]).all()


def test_samp_p_multi_drives_its_handles_concurrently(T, oracle):
    """One worker thread per handle: handle 1 must have ENQUEUED its launch sequence before handle 0's last row has landed in the caller's
    (pageable) buffer -- with the single-threaded form of round 2 every handle's window began after the previous one's had ended.  All handles sit on
    one GPU here, so the device work itself still serialises; what the windows show is that nothing on the host waits for the neighbour.  On a
    node with several GPUs the same windows overlap for their whole length."""
    from tools_amd.psf import samp_p_multi, multi_timing
    n, q, r, s, B = 64, 128, 6.0, 100.0, 6144                    # benches/psf.rs:78-93 parameters; 2048 rows per handle, ~50 MB of output each
    gp = T.GadgetParameters.init_default(n, q)
    handles = [T.PSFPerturbation(gp, r, s) for _ in range(3)]
    for h in handles:
        h.trap_gen(5, export=False)
    u = oracle.uniform_targets(2, B, n, q)
    samp_p_multi(handles, u, seed=1)                              # warm-up: buffers, first-touch of the output pages
    multi = samp_p_multi(handles, u, seed=77, first_index=10)
    win = [multi_timing(h) for h in handles]
    assert all(0 <= a <= b for a, b in win), win
    earliest_done = min(b for _, b in win)
    # The single-threaded form launched exactly one handle before the first completion.  With a worker per handle all three normally are (their windows
    # overlap from the start); on one GPU the runtime may still hold one worker's first call behind a neighbour's blocking copy, so two of three is the bar.
    assert sum(1 for a, _ in win if a < earliest_done) >= 2, win
    single = handles[0].samp_p(u, seed=77, first_index=10)
    assert (multi == single).all()
    with pytest.raises(T.PsfError):                               # a handle may appear once
        samp_p_multi([handles[0], handles[0]], u[:4], seed=1)
