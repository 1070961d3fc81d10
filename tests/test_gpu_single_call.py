"""The single-call / small-batch path of PSFPerturbation::samp_p (psf.rs:48-80: one call = one preimage; benches/psf.rs:38,63-65,90-92 time
exactly that): the streaming product k_trmm_stream must give the oracle's bits at every batch size around its fragment boundaries, for
row-tile counts that are odd / not a multiple of the workgroup's eight tasks, and the same bits as the batch kernel for every tile shape."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (n, q, r, s): m = 496 (31 tiles of 16 rows, 4 row blocks), m = 932 (59 tiles, benches/psf.rs:79), m = 1316 (83 tiles, 11 row blocks)
SHAPES = [(12, 2**20, np.log2(12), 60.0), (64, 128, np.log2(64), 100.0), (40, 2**16, 4.0, 120.0)]


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


@pytest.fixture(scope="module", params=SHAPES, ids=lambda p: f"n{p[0]}")
def pair(request, T, oracle):
    n, q, r, s = request.param
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    A, (R, Lp, _) = psf.trap_gen(21)
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    orc.load_key(A, R, Lp)
    yield psf, orc, n, q
    psf.close()


@pytest.fixture(scope="module", params=SHAPES, ids=lambda p: f"n{p[0]}")
def exp_pair(request, T, oracle):
    """the same pair with the handle created in the EXPERIMENTS build of the library (tests that flip PSF_* switches; a handle stays with the build that made it)"""
    from tests.conftest import EXP_LIB
    from tools_amd import _ffi
    if not os.path.exists(EXP_LIB):
        pytest.skip("the experiments build of the library is missing (make -C tools_amd/csrc exp)")
    n, q, r, s = request.param
    _ffi.lib()
    old, _ffi._lib = _ffi._lib, _ffi.open_library(EXP_LIB)
    try:
        psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
        A, (R, Lp, _) = psf.trap_gen(21)
    finally:
        _ffi._lib = old
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    orc.load_key(A, R, Lp)
    yield psf, orc, n, q
    psf.close()


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65])
def test_small_batch_stages_bitwise(pair, oracle, B):
    psf, orc, n, q = pair
    u = oracle.uniform_targets(90 + B, B, n, q)
    st = psf.samp_p_stages(u, seed=500 + B, first_index=10 * B)
    rows = sorted(set([0, B - 1, B // 2, min(B - 1, 16), min(B - 1, 47)]))
    for b in rows:
        tr = orc.samp_p_trace(500 + B, 10 * B + b, u[b])
        assert (st["d"][b].view(np.uint64) == tr["d"].view(np.uint64)).all(), "normals differ"
        assert (st["x"][b].view(np.uint64) == tr["x"].view(np.uint64)).all(), "centres x = sqrt(Sigma_2) d differ"
        assert (st["p"][b] == tr["p"]).all() and (st["v"][b] == tr["v"]).all() and (st["z"][b] == tr["z"]).all()
    e_ref = orc.samp_p(500 + B, u, first_index=10 * B)
    assert (st["e"] == e_ref).all()
    assert psf.check_domain(st["e"]).all() and (psf.f_a(st["e"]) == u).all()      # mp_perturbation.rs:433-448


@pytest.mark.parametrize("shape,B", [("1,1", 16), ("1,1", 100), ("1,2", 100), ("1,4", 100), ("1,8", 100), ("2,1", 5), ("2,1", 100), ("2,2", 20), ("2,2", 100),
                                     ("2,4", 100), ("4,2", 100), ("1,4", 300), ("2,4", 257), ("4,2", 513)])
def test_every_tile_shape_gives_the_batch_kernels_bits(exp_pair, exp_lib, oracle, shape, B):
    """PSF_TRMM_STREAM_SHAPE forces (row tiles, column fragments) per wave; PSF_TRMM_STREAM_MAX = 0 is the batch kernel (k_trmm_f64_big)."""
    psf, orc, n, q = exp_pair
    u = oracle.uniform_targets(4, B, n, q)
    old = {k: os.environ.get(k) for k in ("PSF_TRMM_STREAM_MAX", "PSF_TRMM_STREAM_SHAPE")}
    try:
        os.environ["PSF_TRMM_STREAM_MAX"] = "0"
        os.environ.pop("PSF_TRMM_STREAM_SHAPE", None)
        ref = psf.samp_p_stages(u, seed=9, first_index=3)
        os.environ["PSF_TRMM_STREAM_MAX"] = "4096"
        os.environ["PSF_TRMM_STREAM_SHAPE"] = shape
        got = psf.samp_p_stages(u, seed=9, first_index=3)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert (got["x"].view(np.uint64) == ref["x"].view(np.uint64)).all(), "streaming product differs from k_trmm_f64_big"
    assert (got["e"] == ref["e"]).all()


def test_one_call_is_one_preimage_of_the_batch(pair, oracle):
    """Row b of a batch equals the single call with first_index + b (the reference's call, repeated): the streaming and the batch kernel meet here."""
    psf, orc, n, q = pair
    B = 200
    u = oracle.uniform_targets(8, B, n, q)
    e = psf.samp_p(u, seed=77, first_index=1000)
    for b in (0, 63, 64, 199):
        e1 = psf.samp_p(u[b:b + 1], seed=77, first_index=1000 + b)
        assert (e1[0] == e[b]).all()


def test_fused_call_with_a_general_base(T, oracle):
    """base 3 / 5 gadgets (gadget_classical.rs:169-229 digits in base b) through the fused kernel"""
    for n, k, base, q in [(4, 5, 3, 243), (3, 4, 5, 600)]:
        m_bar = n * int(np.ceil(np.log2(q))) + 7
        gp = T.GadgetParameters(n, k, m_bar, base, q)
        r = 3.0
        s = r * np.sqrt(base * base + 1) * (np.sqrt(m_bar) + np.sqrt(n * k) + 4.0) * 1.5
        psf = T.PSFPerturbation(gp, r, s)
        assert psf.m <= 256
        A, (R, Lp, _) = psf.trap_gen(8)
        orc = oracle.PSFPerturbation(oracle.GadgetParams(n, k, m_bar, base, q), r, s)
        orc.load_key(A, R, Lp)
        u = oracle.uniform_targets(2, 9, n, q)
        e = psf.samp_p(u, seed=19)
        assert (e == orc.samp_p(19, u)).all() and (psf.f_a(e) == u).all()
        psf.close()


FUSED = [  # (n, q, r, s): m <= 256, every kind of modulus / sampler word the stage kernels know
    (8, 128, 3.0, 30.0),             # benches/psf.rs:51-66 (m = 121)
    (8, 64, 3.0, 25.0),              # README.md:62-66
    (15, 157, np.log2(15), 40.0),    # m = 256 exactly; prime modulus: digit column in S_k
    (2, 2**60, 2.0, 70.0),           # k = 60, 128-bit products in v = u - A p
    (2, 2**61 - 1, 2.0, 70.0),
    (8, 64, 100.0, 25.0),            # |z| > 127
    (8, 64, 400.0, 25.0),            # 32-bit attempt words
]


@pytest.mark.parametrize("n,q,r,s", FUSED)
@pytest.mark.parametrize("B", [1, 5, 64])
def test_fused_one_launch_call_gives_the_oracles_rows(T, oracle, monkeypatch, exp_lib, n, q, r, s, B):
    """k_samp_p_small: the whole samp_p of a preimage in one workgroup (small m, few preimages -- the reference's own benchmarks, benches/psf.rs:51-66).
    Same rows as the oracle and as the stage kernels (PSF_FUSED_MAX=0)."""
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    assert psf.m <= 256
    A, (R, Lp, _) = psf.trap_gen(31)
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    orc.load_key(A, R, Lp)
    u = oracle.uniform_targets(12, B, n, q)
    psf.enable_timing(True)
    e = psf.samp_p(u, seed=600 + B, first_index=77)
    assert "k_samp_p_small" in dict(psf.get_timing()), "the fused kernel did not run"
    psf.enable_timing(False)
    assert (e == orc.samp_p(600 + B, u, first_index=77)).all()
    assert psf.check_domain(e).all() and (psf.f_a(e) == u).all()
    monkeypatch.setenv("PSF_FUSED_MAX", "0")
    assert (psf.samp_p(u, seed=600 + B, first_index=77) == e).all()
    psf.close()


@pytest.mark.parametrize("B", [1, 2, 3, 4])
def test_streaming_stage_kernels_of_a_handful_of_preimages_equal_the_matrix_core_ones(exp_pair, exp_lib, oracle, B):
    """Up to four preimages e = p + [R; I] z streams R once (k_recombine_small), one preimage also streams A for v = u - A p (k_syndrome_small), up to
    n B = 4096 the gadget walk runs one wave per problem: PSF_RECOMBINE_SMALL=0 / PSF_SYNDROME_SMALL=0 / PSF_GADGET_WAVE=0 are the matrix-core and queue forms.
    PSF_SYNDROME_SMALL=4 takes the streaming syndrome where it is not the default."""
    psf, orc, n, q = exp_pair
    u = oracle.uniform_targets(61 + B, B, n, q)
    names = ("PSF_RECOMBINE_SMALL", "PSF_SYNDROME_SMALL", "PSF_GADGET_WAVE")
    old = {k: os.environ.get(k) for k in names}
    try:
        for k in names:
            os.environ[k] = "0"
        ref = psf.samp_p_stages(u, seed=33, first_index=7)
        for k in names:
            os.environ.pop(k, None)
        got = psf.samp_p_stages(u, seed=33, first_index=7)
        os.environ["PSF_SYNDROME_SMALL"] = "4"
        got4 = psf.samp_p_stages(u, seed=33, first_index=7)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for g in (got, got4):
        assert (g["v"] == ref["v"]).all() and (g["z"] == ref["z"]).all() and (g["e"] == ref["e"]).all()
    assert (ref["e"] == orc.samp_p(33, u, first_index=7)).all()


@pytest.mark.parametrize("q", [2**61 - 1, 2**60, 1073741789])
def test_streaming_syndrome_with_wide_moduli(T, oracle, q):
    """k_syndrome_small joins two signed 64-bit sums in 128 bits and reduces once: words of A up to 2^62, negative p."""
    n, r, s = 6, 4.0, 90.0
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = psf.trap_gen(3)
    assert psf.m > 256                                      # not the one-launch kernel
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    orc.load_key(A, R, Lp)
    for B in (1, 2):
        u = oracle.uniform_targets(5 + B, B, n, q)
        e = psf.samp_p(u, seed=12, first_index=4)
        assert (e == orc.samp_p(12, u, first_index=4)).all()
        assert (psf.f_a(e) == u).all()
    psf.close()


@pytest.mark.parametrize("ternary", [True, False])
def test_compact_key_copies_of_small_calls_give_the_same_rows(T, oracle, ternary):
    """A call with a handful of preimages reads a two-bit copy of R (k_recombine_small2; only a {-1, 0, 1} trapdoor has one) and a 32-bit copy of A
    (k_syndrome_small32, q <= 2^32).  The copies are built behind the first small call after a key change -- that call still reads the full-size matrices -- and
    taken over by the calls after it; a trapdoor with other entries (psfp_load_key) keeps the int8 kernel.  Every call must return the oracle's rows."""
    import torch
    n, q, r, s = 40, 2**16, 4.0, 120.0
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s if ternary else 3 * s)
    A, (R, Lp, _) = psf.trap_gen(33)
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s if ternary else 3 * s)
    if not ternary:                                             # a caller's own trapdoor with entries beyond {-1, 0, 1}: A = [A_bar | G - A_bar R] for this R
        rng = np.random.default_rng(3)
        R2 = rng.integers(-2, 3, size=R.shape).astype(np.int8)
        A2 = T.gadget.gen_trapdoor_with_r(gp, A[:, :psf.m_bar], R2)
        psf.load_trapdoor(R2, A2)
        psf.compute_sqrt_sigma_2(3 * s)
        A, (R, Lp, _) = psf.export_key()
    orc.load_key(A, R, Lp)
    for B in (1, 3, 4, 2):
        u = oracle.uniform_targets(60 + B, B, n, q)
        want = orc.samp_p(700 + B, u, first_index=5)
        for rep in range(3):                                    # first call after the key change: full-size matrices; later calls: the compact copies
            got = psf.samp_p(u, seed=700 + B, first_index=5)
            torch.cuda.synchronize()
            assert (got == want).all(), (B, rep, ternary)
    psf.close()


@pytest.mark.parametrize("B", [1, 2])
def test_fused_tail_of_one_or_two_preimages_matches_the_oracle_in_every_stage(pair, oracle, B):
    """k_round_syndrome_small (round 6): with one or two preimages ONE launch behind the product rounds x (four lanes per sample) and adds every tile's share
    A[:, rows] p[rows] of the syndrome from the transposed compact copy of A; the separate rounding and syndrome launches disappear.  The compact copies are
    built beside the first small calls after a key change, so the first call runs the separate kernels and a later one the fused launch: both must give the
    oracle's d, x, p, v, z, e, and the timing slots say which form ran."""
    import time
    psf, orc, n, q = pair
    u = oracle.uniform_targets(700 + B, B, n, q)
    trace = [orc.samp_p_trace(44, 9 + b, u[b]) for b in range(B)]
    fused_seen = False
    for attempt in range(6):
        psf.enable_timing(True)
        st = psf.samp_p_stages(u, seed=44, first_index=9)
        names = set(dict(psf.get_timing()))
        psf.enable_timing(False)
        fused = "k_perturb_round" not in names and "k_trmm_f64" in names
        fused_seen = fused_seen or fused
        assert (st["e"] == orc.samp_p(44, u, first_index=9)).all(), (attempt, fused)
        for b in range(B):
            assert (st["x"][b].view(np.uint64) == trace[b]["x"].view(np.uint64)).all(), (attempt, fused)
            for key in ("p", "v", "z"):
                assert (st[key][b] == trace[b][key]).all(), (key, attempt, fused)
        assert psf.check_domain(st["e"]).all() and (psf.f_a(st["e"]) == u).all()
        if fused:
            break
        time.sleep(0.05)                              # the packers of the compact copies finish in the background
    if q <= 2**32 and n % 8 == 0:
        assert fused_seen, "the fused launch never ran"


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("B", [17, 24, 32, 33, 48, 64, 65, 80, 96, 97, 100, 128, 129, 160, 192, 300, 449, 1000, 1100, 1300, 1472, 1600, 1728])
def test_shared_operand_tiles_give_the_batch_kernels_bits(exp_pair, exp_lib, oracle, B):
    """k_trmm_stream_wg (64 x 64 tiles, operands shared through LDS; the default at 33 ... 64 preimages; beyond 64 the experiments build's column groups of 128 on
    halves of eight waves) at every batch size the experiments build lets it serve (PSF_STREAM_WG = smallest batch, PSF_STREAM_WG_MAX = largest), the default form of
    the batch size (17 ... 32: k_trmm_stream_wg32, 64 x 32 tiles; 65 ... 96: one launch of each over a stream of six fragments; beyond 128 with an odd number of column groups of 64: the 64 x 64 tiles again) and the one-wave tasks: the bits of k_trmm_f64_big (PSF_TRMM_STREAM_MAX = 0).  The shapes have 31 / 59 / 83 sixteen-row
    tiles: the last tile group is ragged, the task count odd or even, and the ring runs 16 k-steps past the diagonal."""
    psf, orc, n, q = exp_pair
    u = oracle.uniform_targets(6, B, n, q)
    ref = _with_env({"PSF_TRMM_STREAM_MAX": "0", "PSF_STREAM_WG": None, "PSF_STREAM_WG_MAX": None}, lambda: psf.samp_p_stages(u, seed=19, first_index=7))
    got = _with_env({"PSF_TRMM_STREAM_MAX": "4096", "PSF_STREAM_WG": "17", "PSF_STREAM_WG_MAX": "1024"}, lambda: psf.samp_p_stages(u, seed=19, first_index=7))
    one = _with_env({"PSF_TRMM_STREAM_MAX": "4096", "PSF_STREAM_WG": "0", "PSF_STREAM_WG32": "0", "PSF_STREAM_WG96": "0", "PSF_STREAM_WG192": "0"}, lambda: psf.samp_p_stages(u, seed=19, first_index=7))
    dflt = _with_env({"PSF_TRMM_STREAM_MAX": None, "PSF_STREAM_WG": None, "PSF_STREAM_WG_MAX": None, "PSF_STREAM_WG32": None, "PSF_STREAM_WG96": None, "PSF_STREAM_WG192": None}, lambda: psf.samp_p_stages(u, seed=19, first_index=7))
    assert (dflt["x"].view(np.uint64) == ref["x"].view(np.uint64)).all(), "the default form of this batch size differs from k_trmm_f64_big"
    assert (got["x"].view(np.uint64) == ref["x"].view(np.uint64)).all(), "the shared-operand tiles differ from k_trmm_f64_big"
    assert (one["x"].view(np.uint64) == ref["x"].view(np.uint64)).all(), "the one-wave tasks differ from k_trmm_f64_big"
    assert (got["e"] == ref["e"]).all()
    if B <= 129:
        assert (got["e"] == orc.samp_p(19, u, first_index=7)).all()


@pytest.mark.parametrize("B", [20, 40, 64, 80])
def test_shared_operand_tiles_with_the_structured_factor(T, oracle, B):
    """the chunk-stream layout of the normals (structured handles keep it): k_trmm_stream_wg<.., 0> is what 33 ... 64 preimages take there"""
    n, q, r, s = 32, 256, 5.0, 120.0
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s, structured=True)
    A, (R, L1, _) = psf.trap_gen(11)
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s, with_L=False)
    orc.load_key(A, R)
    u = oracle.uniform_targets(3, B, n, q)
    st = psf.samp_p_stages(u, seed=77, first_index=1000)
    for b in sorted({0, 15, 16, min(33, B - 1), min(64, B - 1), B - 1}):
        tr = orc.samp_p_structured_trace(L1, s, 77, 1000 + b, u[b])
        assert (st["x"][b].view(np.uint64) == tr["x"].view(np.uint64)).all(), "centres differ"
        assert (st["e"][b] == tr["e"]).all()
    psf.close()


@pytest.mark.parametrize("B", [1, 3, 16, 17, 64, 65, 100, 257])
def test_gadget_walk_with_a_quad_per_problem_equals_the_other_forms(exp_pair, exp_lib, oracle, B):
    """k_gadget_quad with four lanes per problem (the default between 10 241 and 98 304 problems) and with sixteen (2 049 ... 10 240) forced at every size against
    the round-5 choice of the same batch, the queue kernel and the one-wave-per-problem kernel: the same z, hence the same rows"""
    psf, orc, n, q = exp_pair
    u = oracle.uniform_targets(12, B, n, q)
    ref = _with_env({"PSF_GADGET_QUAD": "0", "PSF_GADGET_ROW": "0"}, lambda: psf.samp_p_stages(u, seed=23, first_index=99))
    quad = _with_env({"PSF_GADGET_QUAD": "100000000", "PSF_GADGET_ROW": "0", "PSF_GADGET_WAVE": "0"}, lambda: psf.samp_p_stages(u, seed=23, first_index=99))
    queue = _with_env({"PSF_GADGET_QUAD": "0", "PSF_GADGET_ROW": "0", "PSF_GADGET_WAVE": "0", "PSF_GADGET_WAVE16": "0"}, lambda: psf.samp_p_stages(u, seed=23, first_index=99))
    row = _with_env({"PSF_GADGET_ROW": "100000000", "PSF_GADGET_WAVE": "0"}, lambda: psf.samp_p_stages(u, seed=23, first_index=99))      # sixteen lanes per problem
    wave = _with_env({"PSF_GADGET_WAVE": "100000000"}, lambda: psf.samp_p_stages(u, seed=23, first_index=99))                             # one wave per problem
    assert (quad["z"] == ref["z"]).all() and (queue["z"] == ref["z"]).all() and (row["z"] == ref["z"]).all() and (wave["z"] == ref["z"]).all()
    assert (quad["e"] == ref["e"]).all()
    assert (quad["e"] == orc.samp_p(23, u, first_index=99)).all()


@pytest.mark.parametrize("n,q,base,k,m_bar,r,s,B", [(70, 625, 5, 4, 70 * 4 + 4, 2.0, 600.0, 64), (70, 625, 5, 4, 70 * 4 + 4, 2.0, 600.0, 200), (24, 2**40, 2, 40, 24 * 40 + 8, 3.0, 600.0, 700), (80, 538, 5, 4, 80 * 4 + 4, 2.0, 600.0, 100), (24, 2**40, 2, 40, 24 * 40 + 8, 3.0, 600.0, 200),
                                                    (9, 2**61 - 1, 2, 61, 9 * 61 + 5, 2.0, 400.0, 500)])
def test_gadget_quad_kernel_general_base_and_long_chains(oracle, n, q, base, k, m_bar, r, s, B):
    """2 049 ... 98 304 problems by default parameters (sixteen lanes per problem up to 10 240, a quad beyond): base 5 with q = base^k and with a digit column, and chains
    of 40 and 61 draws (k_gadget_quad<16, 4>, <4, 16>), against the oracle"""
    import tools_amd as T
    gp = T.GadgetParameters(n, k, m_bar, base, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = psf.trap_gen(2)
    orc = oracle.PSFPerturbation(oracle.GadgetParams(n, k, m_bar, base, q), r, s)
    orc.load_key(A, R, Lp)
    assert 2048 < n * B <= 98304
    u = oracle.uniform_targets(4, B, n, q)
    e = psf.samp_p(u, seed=5, first_index=17)
    assert psf.last_status() == 0
    sub = [0, 1, B // 2, B - 1]
    for b in sub:
        assert (e[b] == orc.samp_p(5, u[b:b + 1], first_index=17 + b)[0]).all()
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all()
    psf.close()
