"""How exact are the centres the nearest-plane walk samples around?  (ADVICE r02 medium.)

The reference keeps the centre of GPV08's loop in exact rationals (MatQ, gpv.rs:158-160); the library keeps running projections in doubles.
This test evaluates the loop of gpv.rs:160 -- c' = <c, b~_i> / |b~_i|^2, c -= z_i b_i -- in 100-digit arithmetic on the integer basis (exact
Gram-Schmidt to 100 digits), following the z_i the restated walk drew, and compares every centre the walk saw (orc_gpv_samp_p_trace; the HIP
path is bit-for-bit this walk: tests/test_gpu_gpv_parity.py) with the exact one, in units of the draw's width s / |b~_i|:
  * moduli with q sqrt(n) <= 2^13 s: one pass, relative centre error <= 2^-40;
  * larger moduli (2^45, 2^60): two passes, the pass that shapes the output is within 2^-35; the same keys forced to ONE pass are
    orders of magnitude off (the defect this round removed) -- asserted, so the rule cannot silently fall back."""
from decimal import Decimal, getcontext

import numpy as np
import pytest

getcontext().prec = 100


def exact_gso(bt):
    d = bt.shape[0]
    B = [[Decimal(int(v)) for v in row] for row in bt]
    G, n2 = [], []
    for i in range(d):
        g = list(B[i])
        for l in range(i):
            mu = sum(x * y for x, y in zip(B[i], G[l])) / n2[l]
            if mu:
                g = [x - mu * y for x, y in zip(g, G[l])]
        G.append(g)
        n2.append(sum(x * x for x in g))
    return G, n2


def worst_relative_centre_error(orc, bt, G, n2, s, seed, u, exact_walk=True):
    e, c0, cen, z = orc.samp_p_trace(seed, u, index=3)
    d = bt.shape[0]
    c = [Decimal(int(v)) for v in c0]
    worst = Decimal(0)
    for i in range(d - 1, -1, -1):
        exact = sum(x * y for x, y in zip(c, G[i])) / n2[i]
        err = abs(Decimal(float(cen[i])) - exact) * n2[i].sqrt() / Decimal(s)       # in units of the width s / |b~_i|
        worst = max(worst, err)
        zi = int(z[i])
        if zi:
            c = [x - zi * int(b) for x, b in zip(c, bt[i])]
    if exact_walk:
        assert [int(-x) for x in c] == [int(v) for v in e]                          # the trace is the preimage's own walk
    return float(worst), e


@pytest.mark.parametrize("n,q,s", [(2, 2**45, 60.0), (1, 2**60, 40.0), (2, 2**45 - 55, 60.0)])
def test_large_moduli_are_sampled_in_two_passes_with_accurate_centres(oracle, n, q, s):
    gp = oracle.gadget_params_default(n, q)
    orc = oracle.PSFGPV(gp, s)
    assert orc.trap_gen(9) == 0
    assert orc.two_pass                                           # q sqrt(n) > 2^13 s
    bt = orc.basis_t
    G, n2 = exact_gso(bt)
    u = oracle.uniform_targets(3, 1, n, q)[0]
    worst, e = worst_relative_centre_error(orc, bt, G, n2, s, 11, u)
    assert worst < 2.0**-35, worst
    A = orc.A
    assert ((A.astype(object) @ e.astype(object)) % q == u.astype(object)).all()
    assert orc.check_domain(e).all()
    assert float((e.astype(np.float64) ** 2).sum()) < 2.0 * s * s / (2 * np.pi) * len(e)      # a short preimage, not merely a valid one
    # the same key through ONE pass: the centres are off by many orders of magnitude more (what two passes are for)
    orc.set_two_pass(0)
    # (coefficients beyond 2^53 take part in the recombination as the doubles the walk used -- psf_oracle_gpv.c -- so a single pass at 2^60 ends on another
    # representative of the coset than the exact integers would: still A e = u, no longer "c0 - sum z_i b_i" digit for digit)
    single, e1p = worst_relative_centre_error(orc, bt, G, n2, s, 11, u, exact_walk=q < 2**50)
    assert ((A.astype(object) @ e1p.astype(object)) % q == u.astype(object)).all()
    assert single > 1000 * worst and single > 2.0**-30, (single, worst)
    orc.set_two_pass(-1)


@pytest.mark.parametrize("n,q,s", [(6, 128, 10.0), (4, 3329, 60.0)])
def test_small_moduli_stay_single_pass_within_2_to_minus_40(oracle, n, q, s):
    gp = oracle.gadget_params_default(n, q)
    orc = oracle.PSFGPV(gp, s)
    assert orc.trap_gen(5) == 0
    assert not orc.two_pass
    bt = orc.basis_t
    G, n2 = exact_gso(bt)
    u = oracle.uniform_targets(2, 1, n, q)[0]
    worst, _ = worst_relative_centre_error(orc, bt, G, n2, s, 31, u)
    assert worst < 2.0**-40, worst


def test_the_rule_is_the_documented_one(oracle):
    import ctypes as C
    L = oracle.lib()
    L.orc_np_two_pass.argtypes = [C.c_uint64, C.c_size_t, C.c_double]
    assert L.orc_np_two_pass(3329, 256, 1024.0) == 0              # C2
    assert L.orc_np_two_pass(3329, 256, 522.56) == 0               # C4
    assert L.orc_np_two_pass(3329, 256, 240.0) == 0                # C2 at the bench rule
    assert L.orc_np_two_pass(2**31 - 1, 6, 100.0) == 1
    assert L.orc_np_two_pass(2**45, 3, 60.0) == 1


def test_report_measured_errors(oracle, capsys):
    """not an assertion: prints the measured figures quoted in DESIGN.md / psf_mi355x.h (run with -s)"""
    rows = []
    for n, q, s in [(2, 2**45, 60.0), (1, 2**60, 40.0), (6, 128, 10.0), (4, 3329, 60.0)]:
        orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
        assert orc.trap_gen(9) == 0
        bt = orc.basis_t
        G, n2 = exact_gso(bt)
        u = oracle.uniform_targets(3, 1, n, q)[0]
        w, _ = worst_relative_centre_error(orc, bt, G, n2, s, 11, u)
        orc.set_two_pass(0)
        w1, _ = worst_relative_centre_error(orc, bt, G, n2, s, 11, u, exact_walk=q < 2**50)
        rows.append((n, q, s, orc.m, w, w1))
    with capsys.disabled():
        for n, q, s, m, w, w1 in rows:
            print(f"\n[centre precision] n={n} q=2^{np.log2(q):.1f} s={s} d={m}: shipped rule 2^{np.log2(w):.1f}, one pass 2^{np.log2(w1):.1f} (units of the draw's width)", end="")
