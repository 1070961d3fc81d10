"""Struct-level serde JSON of the reference's parameter structs and key tuples: field names / nesting / typetag tags as the
derives lay them out (mp_perturbation.rs:57-62, gpv.rs:53-57, gpv_ring.rs:62-67, gadget_parameters.rs:44-52, :73-81,
trapdoor_distribution.rs:52-59); leaf encodings are marked unverified in tools_amd/serde_json.py.  Host-side: no device needed."""
import json
from fractions import Fraction

import numpy as np
import pytest


@pytest.fixture(scope="module")
def S():
    from tools_amd import serde_json
    return serde_json


class GP:            # stand-in with the fields of GadgetParameters (the real class needs the library only for init_default)
    def __init__(self, n, k, m_bar, base, q):
        self.n, self.k, self.m_bar, self.base, self.q = n, k, m_bar, base, q


def test_parameter_structs_have_the_references_layout(S):
    text = S.dumps_psf_perturbation(GP(8, 6, 57, 2, 64), 3, 25)
    o = json.loads(text)
    assert list(o.keys()) == ["gp", "r", "s"]                                           # mp_perturbation.rs:58-62, declaration order
    assert list(o["gp"].keys()) == ["n", "k", "m_bar", "base", "q", "distribution"]     # gadget_parameters.rs:45-51
    assert o["gp"]["distribution"] == {"PlusMinusOneZero": None}                         # typetag, externally tagged unit struct
    assert o["gp"]["n"] == {"value": "8"} and o["gp"]["q"] == {"value": "64"} and o["r"] == {"value": "3"}
    gp, r, s = S.loads_psf_perturbation(text)
    assert (gp.n, gp.k, gp.m_bar, gp.base, gp.q) == (8, 6, 57, 2, 64) and (r, s) == (3, 25)
    # Q from a float is the exact binary rational, like Q::from(1.005_f64) at gpv_ring.rs:52
    o = json.loads(S.dumps_psf_gpv_ring(GP(8, 9, 11, 2, 512), 100, 1.005))
    assert list(o.keys()) == ["gp", "s", "s_td"] and list(o["gp"].keys()) == ["n", "k", "m_bar", "base", "modulus", "distribution"]
    assert o["gp"]["distribution"] == {"SampleZ": None}
    assert o["gp"]["modulus"] == {"poly": "9  1 0 0 0 0 0 0 0 1 mod 512"}               # X^8 + 1 (common_moduli.rs:41-48)
    assert Fraction(o["s_td"]["value"]) == Fraction(1.005)
    gp, s, s_td = S.loads_psf_gpv_ring(json.dumps(o))
    assert (gp.n, gp.q, float(s_td)) == (8, 512, 1.005)
    o = json.loads(S.dumps_psf_gpv(GP(5, 8, 49, 2, 256), Fraction(21, 2)))
    assert list(o.keys()) == ["gp", "s"] and o["s"] == {"value": "21/2"}
    with pytest.raises(ValueError):                                                      # a distribution the device path does not implement
        bad = json.loads(S.dumps_psf_gpv(GP(5, 8, 49, 2, 256), 10))
        bad["gp"]["distribution"] = {"SampleZ": None}
        S.loads_psf_gpv(json.dumps(bad))


def test_key_tuples_round_trip(S):
    rng = np.random.default_rng(1)
    n, k, mb, q = 2, 3, 7, 8
    m = mb + n * k
    A = rng.integers(0, q, size=(n, m), dtype=np.uint64)
    R = rng.integers(-1, 2, size=(mb, n * k)).astype(np.int8)
    packed = rng.standard_normal(m * (m + 1) // 2)
    Sk = np.array([[2, 0, 0], [-1, 2, 0], [0, -1, 2]], dtype=np.int64)
    text = S.dumps_perturbation_key(A, q, R, packed, Sk, Sk.astype(np.float64), n)
    o = json.loads(text)
    assert isinstance(o, list) and len(o) == 2 and len(o[1]) == 3 and len(o[1][2]) == 2   # (A, (R, sqrt, (S, S~)))
    assert o[0]["matrix"].endswith(" mod 8") and o[1][0]["matrix"].startswith("[[")
    A2, q2, R2, p2 = S.loads_perturbation_key(text)
    assert q2 == q and (A2 == A).all() and (R2 == R).all() and (p2 == packed).all()      # floats survive exactly (binary rationals)
    bt = rng.integers(-9, 9, size=(m, m)).astype(np.int32)
    gt = rng.standard_normal((m, m))
    A3, q3, bt3, gt3 = S.loads_gpv_key(S.dumps_gpv_key(A, q, bt, gt))
    assert (A3 == A).all() and (bt3 == bt).all() and (gt3 == gt).all()
    nn, kk, qq = 4, 4, 16
    a = rng.integers(0, qq, size=(kk + 2, nn), dtype=np.uint64)
    r = rng.integers(-3, 4, size=(kk, nn))
    e = rng.integers(-3, 4, size=(kk, nn))
    a4, q4, r4, e4 = S.loads_ring_key(S.dumps_ring_key(a, qq, r, e), nn)
    assert q4 == qq and (a4 == a).all() and (r4 == r).all() and (e4 == e).all()
