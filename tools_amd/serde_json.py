"""serde-JSON exchange of parameters and keys with a qfall-tools session.

Struct level -- VERIFIED against the reference's derives: field names and nesting are those of `#[derive(Serialize, Deserialize)]` at
mp_perturbation.rs:57-62 (`PSFPerturbation {gp, r, s}`), gpv.rs:53-57 (`PSFGPV {gp, s}`), gpv_ring.rs:62-67 (`PSFGPVRing {gp, s, s_td}`),
gadget_parameters.rs:44-52 (`GadgetParameters {n, k, m_bar, base, q, distribution}`) and :73-81 (`GadgetParametersRing {..., modulus,
distribution}`); the `distribution` trait objects are `#[typetag::serde]` (trapdoor_distribution.rs:21,35,52-59), whose default
representation is externally tagged, so the unit structs appear as `{"PlusMinusOneZero": null}` / `{"SampleZ": null}`.  Tuples
(the trapdoors: mp_perturbation.rs:195, gpv.rs:61, gpv_ring.rs:72) are JSON arrays, as serde writes them.

Leaf level -- UNVERIFIED: `Z`, `Q`, `Modulus`, `ModulusPolynomialRingZq`, `MatZ`, `MatZq`, `MatQ`, `MatPolyOverZ`, `MatPolynomialRingZq`
are serialised by qfall-math, which is not in this tree (Cargo.toml:18, no lockfile).  The encodings below -- one-field objects
holding the type's `Display` string, `{"value": "17"}`, `{"matrix": "[[1, 2],[3, 4]] mod 7"}`, `{"poly": "3  1 0 1"}` -- are this
author's recollection of upstream; the strings inside are exactly the `Display` / `from_str` forms the reference's own tests use
(tools_amd/textio.py).  `LEAF` maps each type to its (field name) so a maintainer can correct a name in one place.

Host-side only; values are Python ints / Fractions / numpy arrays in the shapes the C ABI takes."""
import json
from fractions import Fraction

import numpy as np

from . import textio

__all__ = ["LEAF", "dumps_psf_perturbation", "loads_psf_perturbation", "dumps_psf_gpv", "loads_psf_gpv", "dumps_psf_gpv_ring",
           "loads_psf_gpv_ring", "dumps_perturbation_key", "loads_perturbation_key", "dumps_gpv_key", "loads_gpv_key",
           "dumps_ring_key", "loads_ring_key"]

LEAF = {"Z": "value", "Q": "value", "Modulus": "value", "MatZ": "matrix", "MatZq": "matrix", "MatQ": "matrix", "PolyOverZ": "poly",
        "MatPolyOverZ": "matrix", "MatPolynomialRingZq": "matrix", "ModulusPolynomialRingZq": "poly"}


# ---------------------------------------------------------------- leaves
def _q_str(x):
    """Q as 'num/den' (or 'num'): floats become the exact binary rational, as Q::from(f64) does."""
    f = x if isinstance(x, Fraction) else Fraction(x)
    return str(f.numerator) if f.denominator == 1 else f"{f.numerator}/{f.denominator}"


def _q_parse(text):
    return Fraction(text.strip())


def enc_z(v):
    return {LEAF["Z"]: str(int(v))}


def dec_z(o):
    return int(o[LEAF["Z"]])


def enc_q(v):
    return {LEAF["Q"]: _q_str(v)}


def dec_q(o):
    return _q_parse(o[LEAF["Q"]])


def enc_modulus(q):
    return {LEAF["Modulus"]: str(int(q))}


def dec_modulus(o):
    return int(o[LEAF["Modulus"]])


def enc_modulus_poly(n, q):
    """X^n + 1 mod q (common_moduli.rs:41-48) as '<n+1>  1 0 ... 0 1 mod q'."""
    return {LEAF["ModulusPolynomialRingZq"]: textio.poly_to_str([1] + [0] * (n - 1) + [1]) + f" mod {int(q)}"}


def dec_modulus_poly(o):
    body, qtxt = o[LEAF["ModulusPolynomialRingZq"]].rsplit(" mod ", 1)
    coeffs = textio.poly_from_str(body)
    n = len(coeffs) - 1
    if coeffs != [1] + [0] * (n - 1) + [1]:
        raise ValueError("only X^n + 1 moduli are supported by the device path")
    return n, int(qtxt)


def enc_matz(a):
    return {LEAF["MatZ"]: textio.matz_to_str(a)}


def dec_matz(o, dtype=np.int64):
    return textio.matz_from_str(o[LEAF["MatZ"]], dtype=dtype)


def enc_matzq(a, q):
    return {LEAF["MatZq"]: textio.matzq_to_str(a, q)}


def dec_matzq(o):
    return textio.matzq_from_str(o[LEAF["MatZq"]])


def enc_matq(a):
    m = np.asarray(a, dtype=object)
    if m.ndim == 1:
        m = m.reshape(-1, 1)
    return {LEAF["MatQ"]: "[" + ",".join("[" + ", ".join(_q_str(x) for x in row) + "]" for row in m) + "]"}


def dec_matq(o):
    rows = [[float(_q_parse(x)) for x in r.split(",")] for r in textio._rows(o[LEAF["MatQ"]])]
    return np.array(rows, dtype=np.float64)


def enc_matpoly(a):
    return {LEAF["MatPolyOverZ"]: textio.matpoly_to_str(a)}


def dec_matpoly(o, n):
    return textio.matpoly_from_str(o[LEAF["MatPolyOverZ"]], n)


def enc_matpolyring(a, q):
    return {LEAF["MatPolynomialRingZq"]: textio.matpoly_to_str(a) + f" mod {textio.poly_to_str([1] + [0] * (np.asarray(a).shape[-1] - 1) + [1])} mod {int(q)}"}


def dec_matpolyring(o, n):
    text = o[LEAF["MatPolynomialRingZq"]]
    body = text.split(" mod ", 1)[0]
    q = int(text.rsplit(" mod ", 1)[1])
    return textio.matpoly_from_str(body, n).astype(np.uint64), q


# ---------------------------------------------------------------- parameter structs
def _gp_obj(gp):
    """GadgetParameters (gadget_parameters.rs:44-52)"""
    return {"n": enc_z(gp.n), "k": enc_z(gp.k), "m_bar": enc_z(gp.m_bar), "base": enc_z(gp.base), "q": enc_modulus(gp.q),
            "distribution": {"PlusMinusOneZero": None}}


def _gp_ring_obj(gp):
    """GadgetParametersRing (gadget_parameters.rs:73-81)"""
    return {"n": enc_z(gp.n), "k": enc_z(gp.k), "m_bar": enc_z(gp.m_bar), "base": enc_z(gp.base), "modulus": enc_modulus_poly(gp.n, gp.q),
            "distribution": {"SampleZ": None}}


def _gp_from(o, ring=False):
    from .psf import GadgetParameters, GadgetParametersRing
    want = "SampleZ" if ring else "PlusMinusOneZero"
    if list(o["distribution"].keys()) != [want]:
        raise ValueError(f"distribution {list(o['distribution'])}: the device path implements {want} only (trapdoor_distribution.rs:52-59)")
    if ring:
        n_mod, q = dec_modulus_poly(o["modulus"])
        if n_mod != dec_z(o["n"]):
            raise ValueError("modulus degree and n disagree")
        return GadgetParametersRing(dec_z(o["n"]), dec_z(o["k"]), dec_z(o["m_bar"]), dec_z(o["base"]), q)
    return GadgetParameters(dec_z(o["n"]), dec_z(o["k"]), dec_z(o["m_bar"]), dec_z(o["base"]), dec_modulus(o["q"]))


def dumps_psf_perturbation(gp, r, s):
    """PSFPerturbation {gp, r, s} (mp_perturbation.rs:57-62)"""
    return json.dumps({"gp": _gp_obj(gp), "r": enc_q(r), "s": enc_q(s)})


def loads_psf_perturbation(text):
    o = json.loads(text)
    return _gp_from(o["gp"]), dec_q(o["r"]), dec_q(o["s"])


def dumps_psf_gpv(gp, s):
    """PSFGPV {gp, s} (gpv.rs:53-57)"""
    return json.dumps({"gp": _gp_obj(gp), "s": enc_q(s)})


def loads_psf_gpv(text):
    o = json.loads(text)
    return _gp_from(o["gp"]), dec_q(o["s"])


def dumps_psf_gpv_ring(gp, s, s_td):
    """PSFGPVRing {gp, s, s_td} (gpv_ring.rs:62-67)"""
    return json.dumps({"gp": _gp_ring_obj(gp), "s": enc_q(s), "s_td": enc_q(s_td)})


def loads_psf_gpv_ring(text):
    o = json.loads(text)
    return _gp_from(o["gp"], ring=True), dec_q(o["s"]), dec_q(o["s_td"])


# ---------------------------------------------------------------- keys: what trap_gen returns, (A, Trapdoor) as a JSON array
def dumps_perturbation_key(A, q, R, sqrt_sigma2_packed, Sk=None, Sk_gso=None, n=None):
    """(MatZq, (MatZ, MatQ, (MatZ, MatQ))) (mp_perturbation.rs:194-195): [A, [R, sqrt(Sigma_2), [S, S~]]].  sqrt(Sigma_2) is given
    packed by rows (the ABI's form) and written as the full lower-triangular MatQ; (S, S~) = I_n (x) S_k and its GSO when the k x k
    blocks are supplied, else null (they are functions of the parameters)."""
    m = np.asarray(A).shape[1]
    L = np.zeros((m, m), dtype=object)
    L[:, :] = Fraction(0)
    it = iter(np.asarray(sqrt_sigma2_packed, dtype=np.float64))
    for i in range(m):
        for j in range(i + 1):
            L[i, j] = Fraction(float(next(it)))
    gad = None
    if Sk is not None and Sk_gso is not None and n is not None:
        k = np.asarray(Sk).shape[0]
        S = np.kron(np.eye(n, dtype=np.int64), np.asarray(Sk, dtype=np.int64))
        G = np.zeros((n * k, n * k), dtype=object)
        G[:, :] = Fraction(0)
        for b in range(n):
            for i in range(k):
                for j in range(k):
                    G[b * k + i, b * k + j] = Fraction(float(np.asarray(Sk_gso)[i, j]))
        gad = [enc_matz(S), enc_matq(G)]
    return json.dumps([enc_matzq(A, q), [enc_matz(R), enc_matq(L), gad]])


def loads_perturbation_key(text):
    """-> (A, q, R, sqrt_sigma2_packed): ready for PSFPerturbation.load_key(A, R, packed)."""
    a_o, (r_o, l_o, _gad) = json.loads(text)
    A, q = dec_matzq(a_o)
    R = dec_matz(r_o, dtype=np.int8)
    L = dec_matq(l_o)
    m = L.shape[0]
    packed = np.concatenate([L[i, :i + 1] for i in range(m)])
    return A, q, R, packed


def dumps_gpv_key(A, q, basis_t, gso_t):
    """(MatZq, (MatZ, MatQ)) (gpv.rs:60-61): the ABI's transposed matrices go back to the reference's column convention."""
    return json.dumps([enc_matzq(A, q), [enc_matz(np.asarray(basis_t).T), enc_matq(np.vectorize(lambda v: Fraction(float(v)), otypes=[object])(np.asarray(gso_t).T))]])


def loads_gpv_key(text):
    """-> (A, q, basis_t, gso_t): ready for PSFGPV.load_key(A, basis_t, gso_t)."""
    a_o, (b_o, g_o) = json.loads(text)
    A, q = dec_matzq(a_o)
    return A, q, np.ascontiguousarray(dec_matz(b_o, dtype=np.int32).T), np.ascontiguousarray(dec_matq(g_o).T)


def dumps_ring_key(a, q, r, e):
    """(MatPolynomialRingZq, (MatPolyOverZ, MatPolyOverZ)) (gpv_ring.rs:70-72): a as 1 x (k+2), r and e as 1 x k polynomials."""
    return json.dumps([enc_matpolyring(np.asarray(a, dtype=object)[None, :, :], q), [enc_matpoly(r), enc_matpoly(e)]])


def loads_ring_key(text, n):
    """-> (a [(k+2) x n], q, r, e [k x n]): ready for PSFGPVRing.load_key(a, r, e)."""
    a_o, (r_o, e_o) = json.loads(text)
    a, q = dec_matpolyring(a_o, n)
    return a[0], q, dec_matpoly(r_o, n)[0], dec_matpoly(e_o, n)[0]
