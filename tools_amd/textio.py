"""Text forms of the reference's matrix and polynomial types, as its tests and doc examples write them
(`MatZ::from_str`, `MatZq::from_str`, `PolyOverZ::from_str`, `MatPolyOverZ::from_str`:
gadget_classical.rs:207,300-343,471 ; gadget_ring.rs:135 ; short_basis_ring.rs:366-375):

    MatZ            [[1, 2],[3, 4]]
    MatZq           [[1, 42],[2, 30],[3, 12]] mod 125
    PolyOverZ       4  -1 7 6 -8            (length, two blanks, coefficients from the constant term; "0" = zero)
    MatPolyOverZ    [[4  -1 7 6 -8, 3  0 -2 4]]

This is the interchange a maintainer has without the serde layer (whose JSON schema lives in the
un-vendored qfall-math): `println!("{a}")` on the Rust side, `matzq_from_str` here, and back.  Host-side only;
values are Python ints / numpy arrays in the shapes the C ABI takes."""
import re

import numpy as np

__all__ = ["matz_to_str", "matz_from_str", "matzq_to_str", "matzq_from_str", "poly_to_str", "poly_from_str",
           "matpoly_to_str", "matpoly_from_str"]

_ROW = re.compile(r"\[([^\[\]]*)\]")


def _rows(text):
    body = text.strip()
    if not (body.startswith("[[") and body.endswith("]]")):
        raise ValueError("matrix text must look like [[a, b],[c, d]]")
    rows = [r for r in _ROW.findall(body)]
    if not rows:
        raise ValueError("empty matrix")
    return rows


def matz_from_str(text, dtype=np.int64):
    """'[[1, 2],[3, 4]]' -> array of shape (rows, cols)."""
    rows = [[int(x) for x in r.split(",")] for r in _rows(text)]
    if len({len(r) for r in rows}) != 1:
        raise ValueError("ragged matrix")
    return np.array(rows, dtype=object).astype(dtype)


def matz_to_str(mat):
    m = np.asarray(mat)
    if m.ndim == 1:
        m = m.reshape(-1, 1)                       # a vector is a column, as in the reference
    return "[" + ",".join("[" + ", ".join(str(int(x)) for x in row) + "]" for row in m) + "]"


def matzq_from_str(text):
    """'[[1, 42],[2, 30]] mod 125' -> (uint64 array of least non-negative residues, q)."""
    if " mod " not in text:
        raise ValueError("MatZq text needs ' mod q'")
    body, qtxt = text.rsplit(" mod ", 1)
    q = int(qtxt.strip())
    if q < 2:
        raise ValueError("modulus must be at least 2")
    vals = matz_from_str(body, dtype=object)
    return np.array([[int(x) % q for x in row] for row in vals], dtype=np.uint64), q


def matzq_to_str(mat, q):
    m = np.asarray(mat)
    if m.ndim == 1:
        m = m.reshape(-1, 1)
    return matz_to_str(np.array([[int(x) % int(q) for x in row] for row in m], dtype=object)) + f" mod {int(q)}"


def poly_from_str(text, n=None):
    """'4  -1 7 6 -8' -> [-1, 7, 6, -8] (constant term first), zero-padded to n coefficients if n is given."""
    toks = text.split()
    if not toks:
        raise ValueError("empty polynomial text")
    length = int(toks[0])
    coeffs = [int(t) for t in toks[1:]]
    if length != len(coeffs):
        raise ValueError(f"polynomial announces {length} coefficients, {len(coeffs)} given")
    if n is not None:
        if length > n:
            raise ValueError("more coefficients than the ring degree")
        coeffs = coeffs + [0] * (n - length)
    return coeffs


def poly_to_str(coeffs):
    c = [int(x) for x in coeffs]
    while c and c[-1] == 0:
        c.pop()                                    # the reference prints normalised polynomials
    return "0" if not c else f"{len(c)}  " + " ".join(str(x) for x in c)


def matpoly_from_str(text, n):
    """'[[4  -1 7 6 -8, 3  0 -2 4]]' -> int64 array (rows, cols, n), constant term first."""
    out = []
    for r in _rows(text):
        out.append([poly_from_str(ent, n) for ent in r.split(",")])
    if len({len(r) for r in out}) != 1:
        raise ValueError("ragged matrix")
    return np.array(out, dtype=np.int64)


def matpoly_to_str(arr):
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[None, :, :]                          # k polynomials = a 1 x k row, like the ring trapdoor (gpv_ring.rs:72)
    return "[" + ",".join("[" + ", ".join(poly_to_str(p) for p in row) + "]" for row in a) + "]"
