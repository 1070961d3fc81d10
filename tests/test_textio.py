"""Text interchange in the reference's own notation (tools_amd/textio.py): the literals below are the strings the
reference's tests and doc examples pass to `from_str` (gadget_classical.rs:207,300,309,471 ; gadget_ring.rs:135 ;
short_basis_ring.rs:372-375), checked against the same data in tests/golden/ref_kats.json where it is recorded."""
import numpy as np
import pytest

from tools_amd import textio as tx


def test_matz_and_matzq_literals(kats):
    v = tx.matz_from_str("[[1],[2],[4],[8],[16]]")                      # gadget_classical.rs:300
    assert v[:, 0].tolist() == kats["gen_gadget_vec/correctness_base_2"]["expect"]
    v = tx.matz_from_str("[[1],[5],[25],[125]]")                        # :309
    assert v[:, 0].tolist() == kats["gen_gadget_vec/correctness_base_5"]["expect"]
    m, q = tx.matzq_from_str("[[1, 42],[2, 40],[3, 90]] mod 125")       # :471
    assert q == 125 and m.dtype == np.uint64
    assert m.tolist() == kats["find_solution_gadget_mat/returns_correct_solution_mat"]["value"]
    assert tx.matzq_to_str(m, q) == "[[1, 42],[2, 40],[3, 90]] mod 125"
    assert tx.matz_to_str(v) == "[[1],[5],[25],[125]]"
    assert tx.matz_to_str([1, 2, 3]) == "[[1],[2],[3]]"                 # vectors are columns


def test_negative_and_large_entries_round_trip():
    big = np.array([[-(2**62), 2**62 - 1], [0, -1]], dtype=np.int64)
    assert (tx.matz_from_str(tx.matz_to_str(big)) == big).all()
    q = 2**61 - 1
    m, q2 = tx.matzq_from_str(f"[[-1, {q}],[{q + 5}, 7]] mod {q}")
    assert q2 == q and m.tolist() == [[q - 1, 0], [5, 7]]               # least non-negative residues


def test_polynomial_literals(kats):
    assert tx.poly_from_str("10  5 124 12 14 14 1 2 4 1 5") == [5, 124, 12, 14, 14, 1, 2, 4, 1, 5]   # gadget_ring.rs:135
    r = tx.matpoly_from_str("[[4  -1 7 6 -8, 3  0 -2 4, 4  0 3 -4 1, 4  6 4 -1 3]]", 4)               # short_basis_ring.rs:372
    assert r.shape == (1, 4, 4)
    rec = kats["short_basis_ring/working_sa_l"]
    assert r[0].tolist() == [list(p) + [0] * (4 - len(p)) for p in rec["r"]]
    assert tx.matpoly_to_str(r) == "[[4  -1 7 6 -8, 3  0 -2 4, 4  0 3 -4 1, 4  6 4 -1 3]]"
    assert tx.poly_to_str([0, 0, 0]) == "0" and tx.poly_from_str("0", 4) == [0, 0, 0, 0]
    assert tx.poly_to_str([3, 0, 0, 0]) == "1  3"


@pytest.mark.parametrize("bad", ["[1, 2]", "[[1, 2],[3]]", "[[1, x]]", ""])
def test_malformed_matrices_are_rejected(bad):
    with pytest.raises(ValueError):
        tx.matz_from_str(bad)


def test_malformed_polynomials_and_moduli():
    with pytest.raises(ValueError):
        tx.poly_from_str("3  1 2")
    with pytest.raises(ValueError):
        tx.poly_from_str("5  1 2 3 4 5", 4)
    with pytest.raises(ValueError):
        tx.matzq_from_str("[[1, 2]]")
    with pytest.raises(ValueError):
        tx.matzq_from_str("[[1, 2]] mod 1")
