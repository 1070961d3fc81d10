"""The reference's known-answer vectors through the C ABI (host helpers + the HIP digit kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import tools_amd
    return tools_amd.gadget


def test_find_solution_gadget_vec_kat(kats, G):
    k = kats["find_solution_gadget_vec/returns_correct_solution_vec"]
    g = G.gen_gadget_vec(k["k"], k["base"])
    vals = np.array(k["values"], dtype=np.uint64).reshape(1, -1)
    sol = G.find_solution_gadget_mat(vals, k["q"], k["k"], k["base"])   # one column per value
    assert ((g @ sol) == vals[0].astype(np.int64)).all()
    assert sol.min() >= 0 and sol.max() < k["base"]


def test_find_solution_gadget_mat_kat(kats, G, oracle):
    k = kats["find_solution_gadget_mat/returns_correct_solution_mat"]
    value = np.array(k["value"], dtype=np.uint64)
    sol = G.find_solution_gadget_mat(value, k["q"], k["k"], k["base"])
    assert (G.gen_gadget_mat(3, k["k"], k["base"]) @ sol == value.astype(np.int64)).all()
    assert (sol == oracle.find_solution_gadget_mat(value, k["q"], k["k"], k["base"])).all()


def test_find_solution_modulus_too_large(G):
    import tools_amd
    with pytest.raises(tools_amd.PsfError) as ei:     # gadget_classical.rs:170-172
        G.find_solution_gadget_vec(5, 1000, 5, 3)
    assert ei.value.status == 4


def test_digits_random_parity(G, oracle):
    rng = np.random.default_rng(1)
    for q, k, base in [(2**30, 30, 2), (1073741789, 30, 2), (3329, 12, 2), (3**20, 20, 3), (2**60, 60, 2)]:
        value = rng.integers(0, q, size=(7, 33), dtype=np.uint64)
        assert (G.find_solution_gadget_mat(value, q, k, base) == oracle.find_solution_gadget_mat(value, q, k, base)).all()
    # empty and ragged shapes
    assert G.find_solution_gadget_mat(np.zeros((0, 4), dtype=np.uint64), 17, 5, 2).shape == (0, 4)
    assert G.find_solution_gadget_mat(np.zeros((3, 0), dtype=np.uint64), 17, 5, 2).shape == (15, 0)
