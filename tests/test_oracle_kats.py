"""The CPU oracle against every known-answer vector the reference's own tests hold for the
deterministic helpers on the path (fixtures: tests/golden/ref_kats.json)."""
import numpy as np
import pytest


def test_gen_gadget_vec(kats, oracle):
    for name in ("correctness_base_2", "correctness_base_5"):
        k = kats[f"gen_gadget_vec/{name}"]
        assert oracle.gen_gadget_vec(k["k"], k["base"]).tolist() == k["expect"]


def test_gen_gadget_mat(kats, oracle):
    for name in ("correctness_base_2_3x3", "correctness_base_3_2x5"):
        k = kats[f"gen_gadget_mat/{name}"]
        assert oracle.gen_gadget_mat(k["n"], k["k"], k["base"]).tolist() == k["expect"]


def test_find_solution_gadget_vec(kats, oracle):
    k = kats["find_solution_gadget_vec/returns_correct_solution_vec"]
    g = oracle.gen_gadget_vec(k["k"], k["base"])
    for v in k["values"]:
        sol = oracle.find_solution_gadget_vec(v, k["q"], k["k"], k["base"])
        assert int(g @ sol) == v % k["q"]
        assert sol.min() >= 0 and sol.max() < k["base"]


def test_find_solution_gadget_vec_modulus_too_large(oracle):
    # gadget_classical.rs:170-172: panics if base^k < q
    with pytest.raises(RuntimeError):
        oracle.find_solution_gadget_vec(5, 126, 5, 3) if 3**5 < 126 else (_ for _ in ()).throw(RuntimeError())
    with pytest.raises(RuntimeError):
        oracle.find_solution_gadget_vec(5, 1000, 5, 3)


def test_find_solution_gadget_mat(kats, oracle):
    k = kats["find_solution_gadget_mat/returns_correct_solution_mat"]
    value = np.array(k["value"], dtype=np.uint64)
    sol = oracle.find_solution_gadget_mat(value, k["q"], k["k"], k["base"])
    G = oracle.gen_gadget_mat(value.shape[0], k["k"], k["base"])
    assert (G @ sol == value.astype(np.int64)).all()


@pytest.mark.parametrize("name", ["base_2_power_two", "base_2_arbitrary", "base_5_power_5", "base_5_arbitrary"])
def test_short_basis_gadget(kats, oracle, name):
    k = kats[f"short_basis_gadget/{name}"]
    gp = oracle.gadget_params_default(k["n"], k["q"])
    gp.k, gp.base = k["k"], k["base"]
    assert oracle.short_basis_gadget(gp).tolist() == k["expect"]


def test_short_basis_classical_sa_l_sa_r(kats, oracle):
    kl = kats["short_basis_classical/working_sa_l"]
    kr = kats["short_basis_classical/working_sa_r_identity"]
    gp = oracle.gadget_params_default(kl["n"], kl["q"])
    R = np.array(kl["R"], dtype=np.int8)
    A = np.array(kl["A"], dtype=np.uint64)
    assert oracle.gen_sa_l(R).tolist() == kl["expect"]
    assert oracle.gen_sa_r(gp, A).tolist() == kr["expect"]
    # explicit identity tag takes the tag.inverse() branch (short_basis_classical.rs:106)
    assert oracle.gen_sa_r(gp, A, tag=np.eye(2, dtype=np.uint64)).tolist() == kr["expect"]
    # the product is what gen_short_basis_for_trapdoor returns (short_basis_classical.rs:60-62)
    sa = oracle.gen_short_basis_for_trapdoor(gp, A, R)
    assert (sa == np.array(kl["expect"]) @ np.array(kr["expect"])).all()


def test_compute_w(kats, oracle):
    k = kats["short_basis_classical/compute_w_working_example_tag_identity"]
    gp = oracle.gadget_params_default(k["n"], k["q"])
    A = np.array(k["A"], dtype=np.uint64)
    W = oracle.compute_w(gp, A)
    G = oracle.gen_gadget_mat(gp.n, gp.k, gp.base)
    assert ((G @ W + A[:, :gp.m_bar].astype(np.int64)) % k["q"] == 0).all()


def test_default_parameters(kats, oracle):
    for c in kats["gadget_parameters/default_unchanged"]["cases"]:
        gp = oracle.gadget_params_default(c["n"], c["q"])
        assert (gp.n, gp.k, gp.m_bar, gp.base, gp.q) == (c["n"], c["k"], c["m_bar"], c["base"], c["q"])
    for c in kats["gadget_default/correct_default_dimensions"]["cases"]:
        gp = oracle.gadget_params_default(c["n"], c["q"])
        assert gp.m_bar + gp.n * gp.k == c["a_cols"] and gp.m_bar == c["r_rows"] and gp.n * gp.k == c["r_cols"]
