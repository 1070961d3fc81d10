"""Every golden record of the reference (tests/golden/ref_kats.json, extracted by make_ref_kats.py from the reference's own #[test] functions) against the PRODUCT,
through its C ABI, in the suite the driver runs on the GPU box: the host-side helpers that the CPU suite already checks (gen_gadget_*, short_basis_gadget,
gen_short_basis_for_trapdoor = sa_l * sa_r, rot_minus(_matrix), default parameters) and the six ring records -- short_basis_ring/compute_s x 4,
working_sa_r, working_sa_l (short_basis_ring.rs:358-444, :457-535) -- through psf_gen_short_basis_for_trapdoor_ring = sa_l * sa_r mod X^n + 1.  The
find_solution_* records run on the device in tests/test_gpu_gadget_kats.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_default_parameters_through_abi_on_the_gpu_box(kats):
    from tests.test_cabi_cpu import check_default_parameters_through_abi
    check_default_parameters_through_abi(kats)


def test_host_helper_kats_through_abi_on_the_gpu_box(kats):
    from tests.test_cabi_cpu import check_host_helper_kats_through_abi
    check_host_helper_kats_through_abi(kats)


def _polys(rows, n):
    out = np.zeros((len(rows), len(rows[0]), n), dtype=np.int64)
    for i, row in enumerate(rows):
        for j, p in enumerate(row):
            out[i, j, :len(p)] = p
    return out


def _negacyclic(x, y):
    n = len(x)
    acc = [0] * n
    for i in range(n):
        for j in range(n):
            if i + j >= n:
                acc[i + j - n] -= int(x[i]) * int(y[j])
            else:
                acc[i + j] += int(x[i]) * int(y[j])
    return acc


def _ring_params(T, n, q, k=None, base=None):
    gp = T.GadgetParametersRing.init_default(n, q)
    if k is not None:
        gp = T.GadgetParametersRing(n, k, k + 2, base, q)
    return gp


@pytest.mark.parametrize("name", ["base_2_power_two", "base_2_arbitrary", "base_5_power_5", "base_5_arbitrary"])
def test_ring_compute_s_through_the_product(kats, name):
    """compute_s (short_basis_ring.rs:137-166) is the lower-right k x k block of sa_r; with r = e = 0 the left factor sa_l is the identity (:88-103), so the
    product's short basis IS sa_r and the block must equal the reference's expected matrix of constant polynomials."""
    import tools_amd as T
    k = kats[f"short_basis_ring/compute_s/{name}"]
    n, K = k["n"], k["k"] + 2
    gp = _ring_params(T, n, k["q"], k["k"], k["base"])
    a = np.zeros((K, n), dtype=np.uint64)
    a[0, 0] = 1                                                    # a = [1 | a_bar | ...]: only its shape matters for the S block
    zero = np.zeros((k["k"], n), dtype=np.int64)
    sb = T.gadget.gen_short_basis_for_trapdoor_ring(gp, a, zero, zero)          # K x nK x n
    expect = _polys(k["expect"], 1)[:, :, 0]                       # k x k constants
    if k["base"] ** k["k"] == k["q"]:
        expect = expect[:, ::-1]                                   # gen_sa_r reverses the columns of S when q is a power of the base (short_basis_ring.rs:109-112)
    kk = k["k"]
    # sa_r = [pd (x) [0; S] | pd (x) [I_2; w]] with pd = [X^0 | X^1 | ...] (:95-123): columns 0 .. k-1 are S itself, columns k .. 2k-1 are X S, ...
    assert (sb[:2, :n * kk, :] == 0).all()
    for t in range(n):
        blk = sb[2:, t * kk:(t + 1) * kk, :]
        assert (blk[:, :, t] == expect).all(), (name, t)
        assert (np.delete(blk, t, axis=2) == 0).all(), (name, t)


def test_ring_working_sa_r_through_the_product(kats):
    """gen_sa_r (short_basis_ring.rs:105-135) at n = 4, q = 16: r = e = 0 makes the product's basis equal sa_r; its coefficient embedding must equal the fixture."""
    import tools_amd as T
    kr = kats["short_basis_ring/working_sa_r"]
    n = kr["n"]
    gp = _ring_params(T, n, kr["q"])
    a = _polys([kr["a"]], n)[0].astype(np.uint64)
    zero = np.zeros((gp.k, n), dtype=np.int64)
    sb = T.gadget.gen_short_basis_for_trapdoor_ring(gp, a, zero, zero)          # K x nK x n
    emb = sb.transpose(0, 2, 1).reshape(sb.shape[0] * n, sb.shape[1])          # coefficient embedding, short_basis_ring.rs:443
    assert (emb == np.array(kr["expect_coefficient_embedding"])).all()


def test_ring_working_sa_l_through_the_product(kats):
    """gen_sa_l (short_basis_ring.rs:81-103) at n = 4, q = 16: the product returns sa_l * sa_r reduced by X^n + 1 (:72-77); it must equal the product of the two
    fixtures (sa_l from working_sa_l, sa_r from working_sa_r -- same a, r, e in both records)."""
    import tools_amd as T
    kl, kr = kats["short_basis_ring/working_sa_l"], kats["short_basis_ring/working_sa_r"]
    n = kl["n"]
    assert kl["a"] == kr["a"] and kl["q"] == kr["q"]
    gp = _ring_params(T, n, kl["q"])
    K, d = gp.k + 2, n * (gp.k + 2)
    a = _polys([kl["a"]], n)[0].astype(np.uint64)
    r = _polys([kl["r"]], n)[0]
    e = _polys([kl["e"]], n)[0]
    sal = _polys(kl["expect"], n)                                  # K x K x n.  The reference test calls gen_sa_l(&r, &e) with the arguments in the order of its
    emb = np.array(kr["expect_coefficient_embedding"])             # signature (e, r): the fixture's row 0 carries r, row 1 carries e -- so the product is fed (r := e, e := r)
    sar = emb.reshape(K, n, d).transpose(0, 2, 1)                  # K x d x n
    expect = np.zeros((K, d, n), dtype=np.int64)
    for row in range(K):
        for col in range(d):
            acc = [0] * n
            for t in range(K):
                if sal[row, t].any() and sar[t, col].any():
                    acc = [x + y for x, y in zip(acc, _negacyclic(sal[row, t], sar[t, col]))]
            expect[row, col] = acc
    got = T.gadget.gen_short_basis_for_trapdoor_ring(gp, a, e, r)  # (the library's (r, e) in the order of gen_short_basis_for_trapdoor_ring, :64-70)
    assert (got == expect).all()
