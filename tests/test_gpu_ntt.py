"""R_q products on the device API (psf_poly_mul_negacyclic_dev / psf_ntt_forward_dev / psf_poly_mul_hat_dev; PolynomialRingZq multiplication under
gadget_ring.rs:78 and gpv_ring.rs:243-247): the wave-level NTT kernels against the exact schoolbook kernel and a big-integer product, in both I/O widths,
for every shape class (coefficients per lane 2 ... 16, leaf degree 1 / 2 / 4, 16-bit and 32-bit Montgomery forms, the generic LDS form)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tools_amd
    return tools_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def big_product(x, y, n, q):
    acc = [0] * n
    for i in range(n):
        xi = int(x[i])
        if xi == 0:
            continue
        for j in range(n):
            if i + j >= n:
                acc[i + j - n] -= xi * int(y[j])
            else:
                acc[i + j] += xi * int(y[j])
    return [v % q for v in acc]


# (q, n): wave kernels -- 3329 (7 levels: d = 1 / 2 / 4 at n = 128 / 256 / 512), complete transforms with 8 / 9 / 10 levels, d = 2 and 4 in the wider 16-bit
# form and in the 32-bit form, one register bit (n = 128, the leaf sign is a lane bit), 16 coefficients per lane
WAVE = [(3329, 128), (3329, 256), (3329, 512), (7681, 256), (12289, 512), (12289, 1024), (257, 128), (7937, 256), (7937, 512), (1153, 128), (1153, 256),
        (13313, 1024), (2013265921, 256), (1073479681, 512), (22273, 256), (20353, 256), (2013265921, 1024)]
# generic LDS form: n below 128 / above 1024, leaf degree above 4
GENERIC = [(17, 8), (5, 2), (257, 64), (3329, 1024), (12289, 2048), (13, 64), (2013265921, 32)]


def test_dev_products_equal_schoolbook_and_big_integers(T, torch):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    for q, n in WAVE + GENERIC:
        count = 37
        a = rng.integers(0, q, size=(count, n), dtype=np.uint64)
        b = rng.integers(-q + 1, q, size=(count, n), dtype=np.int64)
        a[0, :] = q - 1
        b[0, :] = np.where(np.arange(n) % 2 == 0, q - 1, -(q - 1))
        da, db = torch.from_numpy(a.view(np.int64)).to(dev), torch.from_numpy(b).to(dev)
        dout = torch.full((count, n), -1, dtype=torch.int64, device=dev)
        T.gadget.poly_mul_negacyclic_dev(da.data_ptr(), db.data_ptr(), dout.data_ptr(), q, n, count, 64)
        torch.cuda.synchronize()
        got = dout.cpu().numpy().view(np.uint64)
        want = T.gadget.poly_mul_negacyclic(a, b, q, method=0)
        assert (got == want).all(), (q, n)
        assert got[0].tolist() == big_product(a[0], b[0], n, q) and got[5].tolist() == big_product(a[5], b[5], n, q), (q, n)
        assert (T.gadget.poly_mul_negacyclic(a, b, q, method=1) == want).all()
        if q < 2**14 and (q, n) in WAVE:                                  # the same products through the 16-bit layout
            a16 = torch.from_numpy(a.astype(np.uint16).view(np.int16)).to(dev)
            b16 = torch.from_numpy(b.astype(np.int16)).to(dev)
            o16 = torch.full((count, n), -1, dtype=torch.int16, device=dev)
            T.gadget.poly_mul_negacyclic_dev(a16.data_ptr(), b16.data_ptr(), o16.data_ptr(), q, n, count, 16)
            torch.cuda.synchronize()
            assert (o16.cpu().numpy().view(np.uint16).astype(np.uint64) == want).all(), (q, n, 16)


def test_unreduced_64_bit_operands(T, torch):
    """the 64-bit layout takes any value: residues above q, negative multiples, entries beyond 32 bits (the one division is at the load)"""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    for q, n in [(3329, 256), (12289, 512), (2013265921, 256), (257, 64)]:
        count = 9
        a = rng.integers(0, 2**62, size=(count, n), dtype=np.uint64)
        b = rng.integers(-2**62, 2**62, size=(count, n), dtype=np.int64)
        a[1] = rng.integers(0, 2**31, size=n, dtype=np.uint64)          # the fast 32-bit input reduction
        b[1] = rng.integers(-2**31, 2**31, size=n, dtype=np.int64)
        b[2, 7] = -2**63
        da, db = torch.from_numpy(a.view(np.int64)).to(dev), torch.from_numpy(b).to(dev)
        dout = torch.empty((count, n), dtype=torch.int64, device=dev)
        T.gadget.poly_mul_negacyclic_dev(da.data_ptr(), db.data_ptr(), dout.data_ptr(), q, n, count, 64)
        torch.cuda.synchronize()
        got = dout.cpu().numpy().view(np.uint64)
        for c in (0, 1, 2):
            assert got[c].tolist() == big_product(a[c] % np.uint64(q), b[c].astype(object), n, q), (q, n, c)
        assert (got == T.gadget.poly_mul_negacyclic(a, b, q, method=0)).all()


def test_a_key_polynomial_is_transformed_once(T, torch):
    """psf_ntt_forward_dev + psf_poly_mul_hat_dev == psf_poly_mul_negacyclic_dev, with one image for all products and with one image each"""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(13)
    for q, n in [(3329, 256), (7681, 256), (12289, 1024), (2013265921, 256), (20353, 256)]:
        count = 21
        a = rng.integers(0, q, size=(count, n), dtype=np.uint64)
        b = rng.integers(-q + 1, q, size=(count, n), dtype=np.int64)
        da, db = torch.from_numpy(a.view(np.int64)).to(dev), torch.from_numpy(b).to(dev)
        hat = torch.empty((count, n), dtype=torch.int32, device=dev)
        out1 = torch.empty((count, n), dtype=torch.int64, device=dev)
        out2 = torch.empty((count, n), dtype=torch.int64, device=dev)
        T.gadget.ntt_forward_dev(da.data_ptr(), hat.data_ptr(), q, n, count, 64)
        T.gadget.poly_mul_hat_dev(hat.data_ptr(), n, db.data_ptr(), out1.data_ptr(), q, n, count, 64)
        T.gadget.poly_mul_hat_dev(hat.data_ptr(), 0, db.data_ptr(), out2.data_ptr(), q, n, count, 64)
        torch.cuda.synchronize()
        want = T.gadget.poly_mul_negacyclic(a, b, q, method=0)
        assert (out1.cpu().numpy().view(np.uint64) == want).all(), (q, n)
        a_rep = np.repeat(a[:1], count, axis=0)
        assert (out2.cpu().numpy().view(np.uint64) == T.gadget.poly_mul_negacyclic(a_rep, b, q, method=0)).all(), (q, n)


def test_every_modulus_has_a_device_product_and_unsupported_shapes_say_so(T, torch):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(14)
    for q, n in [(16, 8), (2**40 + 15, 64), (3329 * 3, 8), (2**31 + 11, 16), (3329, 6)]:      # no NTT: the schoolbook kernel behind the same entry point
        a = rng.integers(0, q, size=(3, n), dtype=np.uint64)
        b = rng.integers(-50, 50, size=(3, n), dtype=np.int64)
        da, db = torch.from_numpy(a.view(np.int64)).to(dev), torch.from_numpy(b).to(dev)
        dout = torch.empty((3, n), dtype=torch.int64, device=dev)
        T.gadget.poly_mul_negacyclic_dev(da.data_ptr(), db.data_ptr(), dout.data_ptr(), q, n, 3, 64)
        torch.cuda.synchronize()
        got = dout.cpu().numpy().view(np.uint64)
        for c in range(3):
            assert got[c].tolist() == big_product(a[c], b[c], n, q)
    buf = torch.zeros(4096, dtype=torch.int64, device=dev)
    for q, n, bits in [(16, 8, 16), (2013265921, 256, 16), (3329, 64, 16)]:                   # 16-bit layout: NTT primes below 2^14 with a wave kernel only
        with pytest.raises(T.PsfError) as ei:
            T.gadget.poly_mul_negacyclic_dev(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), q, n, 1, bits)
        assert ei.value.status == 8
    for q, n in [(16, 8), (3329, 64), (3329, 2048)]:                                          # images exist for wave shapes only
        with pytest.raises(T.PsfError) as ei:
            T.gadget.ntt_forward_dev(buf.data_ptr(), buf.data_ptr(), q, n, 1, 64)
        assert ei.value.status == 8
    T.gadget.poly_mul_negacyclic_dev(0, 0, 0, 3329, 256, 0, 64)                                # empty batch


def test_ring_f_a_by_ntt_equals_the_embedded_matrix_product(T, torch):
    """PSFGPVRing::f_a (gpv_ring.rs:243-247): k+2 R_q products against the cached images of a == rot^-(iota(a)) iota(sigma) on the matrix cores."""
    import math
    import os
    import subprocess
    import sys
    code = r'''
import sys, math, numpy as np
sys.path.insert(0, %r)
import tools_amd as T
n, q = int(sys.argv[1]), int(sys.argv[2])
s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
psf.trap_gen(4)
sg = psf.samp_d(seed=5, B=33)
u = psf.f_a(sg)
np.save(sys.argv[3], u)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    for n, q in [(128, 3329), (256, 3329), (256, 7681)]:
        outs = []
        for mode in ("ntt", "matmul"):
            with tempfile.NamedTemporaryFile(suffix=".npy") as f:
                from tests.conftest import exp_env
                env = exp_env(PSF_RING_FA=mode) if mode == "matmul" else exp_env()      # "ntt" is the release library's default route; "matmul" the experiments build's switch
                subprocess.check_call([sys.executable, "-c", code, str(n), str(q), f.name], env=env, timeout=900)
                outs.append(np.load(f.name))
        assert (outs[0] == outs[1]).all(), (n, q)
        assert outs[0].any()


@pytest.mark.parametrize("q,n", [(65537, 4096), (40961, 4096), (65537, 8192), (12289, 4096)])
def test_large_rings_take_the_ntt_route(T, q, n):
    """n = 4096 with a fully splitting prime and n = 8192: the generic LDS transform needs ((2 << L) + 3 n) words -- 96 to 160 KiB, above the 64 KiB a kernel gets
    by default; gfx950 has 160 KiB per workgroup and the library raises the kernel's limit (these sizes fell back to the O(n^2) kernel, and method = 1
    was refused although a plan existed).  NTT route == schoolbook kernel, explicitly (method 1) and by default."""
    rng = np.random.default_rng(n + q)
    count = 5
    a = rng.integers(0, q, size=(count, n), dtype=np.uint64)
    b = rng.integers(-q + 1, q, size=(count, n), dtype=np.int64)
    a[0, :] = q - 1
    b[0, :] = np.where(np.arange(n) % 2 == 0, q - 1, -(q - 1))
    want = T.gadget.poly_mul_negacyclic(a, b, q, method=0)
    assert (T.gadget.poly_mul_negacyclic(a, b, q, method=1) == want).all()
    assert (T.gadget.poly_mul_negacyclic(a, b, q) == want).all()
    # one coefficient of one product against exact integers
    c = 7
    exact = sum((int(a[1, i]) * int(b[1, (c - i) % n]) * (1 if i <= c else -1)) for i in range(n)) % q
    assert int(want[1, c]) == exact
