"""Gadget helpers of sample::g_trapdoor (gadget_classical.rs, short_basis_classical.rs, rotation_matrix.rs)
through the C ABI."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import GadgetParams, check, lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def gen_gadget_vec(k, base):
    """gadget_classical.rs:128-136"""
    out = np.zeros(k, dtype=np.int64)
    check(lib().psf_gen_gadget_vec(C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)), "gen_gadget_vec")
    return out


def gen_gadget_mat(n, k, base):
    """gadget_classical.rs:91-107"""
    out = np.zeros((n, n * k), dtype=np.int64)
    check(lib().psf_gen_gadget_mat(C.c_uint64(n), C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)), "gen_gadget_mat")
    return out


def find_solution_gadget_mat(value, q, k, base, device=0):
    """gadget_classical.rs:219-229 (HIP digit-decomposition kernel)."""
    value = np.ascontiguousarray(value, dtype=np.uint64)
    if value.ndim == 1:
        value = value.reshape(-1, 1)
    rows, cols = value.shape
    out = np.zeros((k * rows, cols), dtype=np.int64)
    check(lib().psf_find_solution_gadget_mat(C.c_int(device), _p(value, C.c_uint64), C.c_size_t(rows), C.c_size_t(cols),
                                             C.c_uint64(q), C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)),
          "find_solution_gadget_mat")
    return out


def find_solution_gadget_vec(value, q, k, base, device=0):
    """gadget_classical.rs:169-182"""
    return find_solution_gadget_mat(np.array([[value]], dtype=np.uint64), q, k, base, device).reshape(k)


def short_basis_gadget(gp):
    """gadget_classical.rs:248-287"""
    c = gp.c if hasattr(gp, "c") else gp
    w = c.n * c.k
    out = np.zeros((w, w), dtype=np.int64)
    check(lib().psf_short_basis_gadget(C.byref(c), _p(out, C.c_int64)), "short_basis_gadget")
    return out


def gen_short_basis_for_trapdoor(gp, A, R, tag=None):
    """short_basis_classical.rs:54-63"""
    c = gp.c if hasattr(gp, "c") else gp
    A = np.ascontiguousarray(A, dtype=np.uint64)
    R = np.ascontiguousarray(R, dtype=np.int8)
    m = c.m_bar + c.n * c.k
    out = np.zeros((m, m), dtype=np.int64)
    tagp = None
    if tag is not None:
        tag = np.ascontiguousarray(tag, dtype=np.uint64)
        tagp = _p(tag, C.c_uint64)
    check(lib().psf_gen_short_basis_for_trapdoor(C.byref(c), tagp, _p(A, C.c_uint64), _p(R, C.c_int8), _p(out, C.c_int64)),
          "gen_short_basis_for_trapdoor")
    return out


def rot_minus_matrix(mat):
    """rotation_matrix.rs:85-96"""
    mat = np.ascontiguousarray(mat, dtype=np.int64)
    rows, cols = mat.shape
    out = np.zeros((rows, rows * cols), dtype=np.int64)
    check(lib().psf_rot_minus_matrix(_p(mat, C.c_int64), C.c_size_t(rows), C.c_size_t(cols), _p(out, C.c_int64)), "rot_minus_matrix")
    return out


def rot_minus(vec):
    """rotation_matrix.rs:41-63"""
    vec = np.ascontiguousarray(vec, dtype=np.int64).reshape(-1, 1)
    return rot_minus_matrix(vec)


def poly_mul_negacyclic(a, b, q, device=0, method=None):
    """a * b in Z_q[X]/(X^n + 1) on the device (PolynomialRingZq product; gadget_ring.rs:78, gpv_ring.rs:245-246).
    a: residues (count x n or n), b: signed integers of the same shape.  method: None = automatic, 0 = schoolbook kernel,
    1 = (incomplete) negacyclic NTT kernel (PsfError(UNSUPPORTED) when q / n do not admit one)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    single = a.ndim == 1
    a2, b2 = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    assert a2.shape == b2.shape
    out = np.zeros_like(a2)
    if method is None:
        check(lib().psf_poly_mul_negacyclic(C.c_int(device), C.c_uint64(q), C.c_size_t(a2.shape[1]), C.c_size_t(a2.shape[0]),
                                            _p(a2, C.c_uint64), _p(b2, C.c_int64), _p(out, C.c_uint64)), "poly_mul_negacyclic")
    else:
        check(lib().psf_poly_mul_negacyclic_method(C.c_int(device), C.c_uint64(q), C.c_size_t(a2.shape[1]), C.c_size_t(a2.shape[0]),
                                                   _p(a2, C.c_uint64), _p(b2, C.c_int64), _p(out, C.c_uint64), C.c_int(method)),
              "poly_mul_negacyclic")
    return out[0] if single else out


def poly_mul_negacyclic_dev(d_a, d_b, d_out, q, n, count, io_bits=64, device=0, stream=None):
    """psf_poly_mul_negacyclic_dev: the same product on device buffers (raw pointers, e.g. torch `data_ptr()`), in `stream`, nothing allocated.
    io_bits 64: a uint64 / b int64 / out uint64; io_bits 16: a uint16 in [0, q) / b int16 in (-q, q) / out uint16 (NTT primes q < 2^14)."""
    check(lib().psf_poly_mul_negacyclic_dev(C.c_int(device), C.c_uint64(q), C.c_size_t(n), C.c_size_t(count), C.c_void_p(d_a), C.c_void_p(d_b),
                                            C.c_void_p(d_out), C.c_int(io_bits), C.c_void_p(stream or 0)), "poly_mul_negacyclic_dev")


def ntt_forward_dev(d_a, d_hat, q, n, count, io_bits=64, device=0, stream=None):
    """psf_ntt_forward_dev: transform `count` polynomials once; d_hat receives count * n 32-bit words (opaque images for poly_mul_hat_dev)."""
    check(lib().psf_ntt_forward_dev(C.c_int(device), C.c_uint64(q), C.c_size_t(n), C.c_size_t(count), C.c_void_p(d_a), C.c_int(io_bits),
                                    C.c_void_p(d_hat), C.c_void_p(stream or 0)), "ntt_forward_dev")


def poly_mul_hat_dev(d_hat, hat_stride, d_b, d_out, q, n, count, io_bits=64, device=0, stream=None):
    """psf_poly_mul_hat_dev: d_out[c] = image[c * hat_stride] * d_b[c]; hat_stride 0 = one image (a key polynomial) for every product."""
    check(lib().psf_poly_mul_hat_dev(C.c_int(device), C.c_uint64(q), C.c_size_t(n), C.c_size_t(count), C.c_void_p(d_hat), C.c_size_t(hat_stride),
                                     C.c_void_p(d_b), C.c_void_p(d_out), C.c_int(io_bits), C.c_void_p(stream or 0)), "poly_mul_hat_dev")


def gen_trapdoor(gp, a_bar, tag=None, seed=0, device=0):
    """gen_trapdoor (gadget_classical.rs:56-68): caller-supplied A_bar and tag H; returns (A, R), R <- PlusMinusOneZero from `seed`."""
    c = gp.c if hasattr(gp, "c") else gp
    a_bar = np.ascontiguousarray(a_bar, dtype=np.uint64)
    assert a_bar.shape == (c.n, c.m_bar)
    w = c.n * c.k
    A = np.zeros((c.n, c.m_bar + w), dtype=np.uint64)
    R = np.zeros((c.m_bar, w), dtype=np.int8)
    tagp = None
    if tag is not None:
        tag = np.ascontiguousarray(tag, dtype=np.uint64)
        assert tag.shape == (c.n, c.n)
        tagp = _p(tag, C.c_uint64)
    check(lib().psf_gen_trapdoor(C.c_int(device), C.byref(c), _p(a_bar, C.c_uint64), tagp, C.c_uint64(seed), _p(A, C.c_uint64), _p(R, C.c_int8)),
          "gen_trapdoor")
    return A, R


def gen_trapdoor_with_r(gp, a_bar, R, tag=None, device=0):
    """gen_trapdoor (gadget_classical.rs:56-68) with the caller's own R -- any TrapdoorDistribution's draw (:62-64); returns A."""
    c = gp.c if hasattr(gp, "c") else gp
    a_bar = np.ascontiguousarray(a_bar, dtype=np.uint64)
    w = c.n * c.k
    R = np.ascontiguousarray(R, dtype=np.int64)
    assert a_bar.shape == (c.n, c.m_bar) and R.shape == (c.m_bar, w)
    A = np.zeros((c.n, c.m_bar + w), dtype=np.uint64)
    tagp = None
    if tag is not None:
        tag = np.ascontiguousarray(tag, dtype=np.uint64)
        assert tag.shape == (c.n, c.n)
        tagp = _p(tag, C.c_uint64)
    check(lib().psf_gen_trapdoor_with_r(C.c_int(device), C.byref(c), _p(a_bar, C.c_uint64), tagp, _p(R, C.c_int64), _p(A, C.c_uint64)), "gen_trapdoor_with_r")
    return A


def gen_trapdoor_ring_lwe_with(gp, a_bar, r, e, device=0):
    """gen_trapdoor_ring_lwe (gadget_ring.rs:62-81) with the caller's own r, e (:69-70); returns a [(k+2) x n]."""
    c = gp.c if hasattr(gp, "c") else gp
    a_bar = np.ascontiguousarray(a_bar, dtype=np.uint64).reshape(c.n)
    r = np.ascontiguousarray(r, dtype=np.int64)
    e = np.ascontiguousarray(e, dtype=np.int64)
    assert r.shape == (c.k, c.n) and e.shape == (c.k, c.n)
    a = np.zeros((c.k + 2, c.n), dtype=np.uint64)
    check(lib().psf_gen_trapdoor_ring_lwe_with(C.c_int(device), C.byref(c), _p(a_bar, C.c_uint64), _p(r, C.c_int64), _p(e, C.c_int64), _p(a, C.c_uint64)),
          "gen_trapdoor_ring_lwe_with")
    return a


def gen_trapdoor_ring_lwe(gp, a_bar, s, seed=0, device=0):
    """gen_trapdoor_ring_lwe (gadget_ring.rs:62-81): returns (a [(k+2) x n], r [k x n], e [k x n])."""
    c = gp.c if hasattr(gp, "c") else gp
    a_bar = np.ascontiguousarray(a_bar, dtype=np.uint64).reshape(c.n)
    a = np.zeros((c.k + 2, c.n), dtype=np.uint64)
    r = np.zeros((c.k, c.n), dtype=np.int64)
    e = np.zeros((c.k, c.n), dtype=np.int64)
    check(lib().psf_gen_trapdoor_ring_lwe(C.c_int(device), C.byref(c), _p(a_bar, C.c_uint64), C.c_double(s), C.c_uint64(seed),
                                          _p(a, C.c_uint64), _p(r, C.c_int64), _p(e, C.c_int64)), "gen_trapdoor_ring_lwe")
    return a, r, e


def gen_gadget_ring(k, base):
    """gen_gadget_ring (gadget_ring.rs:103-109): constant terms of the k constant polynomials."""
    out = np.zeros(k, dtype=np.int64)
    check(lib().psf_gen_gadget_ring(C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)), "gen_gadget_ring")
    return out


def find_solution_gadget_ring(u, q, k, base, device=0):
    """find_solution_gadget_ring (gadget_ring.rs:145-166): u (n coefficients) -> k x n digit polynomials."""
    u = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1)
    out = np.zeros((k, u.size), dtype=np.int64)
    check(lib().psf_find_solution_gadget_ring(C.c_int(device), _p(u, C.c_uint64), C.c_size_t(u.size), C.c_uint64(q), C.c_uint64(k), C.c_uint64(base),
                                              _p(out, C.c_int64)), "find_solution_gadget_ring")
    return out


def gen_short_basis_for_trapdoor_ring(gp, a, r, e):
    """gen_short_basis_for_trapdoor_ring (short_basis_ring.rs:64-79): (k+2) x n(k+2) x n array of polynomial coefficients."""
    c = gp.c if hasattr(gp, "c") else gp
    a = np.ascontiguousarray(a, dtype=np.uint64)
    r = np.ascontiguousarray(r, dtype=np.int64)
    e = np.ascontiguousarray(e, dtype=np.int64)
    K = c.k + 2
    out = np.zeros((K, c.n * K, c.n), dtype=np.int64)
    check(lib().psf_gen_short_basis_for_trapdoor_ring(C.byref(c), _p(a, C.c_uint64), _p(r, C.c_int64), _p(e, C.c_int64), _p(out, C.c_int64)),
          "gen_short_basis_for_trapdoor_ring")
    return out


def shard_range(total, world, rank):
    """psf_shard_range: (first, count) of worker `rank`'s contiguous share of `total` rows."""
    first, count = C.c_size_t(0), C.c_size_t(0)
    check(lib().psf_shard_range(C.c_size_t(total), C.c_int(world), C.c_int(rank), C.byref(first), C.byref(count)), "shard_range")
    return first.value, count.value


def gso_rows(basis_t, device=0):
    """MatQ::gso (gpv.rs:88-91) on the device: Gram-Schmidt vectors of the rows of an integer matrix (rows x width)."""
    bt = np.ascontiguousarray(basis_t, dtype=np.int32)
    rows, width = bt.shape
    out = np.zeros((rows, width), dtype=np.float64)
    check(lib().psf_gso_rows(C.c_int(device), _p(bt, C.c_int32), C.c_size_t(rows), C.c_size_t(width), _p(out, C.c_double)), "gso_rows")
    return out
