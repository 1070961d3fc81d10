"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (tools_amd) never does: it fails loudly when its HIP library is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpsf_oracle.so")

TAG_ABAR, TAG_R, TAG_NORMAL, TAG_PERTURB, TAG_GADGET, TAG_SAMPD, TAG_TARGET, TAG_GPV = 1, 2, 3, 4, 5, 6, 7, 8
TAG_RING_R, TAG_RING_E, TAG_RING_A = 9, 10, 11

OK, ERR_PARAM, ERR_NOT_PD, ERR_DOMAIN, ERR_MODULUS, ERR_NO_SOLUTION = range(6)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("psf_oracle.c", "psf_oracle_gpv.c", "psf_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libpsf_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class GadgetParams(C.Structure):
    _fields_ = [("n", C.c_uint64), ("k", C.c_uint64), ("m_bar", C.c_uint64), ("base", C.c_uint64), ("q", C.c_uint64)]

    def as_tuple(self):
        return (self.n, self.k, self.m_bar, self.base, self.q)


class _Psfp(C.Structure):
    _fields_ = [("gp", GadgetParams), ("r", C.c_double), ("s", C.c_double), ("m", C.c_size_t),
                ("A", C.POINTER(C.c_uint64)), ("R", C.POINTER(C.c_int8)), ("L", C.POINTER(C.c_double)),
                ("Sk", C.POINTER(C.c_int64)), ("Sk_gso", C.POINTER(C.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_det_exp.restype = C.c_double
        _lib.orc_det_exp.argtypes = [C.c_double]
        _lib.orc_sample_z.restype = C.c_int64
        _lib.orc_sample_z.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_double, C.c_double]
        _lib.orc_sample_normal.restype = C.c_double
        _lib.orc_sample_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        _lib.orc_uniform_mod.restype = C.c_uint64
        _lib.orc_uniform_mod.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
        _lib.orc_psfp_new.restype = C.POINTER(_Psfp)
        _lib.orc_psfp_new.argtypes = [C.POINTER(GadgetParams), C.c_double, C.c_double]
        _lib.orc_psfp_free.argtypes = [C.POINTER(_Psfp)]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def _check(rc, allow=()):
    if rc != OK and rc not in allow:
        raise RuntimeError(f"oracle status {rc}")
    return rc


# ---------------------------------------------------------------- randomness primitives
def philox(seed, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(C.c_uint64(seed), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), out)
    return list(out)


def det_exp(y):
    return lib().orc_det_exp(y)


def sample_z(seed, tag, index, coord, center, s):
    return lib().orc_sample_z(seed, tag, index, coord, center, s)


def sample_normal(seed, index, coord):
    return lib().orc_sample_normal(seed, index, coord)


def uniform_targets(seed, B, n, q, first_index=0):
    """Synthetic uniform syndromes u in Z_q^{B x n} (benches/psf.rs:35,60,87), stream TAG_TARGET."""
    u = np.empty((B, n), dtype=np.uint64)
    L = lib()
    for b in range(B):
        for i in range(n):
            u[b, i] = L.orc_uniform_mod(seed, TAG_TARGET, i, (first_index + b) & 0xFFFFFFFF, q)
    return u


# ---------------------------------------------------------------- deterministic helpers
def gadget_params_default(n, q):
    gp = GadgetParams()
    _check(lib().orc_gadget_params_default(C.c_uint64(n), C.c_uint64(q), C.byref(gp)))
    return gp


def gadget_params_ring_default(n, q):
    gp = GadgetParams()
    _check(lib().orc_gadget_params_ring_default(C.c_uint64(n), C.c_uint64(q), C.byref(gp)))
    return gp


def gen_gadget_vec(k, base):
    out = np.zeros(k, dtype=np.int64)
    _check(lib().orc_gen_gadget_vec(C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)))
    return out


def gen_gadget_mat(n, k, base):
    out = np.zeros((n, n * k), dtype=np.int64)
    _check(lib().orc_gen_gadget_mat(C.c_uint64(n), C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)))
    return out


def find_solution_gadget_vec(value, q, k, base):
    out = np.zeros(k, dtype=np.int64)
    rc = lib().orc_find_solution_gadget_vec(C.c_uint64(value), C.c_uint64(q), C.c_uint64(k), C.c_uint64(base),
                                            _p(out, C.c_int64))
    _check(rc)
    return out


def find_solution_gadget_mat(value, q, k, base):
    value = _u64(value)
    rows, cols = value.shape
    out = np.zeros((k * rows, cols), dtype=np.int64)
    _check(lib().orc_find_solution_gadget_mat(_p(value, C.c_uint64), C.c_size_t(rows), C.c_size_t(cols),
                                              C.c_uint64(q), C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)))
    return out


def short_basis_gadget(gp):
    w = gp.n * gp.k
    out = np.zeros((w, w), dtype=np.int64)
    _check(lib().orc_short_basis_gadget(C.byref(gp), _p(out, C.c_int64)))
    return out


def short_basis_gadget_block(gp):
    out = np.zeros((gp.k, gp.k), dtype=np.int64)
    _check(lib().orc_short_basis_gadget_block(C.byref(gp), _p(out, C.c_int64)))
    return out


def gso_columns(basis):
    basis = _i64(basis)
    d = basis.shape[0]
    out = np.zeros((d, d), dtype=np.float64)
    lib().orc_gso_columns(_p(basis, C.c_int64), C.c_size_t(d), _p(out, C.c_double))
    return out


def sample_r(seed, m_bar, w):
    R = np.zeros((m_bar, w), dtype=np.int8)
    lib().orc_sample_r(C.c_uint64(seed), C.c_size_t(m_bar), C.c_size_t(w), _p(R, C.c_int8))
    return R


def sample_a_bar(seed, n, m_bar, q):
    a = np.zeros((n, m_bar), dtype=np.uint64)
    lib().orc_sample_a_bar(C.c_uint64(seed), C.c_size_t(n), C.c_size_t(m_bar), C.c_uint64(q), _p(a, C.c_uint64))
    return a


def gen_trapdoor(gp, a_bar, R, tag=None):
    a_bar, R = _u64(a_bar), _i8(R)
    m = gp.m_bar + gp.n * gp.k
    A = np.zeros((gp.n, m), dtype=np.uint64)
    tagp = _p(_u64(tag), C.c_uint64) if tag is not None else None
    _check(lib().orc_gen_trapdoor(C.byref(gp), _p(a_bar, C.c_uint64), tagp, _p(R, C.c_int8), _p(A, C.c_uint64)))
    return A


def gen_sa_l(R):
    R = _i8(R)
    mb, w = R.shape
    out = np.zeros((mb + w, mb + w), dtype=np.int64)
    _check(lib().orc_gen_sa_l(_p(R, C.c_int8), C.c_size_t(mb), C.c_size_t(w), _p(out, C.c_int64)))
    return out


def gen_sa_r(gp, A, tag=None):
    A = _u64(A)
    m = gp.m_bar + gp.n * gp.k
    out = np.zeros((m, m), dtype=np.int64)
    tagp = _p(_u64(tag), C.c_uint64) if tag is not None else None
    _check(lib().orc_gen_sa_r(C.byref(gp), tagp, _p(A, C.c_uint64), _p(out, C.c_int64)))
    return out


def compute_w(gp, A, tag=None):
    A = _u64(A)
    out = np.zeros((gp.n * gp.k, gp.m_bar), dtype=np.int64)
    tagp = _p(_u64(tag), C.c_uint64) if tag is not None else None
    _check(lib().orc_compute_w(C.byref(gp), tagp, _p(A, C.c_uint64), _p(out, C.c_int64)))
    return out


def gen_short_basis_for_trapdoor(gp, A, R, tag=None):
    A, R = _u64(A), _i8(R)
    m = gp.m_bar + gp.n * gp.k
    out = np.zeros((m, m), dtype=np.int64)
    tagp = _p(_u64(tag), C.c_uint64) if tag is not None else None
    _check(lib().orc_gen_short_basis_for_trapdoor(C.byref(gp), tagp, _p(A, C.c_uint64), _p(R, C.c_int8),
                                                  _p(out, C.c_int64)))
    return out


# ---------------------------------------------------------------- PSFPerturbation
class PSFPerturbation:
    """Oracle mirror of mp_perturbation.rs:57-62 / :193-403."""

    def __init__(self, gp, r, s, with_L=True):
        self.gp, self.r, self.s = gp, float(r), float(s)
        L = lib()
        L.orc_psfp_new_nokey.restype = C.POINTER(_Psfp)
        L.orc_psfp_new_nokey.argtypes = [C.POINTER(GadgetParams), C.c_double, C.c_double]
        self._h = (L.orc_psfp_new if with_L else L.orc_psfp_new_nokey)(C.byref(gp), C.c_double(r), C.c_double(s))
        if not self._h:
            raise ValueError("bad parameters")
        self.with_L = with_L
        self.n, self.k, self.m_bar = gp.n, gp.k, gp.m_bar
        self.w = gp.n * gp.k
        self.m = self.m_bar + self.w

    def __del__(self):
        try:                                   # at interpreter shutdown module globals may already be gone
            if getattr(self, "_h", None):
                lib().orc_psfp_free(self._h)
                self._h = None
        except Exception:
            pass

    # key material views (copies)
    @property
    def A(self):
        return np.ctypeslib.as_array(self._h.contents.A, shape=(self.n, self.m)).copy()

    @property
    def R(self):
        return np.ctypeslib.as_array(self._h.contents.R, shape=(self.m_bar, self.w)).copy()

    @property
    def L_packed(self):
        return np.ctypeslib.as_array(self._h.contents.L, shape=(self.m * (self.m + 1) // 2,)).copy()

    @property
    def Sk(self):
        return np.ctypeslib.as_array(self._h.contents.Sk, shape=(self.k, self.k)).copy()

    @property
    def Sk_gso(self):
        return np.ctypeslib.as_array(self._h.contents.Sk_gso, shape=(self.k, self.k)).copy()

    def L_dense(self):
        Ld = np.zeros((self.m, self.m))
        Ld[np.tril_indices(self.m)] = self.L_packed
        return Ld

    def trap_gen(self, seed):
        return lib().orc_psfp_trap_gen(self._h, C.c_uint64(seed))

    def load_key(self, A, R, L_packed=None):
        A, R = _u64(A), _i8(R)
        assert A.shape == (self.n, self.m) and R.shape == (self.m_bar, self.w)
        if L_packed is None:
            assert not self.with_L, "an oracle object created with storage for sqrt(Sigma_2) needs the factor"
            _check(lib().orc_psfp_load_key(self._h, _p(A, C.c_uint64), _p(R, C.c_int8), None))
            return
        Lp = np.ascontiguousarray(L_packed, dtype=np.float64)
        assert Lp.size == self.m * (self.m + 1) // 2
        _check(lib().orc_psfp_load_key(self._h, _p(A, C.c_uint64), _p(R, C.c_int8), _p(Lp, C.c_double)))

    def sqrt_sigma_2_leading(self, R, s_cov, m0):
        """Rows 0..m0-1 of the Cholesky factor of Sigma_2 (unblocked recurrence), packed."""
        R = _i8(R)
        Lp = np.zeros(m0 * (m0 + 1) // 2)
        rc = lib().orc_psfp_sqrt_sigma_2_leading(self._h, _p(R, C.c_int8), C.c_double(s_cov), C.c_size_t(m0), _p(Lp, C.c_double))
        return rc, Lp

    def structured_sqrt(self, R, s_cov):
        """L_1 of the structured square root of Sigma_2 (m_bar(m_bar+1)/2 doubles, packed rows)."""
        R = _i8(R)
        Lp = np.zeros(self.m_bar * (self.m_bar + 1) // 2)
        rc = lib().orc_psfp_structured_sqrt(self._h, _p(R, C.c_int8), C.c_double(s_cov), _p(Lp, C.c_double))
        return rc, Lp

    def samp_p_structured_trace(self, L1_packed, s_cov, seed, index, u):
        """One preimage through the structured square root (A, R from load_key); all stages."""
        u = _u64(u).reshape(self.n)
        L1 = np.ascontiguousarray(L1_packed, dtype=np.float64)
        d, x = np.zeros(self.m), np.zeros(self.m)
        p, e = np.zeros(self.m, dtype=np.int64), np.zeros(self.m, dtype=np.int64)
        v, z = np.zeros(self.n, dtype=np.uint64), np.zeros(self.w, dtype=np.int64)
        _check(lib().orc_psfp_samp_p_structured_trace(self._h, _p(L1, C.c_double), C.c_double(s_cov), C.c_uint64(seed), C.c_uint64(index), _p(u, C.c_uint64),
                                                      _p(d, C.c_double), _p(x, C.c_double), _p(p, C.c_int64), _p(v, C.c_uint64), _p(z, C.c_int64), _p(e, C.c_int64)))
        return dict(d=d, x=x, p=p, v=v, z=z, e=e)

    def samp_p_from_x(self, seed, index, u, x):
        """p, v, z, e of one preimage from its centres x (every stage after x = sqrt(Sigma_2) d)."""
        u = _u64(u).reshape(self.n)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.m)
        p, e = np.zeros(self.m, dtype=np.int64), np.zeros(self.m, dtype=np.int64)
        v, z = np.zeros(self.n, dtype=np.uint64), np.zeros(self.w, dtype=np.int64)
        _check(lib().orc_psfp_samp_p_from_x(self._h, C.c_uint64(seed), C.c_uint64(index), _p(u, C.c_uint64), _p(x, C.c_double),
                                            _p(p, C.c_int64), _p(v, C.c_uint64), _p(z, C.c_int64), _p(e, C.c_int64)))
        return dict(p=p, v=v, z=z, e=e)

    def compute_sqrt_sigma_2(self, R, s_cov):
        R = _i8(R)
        Lp = np.zeros(self.m * (self.m + 1) // 2)
        rc = lib().orc_psfp_compute_sqrt_sigma_2(self._h, _p(R, C.c_int8), C.c_double(s_cov), _p(Lp, C.c_double))
        return rc, Lp

    def compute_sqrt_sigma_2_dense(self, R, sigma_packed):
        """mp_perturbation.rs:111-139 for a general covariance: sigma_packed = lower triangle of mat_sigma, row i holding i + 1 entries"""
        R = _i8(R)
        sg = np.ascontiguousarray(sigma_packed, dtype=np.float64)
        assert sg.size == self.m * (self.m + 1) // 2
        Lp = np.zeros(self.m * (self.m + 1) // 2)
        rc = lib().orc_psfp_compute_sqrt_sigma_2_dense(self._h, _p(R, C.c_int8), _p(sg, C.c_double), _p(Lp, C.c_double))
        return rc, Lp

    def samp_p(self, seed, u, first_index=0, nthreads=0):
        u = _u64(u).reshape(-1, self.n)
        B = u.shape[0]
        e = np.zeros((B, self.m), dtype=np.int64)
        _check(lib().orc_psfp_samp_p(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B),
                                     _p(u, C.c_uint64), _p(e, C.c_int64), C.c_int(nthreads)))
        return e

    def samp_p_trace(self, seed, index, u):
        u = _u64(u).reshape(self.n)
        d, x = np.zeros(self.m), np.zeros(self.m)
        p, e = np.zeros(self.m, dtype=np.int64), np.zeros(self.m, dtype=np.int64)
        v, z = np.zeros(self.n, dtype=np.uint64), np.zeros(self.w, dtype=np.int64)
        _check(lib().orc_psfp_samp_p_trace(self._h, C.c_uint64(seed), C.c_uint64(index), _p(u, C.c_uint64),
                                           _p(d, C.c_double), _p(x, C.c_double), _p(p, C.c_int64), _p(v, C.c_uint64),
                                           _p(z, C.c_int64), _p(e, C.c_int64)))
        return dict(d=d, x=x, p=p, v=v, z=z, e=e)

    def samp_d(self, seed, B=1, first_index=0):
        e = np.zeros((B, self.m), dtype=np.int64)
        _check(lib().orc_psfp_samp_d(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(e, C.c_int64)))
        return e

    def f_a(self, e, allow_domain_error=False):
        e = _i64(e).reshape(-1, self.m)
        B = e.shape[0]
        u = np.zeros((B, self.n), dtype=np.uint64)
        rc = lib().orc_psfp_f_a(self._h, C.c_size_t(B), _p(e, C.c_int64), _p(u, C.c_uint64))
        if rc == ERR_DOMAIN and not allow_domain_error:
            raise AssertionError("sigma not in domain (mp_perturbation.rs:367)")
        return u

    def check_domain(self, e):
        e = _i64(e)
        if e.ndim == 1:
            e = e.reshape(1, -1)
        B, ln = e.shape
        ok = np.zeros(B, dtype=np.uint8)
        _check(lib().orc_psfp_check_domain(self._h, C.c_size_t(B), _p(e, C.c_int64), C.c_size_t(ln), _p(ok, C.c_uint8)))
        return ok.astype(bool)

    def gadget_sample(self, seed, index, v):
        v = _u64(v).reshape(self.n)
        z = np.zeros(self.w, dtype=np.int64)
        _check(lib().orc_randomized_nearest_plane_gadget(self._h, C.c_uint64(seed), C.c_uint64(index),
                                                         _p(v, C.c_uint64), _p(z, C.c_int64)))
        return z


def normals(seed, index, m):
    """d <- N(0,1)^m of preimage `index` (stream TAG_NORMAL)."""
    L = lib()
    return np.array([L.orc_sample_normal(seed, index, j) for j in range(m)])


def centres_rows(L_rows, row0, nrows, m, d):
    """x[b][r] = sum_{j <= row0 + r} L[row0 + r][j] d[b][j] for a packed row block of sqrt(Sigma_2) (ascending fma chain)."""
    L_rows = np.ascontiguousarray(L_rows, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64).reshape(-1, m)
    first = row0 * (row0 + 1) // 2
    last = (row0 + nrows) * (row0 + nrows + 1) // 2
    assert L_rows.size == last - first
    x = np.zeros((d.shape[0], nrows))
    _check(lib().orc_psfp_centres_rows(_p(L_rows, C.c_double), C.c_size_t(row0), C.c_size_t(nrows), C.c_size_t(m),
                                       _p(d, C.c_double), C.c_size_t(d.shape[0]), _p(x, C.c_double)))
    return x


# ---------------------------------------------------------------- PSFGPV
def solve_gaussian_elimination(A, q, u):
    A = _u64(A)
    n, m = A.shape
    u = _u64(u).reshape(n)
    sol = np.zeros(m, dtype=np.uint64)
    rc = lib().orc_solve_gaussian_elimination(_p(A, C.c_uint64), C.c_size_t(n), C.c_size_t(m), C.c_uint64(q),
                                              _p(u, C.c_uint64), _p(sol, C.c_uint64))
    return rc, sol


def rot_minus_matrix(mat):
    mat = _i64(mat)
    rows, cols = mat.shape
    out = np.zeros((rows, rows * cols), dtype=np.int64)
    lib().orc_rot_minus_matrix(_p(mat, C.c_int64), C.c_size_t(rows), C.c_size_t(cols), _p(out, C.c_int64))
    return out


class PSFGPV:
    """Oracle mirror of gpv.rs:53-57 / :59-225."""

    def __init__(self, gp, s):
        L = lib()
        L.orc_gpv_new.restype = C.c_void_p
        L.orc_gpv_new.argtypes = [C.POINTER(GadgetParams), C.c_double]
        for nm, rt in (("orc_gpv_A", C.POINTER(C.c_uint64)), ("orc_gpv_R", C.POINTER(C.c_int8)),
                       ("orc_gpv_basis_t", C.POINTER(C.c_int32)), ("orc_gpv_gso_t", C.POINTER(C.c_double))):
            getattr(L, nm).restype = rt
            getattr(L, nm).argtypes = [C.c_void_p]
        L.orc_gpv_free.argtypes = [C.c_void_p]
        self.gp, self.s = gp, float(s)
        self._h = C.c_void_p(L.orc_gpv_new(C.byref(gp), C.c_double(s)))
        if not self._h:
            raise ValueError("bad parameters")
        self.n, self.k, self.m_bar = gp.n, gp.k, gp.m_bar
        self.w = gp.n * gp.k
        self.m = self.m_bar + self.w

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_gpv_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def A(self):
        return np.ctypeslib.as_array(lib().orc_gpv_A(self._h), shape=(self.n, self.m)).copy()

    @property
    def R(self):
        return np.ctypeslib.as_array(lib().orc_gpv_R(self._h), shape=(self.m_bar, self.w)).copy()

    @property
    def basis_t(self):
        return np.ctypeslib.as_array(lib().orc_gpv_basis_t(self._h), shape=(self.m, self.m)).copy()

    @property
    def gso_t(self):
        return np.ctypeslib.as_array(lib().orc_gpv_gso_t(self._h), shape=(self.m, self.m)).copy()

    def trap_gen(self, seed):
        return lib().orc_gpv_trap_gen(self._h, C.c_uint64(seed))

    def load_key(self, A, basis_t, gso_t):
        A = _u64(A)
        bt = np.ascontiguousarray(basis_t, dtype=np.int32)
        gt = np.ascontiguousarray(gso_t, dtype=np.float64)
        assert A.shape == (self.n, self.m) and bt.shape == (self.m, self.m) and gt.shape == (self.m, self.m)
        return lib().orc_gpv_load_key(self._h, _p(A, C.c_uint64), _p(bt, C.c_int32), _p(gt, C.c_double))

    def samp_p(self, seed, u, first_index=0, percall=False, nthreads=0):
        u = _u64(u).reshape(-1, self.n)
        B = u.shape[0]
        e = np.zeros((B, self.m), dtype=np.int64)
        _check(lib().orc_gpv_samp_p(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u, C.c_uint64),
                                    _p(e, C.c_int64), C.c_int(1 if percall else 0), C.c_int(nthreads)))
        return e

    def set_two_pass(self, mode):
        """-1: by the rule q sqrt(n) > 2^13 s; 0 / 1: forced"""
        lib().orc_gpv_set_two_pass(self._h, C.c_int(mode))

    @property
    def two_pass(self):
        return bool(lib().orc_gpv_two_pass(self._h))

    def samp_p_trace(self, seed, u, index=0):
        """one preimage: (e, integer centre vector the final pass started from, centres c'_i as doubles, coefficients z_i)"""
        u = _u64(u).reshape(self.n)
        e = np.zeros(self.m, dtype=np.int64)
        c0 = np.zeros(self.m, dtype=np.int64)
        cen = np.zeros(self.m, dtype=np.float64)
        z = np.zeros(self.m, dtype=np.int64)
        _check(lib().orc_gpv_samp_p_trace(self._h, C.c_uint64(seed), C.c_uint64(index), _p(u, C.c_uint64), _p(e, C.c_int64),
                                          _p(c0, C.c_int64), _p(cen, C.c_double), _p(z, C.c_int64)))
        return e, c0, cen, z

    def samp_d(self, seed, B=1, first_index=0):
        e = np.zeros((B, self.m), dtype=np.int64)
        _check(lib().orc_gpv_samp_d(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(e, C.c_int64)))
        return e

    def f_a(self, e):
        e = _i64(e).reshape(-1, self.m)
        u = np.zeros((e.shape[0], self.n), dtype=np.uint64)
        rc = lib().orc_gpv_f_a(self._h, C.c_size_t(e.shape[0]), _p(e, C.c_int64), _p(u, C.c_uint64))
        if rc == ERR_DOMAIN:
            raise AssertionError("sigma not in domain (gpv.rs:191)")
        return u

    def check_domain(self, e):
        e = _i64(e)
        if e.ndim == 1:
            e = e.reshape(1, -1)
        ok = np.zeros(e.shape[0], dtype=np.uint8)
        _check(lib().orc_gpv_check_domain(self._h, C.c_size_t(e.shape[0]), _p(e, C.c_int64), C.c_size_t(e.shape[1]), _p(ok, C.c_uint8)))
        return ok.astype(bool)


# ---------------------------------------------------------------- ring variant
def gso_rows(basis_t):
    bt = np.ascontiguousarray(basis_t, dtype=np.int32)
    d = bt.shape[0]
    out = np.zeros((d, d), dtype=np.float64)
    lib().orc_gso_rows(_p(bt, C.c_int32), C.c_size_t(d), _p(out, C.c_double))
    return out


def gso_rows_leading(basis_rows):
    """Gram-Schmidt of the leading rows of a (transposed) basis: basis_rows is nrows x width, row i = basis vector i."""
    bt = np.ascontiguousarray(basis_rows, dtype=np.int32)
    nrows, width = bt.shape
    out = np.zeros((nrows, width), dtype=np.float64)
    lib().orc_gso_rows_leading(_p(bt, C.c_int32), C.c_size_t(nrows), C.c_size_t(width), _p(out, C.c_double))
    return out


def ring_trap_gen(gp, s_td, seed):
    n, k = gp.n, gp.k
    a = np.zeros((k + 2, n), dtype=np.uint64)
    r = np.zeros((k, n), dtype=np.int64)
    e = np.zeros((k, n), dtype=np.int64)
    _check(lib().orc_ring_trap_gen(C.byref(gp), C.c_double(s_td), C.c_uint64(seed), _p(a, C.c_uint64), _p(r, C.c_int64), _p(e, C.c_int64)))
    return a, r, e


def ring_compute_s(gp):
    out = np.zeros((gp.k, gp.k), dtype=np.int64)
    _check(lib().orc_ring_compute_s(C.byref(gp), _p(out, C.c_int64)))
    return out


def find_solution_gadget_ring(u, q, k, base):
    u = _u64(u)
    out = np.zeros((k, u.size), dtype=np.int64)
    _check(lib().orc_find_solution_gadget_ring(_p(u, C.c_uint64), C.c_size_t(u.size), C.c_uint64(q), C.c_uint64(k), C.c_uint64(base), _p(out, C.c_int64)))
    return out


def ring_gen_sa_l(first, second):
    first, second = _i64(first), _i64(second)
    k, n = first.shape
    out = np.zeros((k + 2, k + 2, n), dtype=np.int64)
    lib().orc_ring_gen_sa_l(_p(first, C.c_int64), _p(second, C.c_int64), C.c_size_t(n), C.c_size_t(k), _p(out, C.c_int64))
    return out


def ring_gen_sa_r(gp, a):
    a = _u64(a)
    K, n = gp.k + 2, gp.n
    out = np.zeros((K, n * K, n), dtype=np.int64)
    _check(lib().orc_ring_gen_sa_r(C.byref(gp), _p(a, C.c_uint64), _p(out, C.c_int64)))
    return out


def ring_short_basis_t(gp, a, r, e):
    a, r, e = _u64(a), _i64(r), _i64(e)
    d = gp.n * (gp.k + 2)
    out = np.zeros((d, d), dtype=np.int32)
    _check(lib().orc_ring_short_basis_t(C.byref(gp), _p(a, C.c_uint64), _p(r, C.c_int64), _p(e, C.c_int64), _p(out, C.c_int32)))
    return out


def ring_embed_a(a, q):
    a = _u64(a)
    K, n = a.shape
    out = np.zeros((n, n * K), dtype=np.uint64)
    lib().orc_ring_embed_a(_p(a, C.c_uint64), C.c_size_t(n), C.c_size_t(K), C.c_uint64(q), _p(out, C.c_uint64))
    return out


def poly_matrix_embedding(P):
    """coefficient embedding of a (rows x cols x n) polynomial matrix: (rows*n) x cols (short_basis_ring.rs:443)."""
    rows, cols, n = P.shape
    return P.transpose(0, 2, 1).reshape(rows * n, cols)


class PSFGPVRing:
    """Oracle mirror of gpv_ring.rs:62-67 / :69-284: ring key generation + the embedded PSFGPV machinery."""

    def __init__(self, gp, s, s_td):
        self.gp, self.s, self.s_td = gp, float(s), float(s_td)
        self.n, self.k = gp.n, gp.k
        self.K = gp.k + 2
        self.d = self.n * self.K
        egp = GadgetParams(gp.n, gp.k, 2 * gp.n, gp.base, gp.q)       # n x d system: d = 2n + nk
        self._gpv = PSFGPV(egp, s)
        self.a = self.r = self.e = None

    def trap_gen(self, seed):
        self.a, self.r, self.e = ring_trap_gen(self.gp, self.s_td, seed)     # gpv_ring.rs:91-98
        return self.load_key(self.a, self.r, self.e)

    def load_key(self, a, r, e, gso_t=None):
        self.a, self.r, self.e = _u64(a), _i64(r), _i64(e)
        self.basis_t = ring_short_basis_t(self.gp, self.a, self.r, self.e)   # gpv_ring.rs:169 (per call in the reference)
        self.A_emb = ring_embed_a(self.a, self.gp.q)                         # gpv_ring.rs:172-178
        self.gso_t = gso_rows(self.basis_t) if gso_t is None else gso_t
        return self._gpv.load_key(self.A_emb, self.basis_t, self.gso_t)

    def samp_p(self, seed, u, first_index=0, percall=False, nthreads=0):
        """u: B x n coefficient vectors of the syndromes; returns B x (k+2) x n."""
        e = self._gpv.samp_p(seed, _u64(u).reshape(-1, self.n), first_index=first_index, percall=percall, nthreads=nthreads)
        return e.reshape(-1, self.K, self.n)

    def samp_d(self, seed, B=1, first_index=0):
        return self._gpv.samp_d(seed, B=B, first_index=first_index).reshape(-1, self.K, self.n)

    def f_a(self, sigma):
        return self._gpv.f_a(_i64(sigma).reshape(-1, self.d))

    def check_domain(self, sigma):
        sigma = _i64(sigma)
        return self._gpv.check_domain(sigma.reshape(-1, self.d) if sigma.size % self.d == 0 and sigma.size else sigma.reshape(1, -1))


def num_threads():
    return lib().orc_num_threads()


# ---------------------------------------------------------------- faithful mode (GMP exact rationals; timing model of reference-style arithmetic)
_faith = None


def faithful_lib():
    """libpsf_faithful.so (oracle/psf_faithful_gmp.c), or None where GMP was not available at build time."""
    global _faith
    if _faith is None:
        path = os.path.join(_HERE, "libpsf_faithful.so")
        if not os.path.exists(path):
            subprocess.call(["make", "-C", _HERE, "faithful"], stdout=subprocess.DEVNULL)
        if not os.path.exists(path):
            return None
        lib()                                   # the oracle proper must be loaded first (the faithful library links against it)
        _faith = C.CDLL(path)
    return _faith


def faithful_psfp_samp_p(orc, seed, index, u):
    """One PSFPerturbation::samp_p call in mpq / mpz arithmetic (dense rational mat-vec, dense nk x nk GSO, [R; I] rebuilt)."""
    F = faithful_lib()
    u = _u64(u).reshape(orc.n)
    e = np.zeros(orc.m, dtype=np.int64)
    _check(F.orc_faithful_psfp_samp_p(orc._h, C.c_uint64(seed), C.c_uint64(index), _p(u, C.c_uint64), _p(e, C.c_int64)))
    return e


def faithful_gpv_samp_p(A, q, basis_t, gso_t, s, seed, index, u):
    """One PSFGPV::samp_p call: per-call elimination, nearest plane on the dense m x m rational Gram-Schmidt matrix."""
    F = faithful_lib()
    A, u = _u64(A), _u64(u).reshape(-1)
    n, m = A.shape
    bt = np.ascontiguousarray(basis_t, dtype=np.int32)
    gt = np.ascontiguousarray(gso_t, dtype=np.float64)
    e = np.zeros(m, dtype=np.int64)
    _check(F.orc_faithful_gpv_samp_p(_p(A, C.c_uint64), C.c_size_t(n), C.c_size_t(m), C.c_uint64(q), _p(bt, C.c_int32), _p(gt, C.c_double),
                                     C.c_double(s), C.c_uint64(seed), C.c_uint64(index), _p(u, C.c_uint64), _p(e, C.c_int64)))
    return e
