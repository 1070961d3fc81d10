"""PSF trait mirror (src/primitive/psf.rs:39-81) over the C ABI.

Batches: every method accepts one vector (reference semantics: one call = one preimage) or a 2-D array
with one row per call.  `seed` / `first_index` select the Philox streams (the reference has no seed).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import GadgetParams, GpvParams, PsfpParams, RingParams, PsfError, check, lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class GadgetParameters:
    """gadget_parameters.rs:44-52.  distribution = PlusMinusOneZero (trapdoor_distribution.rs:52-53)."""

    def __init__(self, n, k, m_bar, base, q):
        self.c = GadgetParams(n, k, m_bar, base, q)

    @classmethod
    def init_default(cls, n, q):
        """GadgetParameters::init_default (gadget_parameters.rs:113-133)."""
        c = GadgetParams()
        check(lib().psf_gadget_params_default(C.c_uint64(n), C.c_uint64(q), C.byref(c)), "init_default")
        return cls(c.n, c.k, c.m_bar, c.base, c.q)

    n = property(lambda self: self.c.n)
    k = property(lambda self: self.c.k)
    m_bar = property(lambda self: self.c.m_bar)
    base = property(lambda self: self.c.base)
    q = property(lambda self: self.c.q)

    def __repr__(self):
        return f"GadgetParameters(n={self.n}, k={self.k}, m_bar={self.m_bar}, base={self.base}, q={self.q})"


class GadgetParametersRing(GadgetParameters):
    """gadget_parameters.rs:73-81; modulus polynomial X^n + 1 (common_moduli.rs:41-48), distribution SampleZ."""

    @classmethod
    def init_default(cls, n, q):
        """GadgetParametersRing::init_default (gadget_parameters.rs:165-185)."""
        c = GadgetParams()
        check(lib().psf_gadget_params_ring_default(C.c_uint64(n), C.c_uint64(q), C.byref(c)), "init_default")
        return cls(c.n, c.k, c.m_bar, c.base, c.q)


class PSFPerturbation:
    """mp_perturbation.rs:57-62 / impl PSF :193-403 on one MI355X."""

    STRUCTURED_SQRT = 2      # PSFP_FLAG_STRUCTURED_SQRT

    def __init__(self, gp, r, s, device=0, structured=False):
        self.gp, self.r, self.s, self.device = gp, float(r), float(s), device
        self.structured = bool(structured)
        prm = PsfpParams(gp.c, self.r, self.s, device, self.STRUCTURED_SQRT if structured else 0)
        h = C.c_void_p()
        check(lib().psfp_create(C.byref(prm), C.byref(h)), "PSFPerturbation")
        self._h = h
        self._destroy = lib().psfp_destroy        # kept so that close() works during interpreter shutdown
        self.n, self.k, self.m_bar = gp.n, gp.k, gp.m_bar
        self.w = gp.n * gp.k
        self.m = self.m_bar + self.w

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    # ---- PSF trait -------------------------------------------------------------------------------
    def trap_gen(self, seed=0, export=True):
        """mp_perturbation.rs:221-244.  Returns (A, (R, sqrt_sigma_2_packed, (S_k, S_k_gso))); with export=False the key
        stays on the device (the factor is 3.8 GB at n=512, 60.5 GB at n=1024)."""
        check(lib().psfp_trap_gen(self._h, C.c_uint64(seed)), "trap_gen")
        return self.export_key() if export else None

    def samp_d(self, seed=0, B=None, first_index=0):
        """mp_perturbation.rs:264-267"""
        nb = 1 if B is None else B
        e = np.zeros((nb, self.m), dtype=np.int64)
        check(lib().psfp_samp_d(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(nb), _p(e, C.c_int64)), "samp_d")
        return e[0] if B is None else e

    def samp_p(self, u, seed=0, first_index=0, out=None):
        """mp_perturbation.rs:304-336 with the key installed in the handle (trap_gen / load_key).
        out: optional (B, m) int64 C-contiguous array to fill -- a reused buffer avoids first-touch page faults on a gigabyte
        of fresh memory (C3 batch: 80 ms per call into a reused buffer, 150-200 ms into a new one; tools/host_path_timing.py)."""
        u = np.ascontiguousarray(u, dtype=np.uint64)
        single = u.ndim == 1
        u2 = u.reshape(-1, self.n)
        B = u2.shape[0]
        if out is None:
            e = np.empty((B, self.m), dtype=np.int64)
        else:
            e = out
            assert e.dtype == np.int64 and e.shape == (B, self.m) and e.flags.c_contiguous
        check(lib().psfp_samp_p(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64),
                                _p(e, C.c_int64)), "samp_p")
        return e[0] if single else e

    def samp_p_async(self, u, out, seed=0, first_index=0):
        """psfp_samp_p_async: enqueue B samp_p calls on host buffers and return; `out` ((B, m) int64, C-contiguous) is complete after wait().
        At most two calls are in flight per handle; `out` must stay alive (and untouched) until wait() -- the wrapper keeps a reference."""
        u2 = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, self.n)
        B = u2.shape[0]
        assert out.dtype == np.int64 and out.shape == (B, self.m) and out.flags.c_contiguous
        check(lib().psfp_samp_p_async(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64), _p(out, C.c_int64)),
              "samp_p_async")
        self._inflight = getattr(self, "_inflight", []) + [out]
        return out

    def wait(self):
        """psfp_wait: every asynchronous call of this handle has completed; raises PsfError with the first failure (oldest call first)."""
        try:
            check(lib().psfp_wait(self._h), "wait")
        finally:
            self._inflight = []

    def next_ticket(self):
        """the ticket the next asynchronous call of this handle will carry (psfp_async_next_ticket)"""
        f = lib().psfp_async_next_ticket
        f.restype = C.c_uint64
        return int(f(self._h))

    def wait_ticket(self, ticket):
        """waits for the asynchronous call with this ticket (and the older one in flight) and raises if THAT call failed (psfp_wait_ticket)"""
        check(lib().psfp_wait_ticket(self._h, C.c_uint64(ticket)), "wait_ticket")

    def f_a(self, sigma):
        """mp_perturbation.rs:366-369; raises PsfError(ERR_DOMAIN) where the reference's assert! panics."""
        sigma = np.ascontiguousarray(sigma, dtype=np.int64)
        single = sigma.ndim == 1
        if sigma.ndim > 2 or sigma.shape[-1] != self.m:
            raise PsfError(_ffi.ERR_DOMAIN, "f_a")   # not a column vector of length m (:398-399)
        e2 = sigma.reshape(-1, self.m)
        B = e2.shape[0]
        u = np.zeros((B, self.n), dtype=np.uint64)
        check(lib().psfp_f_a(self._h, C.c_size_t(B), _p(e2, C.c_int64), _p(u, C.c_uint64)), "f_a")
        return u[0] if single else u

    def check_domain(self, sigma):
        """mp_perturbation.rs:396-402"""
        sigma = np.ascontiguousarray(sigma, dtype=np.int64)
        single = sigma.ndim == 1
        e2 = sigma.reshape(1, -1) if single else sigma
        B, ln = e2.shape
        ok = np.zeros(B, dtype=np.uint8)
        check(lib().psfp_check_domain(self._h, C.c_size_t(B), _p(e2, C.c_int64), C.c_size_t(ln), _p(ok, C.c_uint8)), "check_domain")
        return bool(ok[0]) if single else ok.astype(bool)

    # ---- key material ------------------------------------------------------------------------------
    def export_key(self):
        A = np.zeros((self.n, self.m), dtype=np.uint64)
        R = np.zeros((self.m_bar, self.w), dtype=np.int8)
        mL = self.m_bar if self.structured else self.m      # structured mode stores L_1 (m_bar x m_bar)
        Lp = np.zeros(mL * (mL + 1) // 2, dtype=np.float64)
        check(lib().psfp_export_key(self._h, _p(A, C.c_uint64), _p(R, C.c_int8), _p(Lp, C.c_double)), "export_key")
        Sk = np.zeros((self.k, self.k), dtype=np.int64)
        gso = np.zeros((self.k, self.k), dtype=np.float64)
        check(lib().psfp_export_gadget_basis(self._h, _p(Sk, C.c_int64), _p(gso, C.c_double)), "export_gadget_basis")
        return A, (R, Lp, (Sk, gso))

    def export_sqrt_sigma2_rows(self, row0, nrows):
        """Rows [row0, row0 + nrows) of sqrt(Sigma_2) (structured mode: of L_1), packed (row i holds i + 1 entries)."""
        total = (row0 + nrows) * (row0 + nrows + 1) // 2 - row0 * (row0 + 1) // 2
        out = np.zeros(total, dtype=np.float64)
        check(lib().psfp_export_sqrt_sigma2_rows(self._h, C.c_size_t(row0), C.c_size_t(nrows), _p(out, C.c_double)), "export_sqrt_sigma2_rows")
        return out

    def export_A_R(self):
        """A and R only (no sqrt(Sigma_2))."""
        A = np.zeros((self.n, self.m), dtype=np.uint64)
        R = np.zeros((self.m_bar, self.w), dtype=np.int8)
        check(lib().psfp_export_key(self._h, _p(A, C.c_uint64), _p(R, C.c_int8), None), "export_key")
        return A, R

    def load_key(self, A, R=None, sqrt_sigma2_packed=None):
        """(A, R, factor): the whole trapdoor tuple; (A, R): the factor is recomputed from R with the handle's s (mp_perturbation.rs:227-231);
        (A,): the public key alone -- a verifier's handle (f_a, check_domain, samp_d)."""
        A = np.ascontiguousarray(A, dtype=np.uint64)
        assert A.shape == (self.n, self.m)
        Rp = Lpp = None
        if R is not None:
            R = np.ascontiguousarray(R, dtype=np.int8)
            assert R.shape == (self.m_bar, self.w)
            Rp = _p(R, C.c_int8)
        if sqrt_sigma2_packed is not None:
            Lp = np.ascontiguousarray(sqrt_sigma2_packed, dtype=np.float64)
            mL = self.m_bar if self.structured else self.m
            assert R is not None and Lp.size == mL * (mL + 1) // 2
            Lpp = _p(Lp, C.c_double)
        check(lib().psfp_load_key(self._h, _p(A, C.c_uint64), Rp, Lpp), "load_key")

    def load_trapdoor(self, R, A=None):
        """(A, R) without a factor and without computing one (psfp_load_trapdoor): the state compute_sqrt_sigma_2 -- a pure function of mat_r and
        mat_sigma in the reference, mp_perturbation.rs:111 -- starts from.  samp_p raises PSF_ERR_NO_KEY until compute_sqrt_sigma_2 has run."""
        R = np.ascontiguousarray(R, dtype=np.int8)
        assert R.shape == (self.m_bar, self.w)
        Ap = None
        if A is not None:
            A = np.ascontiguousarray(A, dtype=np.uint64)
            assert A.shape == (self.n, self.m)
            Ap = _p(A, C.c_uint64)
        check(lib().psfp_load_trapdoor(self._h, Ap, _p(R, C.c_int8)), "load_trapdoor")

    def compute_sqrt_sigma_2(self, s_cov=None, sigma=None):
        """mp_perturbation.rs:111-139: Sigma = s_cov^2 I, or any symmetric m x m covariance `sigma` (a full matrix or its packed lower triangle)."""
        if sigma is None:
            check(lib().psfp_compute_sqrt_sigma_2(self._h, C.c_double(s_cov)), "compute_sqrt_sigma_2")
            return
        sg = np.asarray(sigma, dtype=np.float64)
        if sg.ndim == 2:
            assert sg.shape == (self.m, self.m)
            sg = sg[np.tril_indices(self.m)]                  # row-major lower triangle: row i holds i + 1 entries
        sg = np.ascontiguousarray(sg)
        assert sg.size == self.m * (self.m + 1) // 2
        check(lib().psfp_compute_sqrt_sigma_2_dense(self._h, _p(sg, C.c_double)), "compute_sqrt_sigma_2_dense")

    # ---- stage-level access (parity tests) ---------------------------------------------------------------
    def samp_p_stages(self, u, seed=0, first_index=0):
        u2 = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, self.n)
        B = u2.shape[0]
        d, x = np.zeros((B, self.m)), np.zeros((B, self.m))
        p, e = np.zeros((B, self.m), dtype=np.int64), np.zeros((B, self.m), dtype=np.int64)
        v, z = np.zeros((B, self.n), dtype=np.uint64), np.zeros((B, self.w), dtype=np.int64)
        check(lib().psfp_samp_p_stages(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64),
                                       _p(d, C.c_double), _p(x, C.c_double), _p(p, C.c_int64), _p(v, C.c_uint64),
                                       _p(z, C.c_int64), _p(e, C.c_int64)), "samp_p_stages")
        return dict(d=d, x=x, p=p, v=v, z=z, e=e)

    # ---- device-resident API (torch tensors / raw pointers) ----------------------------------------------
    def samp_p_dev(self, d_u_ptr, d_e_ptr, B, seed=0, first_index=0, stream=None):
        check(lib().psfp_samp_p_dev(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), C.c_void_p(d_u_ptr),
                                    C.c_void_p(d_e_ptr), C.c_void_p(stream or 0)), "samp_p_dev")

    def f_a_dev(self, d_e_ptr, d_u_ptr, d_ok_ptr, B, stream=None):
        check(lib().psfp_f_a_dev(self._h, C.c_size_t(B), C.c_void_p(d_e_ptr), C.c_void_p(d_u_ptr), C.c_void_p(d_ok_ptr),
                                 C.c_void_p(stream or 0)), "f_a_dev")

    def uniform_targets_dev(self, d_u_ptr, B, seed=0, first_index=0, stream=None):
        check(lib().psfp_uniform_targets_dev(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B),
                                             C.c_void_p(d_u_ptr), C.c_void_p(stream or 0)), "uniform_targets_dev")

    def last_status(self):
        return lib().psfp_last_status(self._h)

    def enable_timing(self, on=True):
        check(lib().psfp_enable_timing(self._h, C.c_int(1 if on else 0)), "enable_timing")

    def get_timing(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_double * 64)()
        cnt = C.c_size_t(64)
        check(lib().psfp_get_timing(self._h, names, C.c_size_t(4096), ms, C.byref(cnt)), "get_timing")
        nm = names.value.decode().split(";") if names.value else []
        return list(zip(nm, list(ms)[:cnt.value]))


def samp_p_multi(psfs, u, seed=0, first_index=0, out=None):
    """psfp_samp_p_multi: one batch over several PSFPerturbation handles (one per GPU, same key), rows cut into contiguous shares.
    out: a C-contiguous int64 array of B x m to fill (a loop of calls reuses it instead of touching a fresh gigabyte per step)."""
    n, m = psfs[0].n, psfs[0].m
    u2 = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, n)
    B = u2.shape[0]
    e = np.zeros((B, m), dtype=np.int64) if out is None else out
    assert e.dtype == np.int64 and e.shape == (B, m) and e.flags["C_CONTIGUOUS"]
    arr = (C.c_void_p * len(psfs))(*[p._h for p in psfs])
    check(lib().psfp_samp_p_multi(arr, C.c_int(len(psfs)), C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64), _p(e, C.c_int64)),
          "samp_p_multi")
    return e


def multi_timing(psf):
    """(launched_ms, done_ms) of this handle inside the last samp_p_multi call, measured from the start of that call; (-1, -1) if it had no rows."""
    a, b = C.c_double(0), C.c_double(0)
    check(lib().psfp_get_multi_timing(psf._h, C.byref(a), C.byref(b)), "multi_timing")
    return a.value, b.value


class PSFGPV:
    """gpv.rs:53-57 / impl PSF :59-225 on one MI355X.  The trapdoor (short basis, GSO) is exchanged TRANSPOSED:
    row i = basis vector i = column i of the reference's matrices."""

    def __init__(self, gp, s, device=0):
        self.gp, self.s, self.device = gp, float(s), device
        prm = GpvParams(gp.c, self.s, device, 0)
        h = C.c_void_p()
        check(lib().psfgpv_create(C.byref(prm), C.byref(h)), "PSFGPV")
        self._h = h
        self._destroy = lib().psfgpv_destroy
        self._destroy.argtypes = [C.c_void_p]
        self.n, self.k, self.m_bar = gp.n, gp.k, gp.m_bar
        self.w = gp.n * gp.k
        self.m = self.m_bar + self.w

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def trap_gen(self, seed=0, export=True):
        """gpv.rs:83-94.  Returns (A, (basis_t, gso_t)) unless export=False."""
        check(lib().psfgpv_trap_gen(self._h, C.c_uint64(seed)), "trap_gen")
        return self.export_key() if export else None

    def export_key(self, with_R=False):
        A = np.zeros((self.n, self.m), dtype=np.uint64)
        bt = np.zeros((self.m, self.m), dtype=np.int32)
        gt = np.zeros((self.m, self.m), dtype=np.float64)
        R = np.zeros((self.m_bar, self.w), dtype=np.int8) if with_R else None
        check(lib().psfgpv_export_key(self._h, _p(A, C.c_uint64), _p(R, C.c_int8) if with_R else None, _p(bt, C.c_int32),
                                      _p(gt, C.c_double)), "export_key")
        return (A, (bt, gt)) if not with_R else (A, R, (bt, gt))

    def load_key(self, A, basis_t, gso_t):
        A = np.ascontiguousarray(A, dtype=np.uint64)
        bt = np.ascontiguousarray(basis_t, dtype=np.int32)
        gt = np.ascontiguousarray(gso_t, dtype=np.float64)
        assert A.shape == (self.n, self.m) and bt.shape == (self.m, self.m) and gt.shape == (self.m, self.m)
        check(lib().psfgpv_load_key(self._h, _p(A, C.c_uint64), _p(bt, C.c_int32), _p(gt, C.c_double)), "load_key")

    def samp_d(self, seed=0, B=None, first_index=0):
        """gpv.rs:113-116"""
        nb = 1 if B is None else B
        e = np.zeros((nb, self.m), dtype=np.int64)
        check(lib().psfgpv_samp_d(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(nb), _p(e, C.c_int64)), "samp_d")
        return e[0] if B is None else e

    def samp_p(self, u, seed=0, first_index=0, out=None):
        """gpv.rs:152-161; out: optional (B, m) int64 C-contiguous array to fill (a reused buffer avoids the first-touch page faults of a fresh one)"""
        u = np.ascontiguousarray(u, dtype=np.uint64)
        single = u.ndim == 1
        u2 = u.reshape(-1, self.n)
        B = u2.shape[0]
        if out is None:
            e = np.empty((B, self.m), dtype=np.int64)
        else:
            e = out
            assert e.dtype == np.int64 and e.shape == (B, self.m) and e.flags.c_contiguous
        check(lib().psfgpv_samp_p(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64),
                                  _p(e, C.c_int64)), "samp_p")
        return e[0] if single else e

    def f_a(self, sigma):
        """gpv.rs:190-193"""
        sigma = np.ascontiguousarray(sigma, dtype=np.int64)
        single = sigma.ndim == 1
        if sigma.ndim > 2 or sigma.shape[-1] != self.m:
            raise PsfError(_ffi.ERR_DOMAIN, "f_a")
        e2 = sigma.reshape(-1, self.m)
        u = np.zeros((e2.shape[0], self.n), dtype=np.uint64)
        check(lib().psfgpv_f_a(self._h, C.c_size_t(e2.shape[0]), _p(e2, C.c_int64), _p(u, C.c_uint64)), "f_a")
        return u[0] if single else u

    def check_domain(self, sigma):
        """gpv.rs:219-224"""
        sigma = np.ascontiguousarray(sigma, dtype=np.int64)
        single = sigma.ndim == 1
        e2 = sigma.reshape(1, -1) if single else sigma
        ok = np.zeros(e2.shape[0], dtype=np.uint8)
        check(lib().psfgpv_check_domain(self._h, C.c_size_t(e2.shape[0]), _p(e2, C.c_int64), C.c_size_t(e2.shape[1]),
                                        _p(ok, C.c_uint8)), "check_domain")
        return bool(ok[0]) if single else ok.astype(bool)

    def samp_p_async(self, u, out, seed=0, first_index=0):
        """psfgpv_samp_p_async: enqueue B samp_p calls on host buffers and return; `out` ((B, m) int64, C-contiguous) is complete after wait().
        At most two calls are in flight per handle; `out` must stay alive (and untouched) until wait() -- the wrapper keeps a reference."""
        u2 = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, self.n)
        B = u2.shape[0]
        assert out.dtype == np.int64 and out.shape == (B, self.m) and out.flags.c_contiguous
        check(lib().psfgpv_samp_p_async(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64), _p(out, C.c_int64)),
              "samp_p_async")
        self._inflight = getattr(self, "_inflight", []) + [out]
        return out

    def wait(self):
        """psfgpv_wait: every asynchronous call of this handle has completed; raises PsfError with the first failure (oldest call first)."""
        try:
            check(lib().psfgpv_wait(self._h), "wait")
        finally:
            self._inflight = []

    def next_ticket(self):
        """the ticket the next asynchronous call of this handle will carry (psfgpv_async_next_ticket)"""
        f = lib().psfgpv_async_next_ticket
        f.restype = C.c_uint64
        return int(f(self._h))

    def wait_ticket(self, ticket):
        """waits for the asynchronous call with this ticket (and the older one in flight) and raises if THAT call failed (psfgpv_wait_ticket)"""
        check(lib().psfgpv_wait_ticket(self._h, C.c_uint64(ticket)), "wait_ticket")

    # device-resident API
    def samp_p_dev(self, d_u_ptr, d_e_ptr, B, seed=0, first_index=0, stream=None):
        check(lib().psfgpv_samp_p_dev(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), C.c_void_p(d_u_ptr),
                                      C.c_void_p(d_e_ptr), C.c_void_p(stream or 0)), "samp_p_dev")

    def f_a_dev(self, d_e_ptr, d_u_ptr, d_ok_ptr, B, stream=None):
        check(lib().psfgpv_f_a_dev(self._h, C.c_size_t(B), C.c_void_p(d_e_ptr), C.c_void_p(d_u_ptr), C.c_void_p(d_ok_ptr),
                                   C.c_void_p(stream or 0)), "f_a_dev")

    def uniform_targets_dev(self, d_u_ptr, B, seed=0, first_index=0, stream=None):
        check(lib().psfgpv_uniform_targets_dev(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B),
                                               C.c_void_p(d_u_ptr), C.c_void_p(stream or 0)), "uniform_targets_dev")

    def last_status(self):
        return lib().psfgpv_last_status(self._h)

    def enable_timing(self, on=True):
        check(lib().psfgpv_enable_timing(self._h, C.c_int(1 if on else 0)), "enable_timing")

    @property
    def two_pass(self):
        """large moduli (q sqrt(n) > 2^13 s): samp_p draws in two passes, see include/psf_mi355x.h"""
        return bool(lib().psfgpv_two_pass(self._h))

    def nearest_plane_stats(self):
        """(64-row blocks walked by the nearest plane, 1 if the last samp_p call recombined in 64-bit integers instead of int8 planes)."""
        a, b = C.c_size_t(0), C.c_size_t(0)
        check(lib().psfgpv_get_nearest_plane_stats(self._h, C.byref(a), C.byref(b)), "nearest_plane_stats")
        return a.value, b.value

    def get_timing(self):
        a, b = C.c_double(0), C.c_double(0)
        check(lib().psfgpv_get_timing(self._h, C.byref(a), C.byref(b)), "get_timing")
        return {"k_np_solve": a.value, "nearest_plane": b.value}

    def nearest_plane_form(self):
        """(form, G, blocks, reruns) of the last samp_p call: form 1 = the walk of gpv.rs:160 in one launch (k_np_walk<G>), 0 = one k_np_step<G> launch per 64-row
        block; reruns = walks of this handle that gave up waiting on a shared GPU and were walked again by k_np_walk_solo (include/psf_mi355x.h)."""
        f, g, b, r = C.c_int(0), C.c_int(0), C.c_size_t(0), C.c_uint64(0)
        check(lib().psfgpv_get_nearest_plane_form(self._h, C.byref(f), C.byref(g), C.byref(b), C.byref(r)), "nearest_plane_form")
        return f.value, g.value, b.value, r.value

    def _debug_set_walk(self, form=-1, spins=0):
        """tests: force the walk's launch form (-1 by batch size, 0 per block, 1 one launch where it fits) / the poll limit of its waits (1: every wait gives up)"""
        check(lib().psfgpv_debug_set_walk(self._h, C.c_int(form), C.c_uint(spins)), "debug_set_walk")

    def _debug_set_split(self, split=-1):
        """tests: the two-halves form of large launch-per-block batches (-1 by shape and size, 0 never, 1 whenever the shape allows)"""
        check(lib().psfgpv_debug_set_split(self._h, C.c_int(split)), "debug_set_split")

    def _debug_last_parts(self):
        return int(lib().psfgpv_debug_last_parts(self._h))


class PSFGPVRing:
    """gpv_ring.rs:62-67 / impl PSF :69-284 on one MI355X.  Polynomials are coefficient rows (constant term first):
    a: (k+2) x n, r / e: k x n, domain elements: (k+2) x n, range elements: n."""

    def __init__(self, gp, s, s_td, device=0):
        self.gp, self.s, self.s_td, self.device = gp, float(s), float(s_td), device
        prm = RingParams(gp.c, self.s, self.s_td, device, 0)
        h = C.c_void_p()
        check(lib().psfring_create(C.byref(prm), C.byref(h)), "PSFGPVRing")
        self._h = h
        self._destroy = lib().psfring_destroy
        self.n, self.k = gp.n, gp.k
        self.K = gp.k + 2
        self.d = self.n * self.K

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def trap_gen(self, seed=0):
        """gpv_ring.rs:91-98.  Returns (a, (r, e))."""
        check(lib().psfring_trap_gen(self._h, C.c_uint64(seed)), "trap_gen")
        a, r, e, _, _ = self.export_key(basis=False)
        return a, (r, e)

    def load_key(self, a, r, e):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        r = np.ascontiguousarray(r, dtype=np.int64)
        e = np.ascontiguousarray(e, dtype=np.int64)
        assert a.shape == (self.K, self.n) and r.shape == (self.k, self.n) and e.shape == (self.k, self.n)
        check(lib().psfring_load_key(self._h, _p(a, C.c_uint64), _p(r, C.c_int64), _p(e, C.c_int64)), "load_key")

    def export_key(self, basis=True):
        a = np.zeros((self.K, self.n), dtype=np.uint64)
        r = np.zeros((self.k, self.n), dtype=np.int64)
        e = np.zeros((self.k, self.n), dtype=np.int64)
        bt = np.zeros((self.d, self.d), dtype=np.int32) if basis else None
        gt = np.zeros((self.d, self.d), dtype=np.float64) if basis else None
        check(lib().psfring_export_key(self._h, _p(a, C.c_uint64), _p(r, C.c_int64), _p(e, C.c_int64),
                                       _p(bt, C.c_int32) if basis else None, _p(gt, C.c_double) if basis else None), "export_key")
        return a, r, e, bt, gt

    def samp_d(self, seed=0, B=None, first_index=0):
        """gpv_ring.rs:118-122"""
        nb = 1 if B is None else B
        sg = np.zeros((nb, self.K, self.n), dtype=np.int64)
        check(lib().psfring_samp_d(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(nb), _p(sg, C.c_int64)), "samp_d")
        return sg[0] if B is None else sg

    def samp_p(self, u, seed=0, first_index=0, out=None):
        """gpv_ring.rs:160-212; u: n coefficients (or B x n); out: optional (B, K, n) int64 C-contiguous array to fill."""
        u = np.ascontiguousarray(u, dtype=np.uint64)
        single = u.ndim == 1
        u2 = u.reshape(-1, self.n)
        B = u2.shape[0]
        if out is None:
            sg = np.empty((B, self.K, self.n), dtype=np.int64)
        else:
            sg = out
            assert sg.dtype == np.int64 and sg.shape == (B, self.K, self.n) and sg.flags.c_contiguous
        check(lib().psfring_samp_p(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64),
                                   _p(sg, C.c_int64)), "samp_p")
        return sg[0] if single else sg

    def f_a(self, sigma):
        """gpv_ring.rs:243-247"""
        sigma = np.ascontiguousarray(sigma, dtype=np.int64)
        if sigma.shape[-2:] != (self.K, self.n) or sigma.ndim > 3:
            raise PsfError(_ffi.ERR_DOMAIN, "f_a")
        single = sigma.ndim == 2
        s2 = sigma.reshape(-1, self.d)
        u = np.zeros((s2.shape[0], self.n), dtype=np.uint64)
        check(lib().psfring_f_a(self._h, C.c_size_t(s2.shape[0]), _p(s2, C.c_int64), _p(u, C.c_uint64)), "f_a")
        return u[0] if single else u

    def check_domain(self, sigma):
        """gpv_ring.rs:274-283: column vector of k+2 polynomials with |iota(sigma)|^2 <= s^2 n (k+2)."""
        sigma = np.ascontiguousarray(sigma, dtype=np.int64)
        if sigma.ndim == 2:
            if sigma.shape[1] != self.n:
                return False
            s2 = sigma.reshape(1, -1)
            single = True
        else:
            s2 = sigma.reshape(sigma.shape[0], -1)
            single = False
        ok = np.zeros(s2.shape[0], dtype=np.uint8)
        check(lib().psfring_check_domain(self._h, C.c_size_t(s2.shape[0]), _p(s2, C.c_int64), C.c_size_t(s2.shape[1]),
                                         _p(ok, C.c_uint8)), "check_domain")
        return bool(ok[0]) if single else ok.astype(bool)

    def samp_p_async(self, u, out, seed=0, first_index=0):
        """psfring_samp_p_async: as PSFGPV.samp_p_async; `out` is (B, k+2, n) int64, complete after wait()."""
        u2 = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, self.n)
        B = u2.shape[0]
        assert out.dtype == np.int64 and out.shape == (B, self.K, self.n) and out.flags.c_contiguous
        check(lib().psfring_samp_p_async(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), _p(u2, C.c_uint64), _p(out, C.c_int64)),
              "samp_p_async")
        self._inflight = getattr(self, "_inflight", []) + [out]
        return out

    def wait(self):
        try:
            check(lib().psfring_wait(self._h), "wait")
        finally:
            self._inflight = []

    def next_ticket(self):
        """the ticket the next asynchronous call of this handle will carry (psfring_async_next_ticket)"""
        f = lib().psfring_async_next_ticket
        f.restype = C.c_uint64
        return int(f(self._h))

    def wait_ticket(self, ticket):
        """waits for the asynchronous call with this ticket (and the older one in flight) and raises if THAT call failed (psfring_wait_ticket)"""
        check(lib().psfring_wait_ticket(self._h, C.c_uint64(ticket)), "wait_ticket")

    def samp_p_dev(self, d_u_ptr, d_sigma_ptr, B, seed=0, first_index=0, stream=None):
        check(lib().psfring_samp_p_dev(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B), C.c_void_p(d_u_ptr),
                                       C.c_void_p(d_sigma_ptr), C.c_void_p(stream or 0)), "samp_p_dev")

    def f_a_dev(self, d_sigma_ptr, d_u_ptr, d_ok_ptr, B, stream=None):
        check(lib().psfring_f_a_dev(self._h, C.c_size_t(B), C.c_void_p(d_sigma_ptr), C.c_void_p(d_u_ptr), C.c_void_p(d_ok_ptr),
                                    C.c_void_p(stream or 0)), "f_a_dev")

    def uniform_targets_dev(self, d_u_ptr, B, seed=0, first_index=0, stream=None):
        check(lib().psfring_uniform_targets_dev(self._h, C.c_uint64(seed), C.c_uint64(first_index), C.c_size_t(B),
                                                C.c_void_p(d_u_ptr), C.c_void_p(stream or 0)), "uniform_targets_dev")

    def last_status(self):
        return lib().psfring_last_status(self._h)

    def enable_timing(self, on=True):
        check(lib().psfring_enable_timing(self._h, C.c_int(1 if on else 0)), "enable_timing")

    def get_timing(self):
        a, b = C.c_double(0), C.c_double(0)
        check(lib().psfring_get_timing(self._h, C.byref(a), C.byref(b)), "get_timing")
        return {"k_np_solve": a.value, "nearest_plane": b.value}

    def nearest_plane_form(self):
        """(form, G, blocks, reruns) of the last samp_p call: form 1 = the walk of gpv.rs:160 in one launch (k_np_walk<G>), 0 = one k_np_step<G> launch per 64-row
        block; reruns = walks of this handle that gave up waiting on a shared GPU and were walked again by k_np_walk_solo (include/psf_mi355x.h)."""
        f, g, b, r = C.c_int(0), C.c_int(0), C.c_size_t(0), C.c_uint64(0)
        check(lib().psfring_get_nearest_plane_form(self._h, C.byref(f), C.byref(g), C.byref(b), C.byref(r)), "nearest_plane_form")
        return f.value, g.value, b.value, r.value

    def _debug_set_walk(self, form=-1, spins=0):
        """tests: force the walk's launch form (-1 by batch size, 0 per block, 1 one launch where it fits) / the poll limit of its waits (1: every wait gives up)"""
        check(lib().psfring_debug_set_walk(self._h, C.c_int(form), C.c_uint(spins)), "debug_set_walk")

    def _debug_set_split(self, split=-1):
        """tests: the two-halves form of large launch-per-block batches (-1 by shape and size, 0 never, 1 whenever the shape allows)"""
        check(lib().psfring_debug_set_split(self._h, C.c_int(split)), "debug_set_split")

    def _debug_last_parts(self):
        return int(lib().psfring_debug_last_parts(self._h))
