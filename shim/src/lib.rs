//! `impl PSF` for the MI355X-native preimage-sampling library (`libpsf_mi355x.so`, C ABI in `include/psf_mi355x.h`).
//!
//! Drop-in for ONE hot path of qfall-tools: `PSF::samp_p` (and the `trap_gen` / `samp_d` / `f_a` / `check_domain` around it) of
//! `PSFPerturbation` (src/primitive/psf/mp_perturbation.rs:193-403), `PSFGPV` (gpv.rs:59-225) and `PSFGPVRing`
//! (gpv_ring.rs:69-284).  The types below wrap the reference's own parameter structs and implement the reference's trait
//! (src/primitive/psf.rs:39-81) with the same associated types, so a caller switches by changing the constructor.
//!
//! UNVERIFIED SOURCE.  No Rust toolchain, qfall-math or FLINT exists in the build environment of this repository, so this file has
//! never been compiled.  Calls into qfall-math are restricted to methods the reference's own sources use (`get_entry`, `set_entry`,
//! `get_num_rows`, `get_num_columns`, `get_q`, `get_mod`, `get_coeff`, `set_coeff`, `MatZ::new`, `MatZq::new`, `MatQ::new`,
//! `Z::from`, `Q::from`, `i64::try_from(&Z)`, `f64::from(&Q)`); where a generic parameter had to be guessed it is annotated.
//!
//! Semantics that differ from the reference, by construction of the library:
//! * randomness comes from Philox streams keyed by a 64-bit seed (the trait has no seed argument, psf.rs:48-80): every wrapper owns a
//!   `seed` and a call counter; call `reseed` for reproducible runs;
//! * where the reference panics (`.unwrap()` / `assert!`: mp_perturbation.rs:190,315,333,367) the library returns a status and the shim
//!   panics with the library's message;
//! * key material handed to `samp_p` / `f_a` is uploaded when it differs from what the handle holds (the reference is stateless).
#![allow(non_snake_case)]

pub mod ffi;

use std::cell::{Cell, RefCell};
use std::ffi::CStr;

use qfall_math::integer::{MatPolyOverZ, MatZ, PolyOverZ, Z};
use qfall_math::integer_mod_q::{MatPolynomialRingZq, MatZq, Modulus};
use qfall_math::rational::{MatQ, Q};
use qfall_math::traits::*;
use qfall_tools::primitive::psf::{PSF, PSFGPV, PSFGPVRing, PSFPerturbation};
use qfall_tools::sample::g_trapdoor::gadget_parameters::{GadgetParameters, GadgetParametersRing};

// ------------------------------------------------------------------------------------------------------------------------
// status handling and conversions between qfall-math values and the flat row-major arrays of the ABI
// ------------------------------------------------------------------------------------------------------------------------

/// Panics with the library's message unless `status` is PSF_OK -- the reference panics in the same situations.
fn check(status: ffi::psf_status, what: &str) {
    if status != ffi::PSF_OK {
        let msg = unsafe { CStr::from_ptr(ffi::psf_status_string(status)) }.to_string_lossy().into_owned();
        panic!("{what}: psf_status {status} ({msg})");
    }
}

fn z_to_u64(z: &Z) -> u64 {
    u64::try_from(z).expect("value does not fit 64 bits")
}

fn z_to_i64(z: &Z) -> i64 {
    i64::try_from(z).expect("value does not fit 64 bits")
}

/// `GadgetParameters` -> `psf_gadget_params` (gadget_parameters.rs:44-52; the distribution is PlusMinusOneZero inside the library).
fn gp_to_c(gp: &GadgetParameters) -> ffi::psf_gadget_params {
    ffi::psf_gadget_params { n: z_to_u64(&gp.n), k: z_to_u64(&gp.k), m_bar: z_to_u64(&gp.m_bar), base: z_to_u64(&gp.base), q: z_to_u64(&Z::from(&gp.q)) }
}

/// `GadgetParametersRing` -> `psf_gadget_params` (gadget_parameters.rs:73-81; modulus polynomial X^n + 1, distribution SampleZ).
fn gp_ring_to_c(gp: &GadgetParametersRing) -> ffi::psf_gadget_params {
    ffi::psf_gadget_params { n: z_to_u64(&gp.n), k: z_to_u64(&gp.k), m_bar: z_to_u64(&gp.m_bar), base: z_to_u64(&gp.base), q: z_to_u64(&Z::from(&gp.modulus.get_q())) }
}

/// MatZq -> rows of least non-negative residues.
fn matzq_to_rows(a: &MatZq) -> Vec<u64> {
    let (rows, cols) = (a.get_num_rows(), a.get_num_columns());
    let mut out = Vec::with_capacity((rows * cols) as usize);
    for i in 0..rows {
        for j in 0..cols {
            let entry: Z = a.get_entry(i, j).unwrap(); // [unverified] GetEntry<Z> for MatZq: the representative in [0, q)
            out.push(z_to_u64(&entry));
        }
    }
    out
}

fn matzq_from_rows(rows: i64, cols: i64, q: &Modulus, data: &[u64]) -> MatZq {
    let mut out = MatZq::new(rows, cols, q);
    for i in 0..rows {
        for j in 0..cols {
            out.set_entry(i, j, Z::from(data[(i * cols + j) as usize])).unwrap();
        }
    }
    out
}

fn matz_to_rows_i64(a: &MatZ) -> Vec<i64> {
    let (rows, cols) = (a.get_num_rows(), a.get_num_columns());
    let mut out = Vec::with_capacity((rows * cols) as usize);
    for i in 0..rows {
        for j in 0..cols {
            let entry: Z = a.get_entry(i, j).unwrap();
            out.push(z_to_i64(&entry));
        }
    }
    out
}

fn matz_to_rows_i8(a: &MatZ) -> Vec<i8> {
    matz_to_rows_i64(a).into_iter().map(|v| i8::try_from(v).expect("trapdoor entry outside {-1,0,1}")).collect()
}

fn matz_from_rows<T: Copy + Into<i64>>(rows: i64, cols: i64, data: &[T]) -> MatZ {
    let mut out = MatZ::new(rows, cols);
    for i in 0..rows {
        for j in 0..cols {
            let v: i64 = data[(i * cols + j) as usize].into();
            out.set_entry(i, j, Z::from(v)).unwrap();
        }
    }
    out
}

/// Lower-triangular MatQ -> packed rows (row i holds i + 1 doubles): the form psfp_load_key takes for sqrt(Sigma_2).
fn matq_lower_to_packed(l: &MatQ) -> Vec<f64> {
    let m = l.get_num_rows();
    let mut out = Vec::with_capacity((m * (m + 1) / 2) as usize);
    for i in 0..m {
        for j in 0..=i {
            let entry: Q = l.get_entry(i, j).unwrap();
            out.push(f64::from(&entry));
        }
    }
    out
}

fn matq_lower_from_packed(m: i64, packed: &[f64]) -> MatQ {
    let mut out = MatQ::new(m, m);
    let mut at = 0usize;
    for i in 0..m {
        for j in 0..=i {
            out.set_entry(i, j, Q::from(packed[at])).unwrap();
            at += 1;
        }
    }
    out
}

/// rows x cols doubles -> MatQ, optionally transposed (the ABI ships the Gram-Schmidt vectors one per ROW, the reference one per COLUMN).
fn matq_from_rows(rows: i64, cols: i64, data: &[f64], transpose: bool) -> MatQ {
    let mut out = if transpose { MatQ::new(cols, rows) } else { MatQ::new(rows, cols) };
    for i in 0..rows {
        for j in 0..cols {
            let v = Q::from(data[(i * cols + j) as usize]);
            if transpose { out.set_entry(j, i, v).unwrap() } else { out.set_entry(i, j, v).unwrap() }
        }
    }
    out
}

fn matq_to_rows_t(a: &MatQ) -> Vec<f64> {
    // column c of the reference's matrix becomes row c
    let (rows, cols) = (a.get_num_rows(), a.get_num_columns());
    let mut out = Vec::with_capacity((rows * cols) as usize);
    for c in 0..cols {
        for r in 0..rows {
            let entry: Q = a.get_entry(r, c).unwrap();
            out.push(f64::from(&entry));
        }
    }
    out
}

fn matz_to_rows_t_i32(a: &MatZ) -> Vec<i32> {
    let (rows, cols) = (a.get_num_rows(), a.get_num_columns());
    let mut out = Vec::with_capacity((rows * cols) as usize);
    for c in 0..cols {
        for r in 0..rows {
            let entry: Z = a.get_entry(r, c).unwrap();
            out.push(i32::try_from(z_to_i64(&entry)).expect("basis entry does not fit 32 bits"));
        }
    }
    out
}

/// 1 x cols (or cols x 1) MatPolyOverZ -> cols rows of n coefficients, constant term first.
fn matpoly_to_rows(p: &MatPolyOverZ, n: i64) -> Vec<i64> {
    let (rows, cols) = (p.get_num_rows(), p.get_num_columns());
    let mut out = Vec::with_capacity((rows * cols * n) as usize);
    for i in 0..rows {
        for j in 0..cols {
            let poly: PolyOverZ = p.get_entry(i, j).unwrap();
            for c in 0..n {
                let coeff: Z = poly.get_coeff(c).unwrap();
                out.push(z_to_i64(&coeff));
            }
        }
    }
    out
}

fn matpoly_from_rows(rows: i64, cols: i64, n: i64, data: &[i64]) -> MatPolyOverZ {
    let mut out = MatPolyOverZ::new(rows, cols);
    for i in 0..rows {
        for j in 0..cols {
            let mut poly = PolyOverZ::default();
            for c in 0..n {
                poly.set_coeff(c, data[((i * cols + j) * n + c) as usize]).unwrap();
            }
            out.set_entry(i, j, &poly).unwrap();
        }
    }
    out
}

/// MatPolynomialRingZq -> least non-negative residues of every coefficient, entry by entry.
fn matpolyring_to_rows(a: &MatPolynomialRingZq, n: i64) -> Vec<u64> {
    let repr: MatPolyOverZ = a.get_representative_least_nonnegative_residue();
    matpoly_to_rows(&repr, n).into_iter().map(|v| u64::try_from(v).unwrap()).collect()
}

/// 64 bits from the operating system's generator (/dev/urandom), for `from_entropy`.
fn os_seed() -> u64 {
    use std::io::Read;
    let mut b = [0u8; 8];
    std::fs::File::open("/dev/urandom").and_then(|mut f| f.read_exact(&mut b)).expect("no /dev/urandom");
    u64::from_le_bytes(b)
}

/// Call counter -> seed of the next sampling call: seeds of different calls never coincide for one wrapper.
fn next_seed(seed: &Cell<u64>, calls: &Cell<u64>) -> u64 {
    let c = calls.get();
    calls.set(c + 1);
    seed.get().wrapping_add(0x9E3779B97F4A7C15u64.wrapping_mul(c + 1))
}

// ------------------------------------------------------------------------------------------------------------------------
// PSFPerturbation (mp_perturbation.rs:57-62, impl PSF :193-403)
// ------------------------------------------------------------------------------------------------------------------------

/// `PSFPerturbation` whose `trap_gen` / `samp_d` / `samp_p` / `f_a` / `check_domain` run on one MI355X.
pub struct GpuPSFPerturbation {
    /// the reference's parameter struct (gp, r, s): kept public so that code written against it keeps compiling
    pub params: PSFPerturbation,
    handle: *mut ffi::psfp_handle,
    seed: Cell<u64>,
    calls: Cell<u64>,
    /// What the handle holds: ONE state, so that no stale cache can outlive a key change (trap_gen / install_key / f_a with another matrix).
    held: RefCell<Held>,
}

/// What the device handle holds, identified by FINGERPRINTS of the caller's matrices.  A (n x m) and R (m_bar x nk) are hashed ENTRY BY ENTRY
/// (`full_print`: small integers, O(n m) reads -- far below the upload they save); sqrt(Sigma_2) (m x m rationals, 4.7e8 `fmpq` at n = 512) is hashed
/// through its dimensions, first and last row, diagonal and a strided sample of eight entries per row (`sampled_print`).  The reference's
/// `samp_p(a, td, u)` is a pure function of its arguments: a caller's A or R that differs anywhere is re-installed; a sqrt(Sigma_2) that differs
/// only outside the sample is not noticed -- callers that edit single entries of a factor in place call `install_key` again.
#[derive(Clone, PartialEq)]
enum Held {
    Nothing,
    /// the public matrix alone (a verifier's handle: `f_a`, `check_domain`, `samp_d`)
    Public(u64),
    /// (A, R, sqrt(Sigma_2))
    Full(u64, u64, u64),
}

fn fnv(h: u64, v: u64) -> u64 {
    (h ^ v).wrapping_mul(0x100000001b3)
}

/// every entry, row by row
fn full_print(rows: i64, cols: i64, entry: &dyn Fn(i64, i64) -> u64) -> u64 {
    let mut h = fnv(fnv(0xcbf29ce484222325, rows as u64), cols as u64);
    for i in 0..rows {
        for j in 0..cols {
            h = fnv(h, entry(i, j));
        }
    }
    h
}

/// dimensions + first row + last row + diagonal of a square matrix + eight strided entries of every row
fn sampled_print(rows: i64, cols: i64, entry: &dyn Fn(i64, i64) -> u64) -> u64 {
    let mut h = fnv(fnv(0xcbf29ce484222325, rows as u64), cols as u64);
    if rows == 0 || cols == 0 {
        return h;
    }
    for j in 0..cols {
        h = fnv(h, entry(0, j));
    }
    for j in 0..cols {
        h = fnv(h, entry(rows - 1, j));
    }
    if rows == cols {
        for i in 0..rows {
            h = fnv(h, entry(i, i));
        }
    }
    for i in 0..rows {
        for t in 0..8i64 {
            h = fnv(h, entry(i, (i * 7 + t * (cols / 8 + 1)) % cols));
        }
    }
    h
}

/// an integer entry as 64 hash bits: its value when it fits, a hash of its decimal digits otherwise (entries beyond 64 bits must not all collide)
fn z_bits(z: &Z) -> u64 {
    match i64::try_from(z) {
        Ok(v) => v as u64,
        Err(_) => z.to_string().bytes().fold(0x9e3779b97f4a7c15u64, |h, b| fnv(h, b as u64)),
    }
}

fn print_matzq(a: &MatZq) -> u64 {
    full_print(a.get_num_rows(), a.get_num_columns(), &|i, j| { let z: Z = a.get_entry(i, j).unwrap(); z_bits(&z) })
}
fn print_matz(a: &MatZ) -> u64 {
    full_print(a.get_num_rows(), a.get_num_columns(), &|i, j| { let z: Z = a.get_entry(i, j).unwrap(); z_bits(&z) })
}
/// a d x d short basis (38 M entries at n = 256, q = 3329): the sampled form, like its Gram-Schmidt matrix
fn print_basis(a: &MatZ) -> u64 {
    sampled_print(a.get_num_rows(), a.get_num_columns(), &|i, j| { let z: Z = a.get_entry(i, j).unwrap(); z_bits(&z) })
}
fn print_matq(a: &MatQ) -> u64 {
    sampled_print(a.get_num_rows(), a.get_num_columns(), &|i, j| { let q: Q = a.get_entry(i, j).unwrap(); f64::from(&q).to_bits() })
}

impl GpuPSFPerturbation {
    /// `device`: HIP device ordinal.  Panics if the parameters are outside the library's limits (psf_mi355x.h, psfp_create).
    pub fn new(params: PSFPerturbation, device: i32, seed: u64) -> Self {
        let c = ffi::psfp_params { gp: gp_to_c(&params.gp), r: f64::from(&params.r), s: f64::from(&params.s), device, flags: 0 };
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::psfp_create(&c, &mut handle) }, "psfp_create");
        Self { params, handle, seed: Cell::new(seed), calls: Cell::new(0), held: RefCell::new(Held::Nothing) }
    }

    /// Like `new`, with the seed drawn from the operating system (the reference samples through the OS-seeded thread-local RNG of qfall-math).
    /// A caller-chosen 64-bit seed (`new`, `reseed`) makes runs reproducible and is meant for tests and benches: it bounds the entropy of every
    /// key and preimage at 64 bits.
    pub fn from_entropy(params: PSFPerturbation, device: i32) -> Self {
        Self::new(params, device, os_seed())
    }

    pub fn reseed(&self, seed: u64) {
        self.seed.set(seed);
        self.calls.set(0);
    }

    fn dims(&self) -> (i64, i64, i64, i64) {
        let gp = gp_to_c(&self.params.gp);
        let (n, k, mb) = (gp.n as i64, gp.k as i64, gp.m_bar as i64);
        (n, k, mb, mb + n * k)
    }

    /// Uploads (A, R, sqrt(Sigma_2)) to the device: the explicit form of what `samp_p` does when it meets a key the handle does not hold.
    /// Call it once per key (a signer's set-up); every later `samp_p` with the same tuple then costs an O(m) fingerprint check and no copy.
    pub fn install_key(&self, a: &MatZq, td: &<Self as PSF>::Trapdoor) {
        let (av, rv, lv) = (matzq_to_rows(a), matz_to_rows_i8(&td.0), matq_lower_to_packed(&td.1));
        *self.held.borrow_mut() = Held::Nothing;
        check(unsafe { ffi::psfp_load_key(self.handle, av.as_ptr(), rv.as_ptr(), lv.as_ptr()) }, "psfp_load_key");
        *self.held.borrow_mut() = Held::Full(print_matzq(a), print_matz(&td.0), print_matq(&td.1));
    }

    /// `install_key` unless the handle already holds this tuple (fingerprints: see `Held`).
    fn ensure_key(&self, a: &MatZq, td: &<Self as PSF>::Trapdoor) {
        let want = Held::Full(print_matzq(a), print_matz(&td.0), print_matq(&td.1));
        if *self.held.borrow() != want {
            self.install_key(a, td);
        }
    }

    /// `PSFPerturbation::compute_sqrt_sigma_2` (mp_perturbation.rs:111-139) on the device for ANY symmetric covariance `mat_sigma`
    /// (the reference's signature; its doctest at :89-107 passes s'^2 I).  A pure function of its arguments, as in the reference: no installed key is
    /// needed and none is produced -- (., mat_r) is installed WITHOUT a factor (`psfp_load_trapdoor`: no Cholesky with the handle's own s in front),
    /// the factor of `mat_sigma` is computed and read back, and the handle is left holding nothing a later `samp_p` / `f_a` could mistake for its key.
    pub fn compute_sqrt_sigma_2(&self, mat_r: &MatZ, mat_sigma: &MatQ) -> MatQ {
        let (_, _, _, m) = self.dims();
        let rv = matz_to_rows_i8(mat_r);
        *self.held.borrow_mut() = Held::Nothing;
        check(unsafe { ffi::psfp_load_trapdoor(self.handle, std::ptr::null(), rv.as_ptr()) }, "psfp_load_trapdoor");
        let sg = matq_lower_to_packed(mat_sigma);
        check(unsafe { ffi::psfp_compute_sqrt_sigma_2_dense(self.handle, sg.as_ptr()) }, "psfp_compute_sqrt_sigma_2_dense");
        let mut l = vec![0f64; (m * (m + 1) / 2) as usize];
        check(unsafe { ffi::psfp_export_sqrt_sigma2_rows(self.handle, 0, m as usize, l.as_mut_ptr()) }, "psfp_export_sqrt_sigma2_rows");
        matq_lower_from_packed(m, &l)
    }

    /// B independent `samp_p` calls with one key: the batched form the library is built for (row b of `targets` = one syndrome).
    pub fn samp_p_batch(&self, a: &MatZq, td: &<Self as PSF>::Trapdoor, targets: &MatZq) -> MatZ {
        let (n, _, _, m) = self.dims();
        self.ensure_key(a, td);
        let b = targets.get_num_rows();
        assert_eq!(targets.get_num_columns(), n, "one syndrome of length n per row");
        let u = matzq_to_rows(targets);
        let mut e = vec![0i64; (b * m) as usize];
        let seed = next_seed(&self.seed, &self.calls);
        check(unsafe { ffi::psfp_samp_p(self.handle, seed, 0, b as usize, u.as_ptr(), e.as_mut_ptr()) }, "psfp_samp_p");
        matz_from_rows(b, m, &e)
    }
}

/// The rows of an asynchronous batch.  It owns the buffers the library reads from and its worker threads write into, and it BORROWS the handle: the
/// handle cannot be dropped while a batch is pending, and the rows can only be taken through `into_matz`, which waits (`psfp_wait`) first.  Dropping a
/// pending batch waits as well, so the buffers are never freed under the workers.
pub struct PendingBatch<'a> {
    owner: Owner<'a>,
    /// the call's ticket (`psfp_async_next_ticket` read right before the call): `into_matz` asks for THIS call's status, whoever joined it in the meantime
    ticket: u64,
    rows: i64,
    cols: i64,
    _u: Vec<u64>,
    e: Vec<i64>,
    waited: Cell<bool>,
}

/// the handle a pending batch belongs to (and waits on)
enum Owner<'a> {
    Perturbation(&'a GpuPSFPerturbation),
    Gpv(&'a GpuPSFGPV),
}

impl<'a> Owner<'a> {
    /// the ticket the next asynchronous call of the handle will carry
    fn next_ticket(&self) -> u64 {
        match self {
            Owner::Perturbation(p) => unsafe { ffi::psfp_async_next_ticket(p.handle) },
            Owner::Gpv(g) => unsafe { ffi::psfgpv_async_next_ticket(g.handle) },
        }
    }
    /// waits for the call with this ticket (and the older one in flight) and returns ITS status: dropping or reading another batch first cannot consume it
    fn wait_ticket(&self, ticket: u64) -> std::os::raw::c_int {
        match self {
            Owner::Perturbation(p) => unsafe { ffi::psfp_wait_ticket(p.handle, ticket) },
            Owner::Gpv(g) => unsafe { ffi::psfgpv_wait_ticket(g.handle, ticket) },
        }
    }
}

impl<'a> PendingBatch<'a> {
    /// the preimages, one per row; waits for this batch (and the older one in flight) and panics if THIS batch failed
    pub fn into_matz(self) -> MatZ {
        check(self.owner.wait_ticket(self.ticket), "wait_ticket");
        self.waited.set(true);
        matz_from_rows(self.rows, self.cols, &self.e)
    }
}

impl<'a> Drop for PendingBatch<'a> {
    fn drop(&mut self) {
        if !self.waited.get() {
            // nothing may write into `e` once it is freed; the status of an abandoned batch is discarded -- and only its own: the other batches keep theirs
            let _ = self.owner.wait_ticket(self.ticket);
        }
    }
}

impl GpuPSFPerturbation {
    /// `samp_p_batch` without waiting (`psfp_samp_p_async`): at most two batches in flight per handle; the rows of batch i cross PCIe and are widened
    /// by worker threads while batch i + 1 computes.  `into_matz` (or dropping the batch) waits.
    pub fn samp_p_batch_async<'a>(&'a self, a: &MatZq, td: &<Self as PSF>::Trapdoor, targets: &MatZq) -> PendingBatch<'a> {
        let (n, _, _, m) = self.dims();
        self.ensure_key(a, td);
        let b = targets.get_num_rows();
        assert_eq!(targets.get_num_columns(), n, "one syndrome of length n per row");
        let u = matzq_to_rows(targets);
        let mut e = vec![0i64; (b * m) as usize];
        let seed = next_seed(&self.seed, &self.calls);
        let owner = Owner::Perturbation(self);
        let ticket = owner.next_ticket();
        check(unsafe { ffi::psfp_samp_p_async(self.handle, seed, 0, b as usize, u.as_ptr(), e.as_mut_ptr()) }, "psfp_samp_p_async");
        PendingBatch { owner, ticket, rows: b, cols: m, _u: u, e, waited: Cell::new(false) }
    }

    /// every asynchronous batch of this handle has completed (`psfp_wait`); panics with the first failure, oldest batch first
    pub fn wait_batches(&self) {
        check(unsafe { ffi::psfp_wait(self.handle) }, "psfp_wait");
    }
}

impl Drop for GpuPSFPerturbation {
    fn drop(&mut self) {
        unsafe { ffi::psfp_destroy(self.handle) }
    }
}

impl PSF for GpuPSFPerturbation {
    type A = MatZq;
    type Trapdoor = (MatZ, MatQ, (MatZ, MatQ));
    type Domain = MatZ;
    type Range = MatZq;

    /// mp_perturbation.rs:221-244 on the device; returns the same tuple shape as the reference.
    fn trap_gen(&self) -> (MatZq, (MatZ, MatQ, (MatZ, MatQ))) {
        let (n, k, mb, m) = self.dims();
        check(unsafe { ffi::psfp_trap_gen(self.handle, next_seed(&self.seed, &self.calls)) }, "psfp_trap_gen");
        let mut a = vec![0u64; (n * m) as usize];
        let mut r = vec![0i8; (mb * n * k) as usize];
        let mut l = vec![0f64; (m * (m + 1) / 2) as usize];
        check(unsafe { ffi::psfp_export_key(self.handle, a.as_mut_ptr(), r.as_mut_ptr(), l.as_mut_ptr()) }, "psfp_export_key");
        let mut sk = vec![0i64; (k * k) as usize];
        let mut gso = vec![0f64; (k * k) as usize];
        check(unsafe { ffi::psfp_export_gadget_basis(self.handle, sk.as_mut_ptr(), gso.as_mut_ptr()) }, "psfp_export_gadget_basis");
        let a_mat = matzq_from_rows(n, m, &self.params.gp.q, &a);
        let r_mat = matz_from_rows(mb, n * k, &r);
        let l_mat = matq_lower_from_packed(m, &l);
        // (S, S~) of mp_perturbation.rs:233-234 are I_n (x) S_k and its GSO: rebuilt from the k x k block
        let mut s_mat = MatZ::new(n * k, n * k);
        let mut s_gso = MatQ::new(n * k, n * k);
        for blk in 0..n {
            for i in 0..k {
                for j in 0..k {
                    s_mat.set_entry(blk * k + i, blk * k + j, Z::from(sk[(i * k + j) as usize])).unwrap();
                    s_gso.set_entry(blk * k + i, blk * k + j, Q::from(gso[(i * k + j) as usize])).unwrap();
                }
            }
        }
        // the handle now holds exactly this tuple; any earlier state (a verifier's public matrix included) is gone
        *self.held.borrow_mut() = Held::Full(print_matzq(&a_mat), print_matz(&r_mat), print_matq(&l_mat));
        (a_mat, (r_mat, l_mat, (s_mat, s_gso)))
    }

    /// mp_perturbation.rs:264-267
    fn samp_d(&self) -> MatZ {
        let (_, _, _, m) = self.dims();
        let mut e = vec![0i64; m as usize];
        check(unsafe { ffi::psfp_samp_d(self.handle, next_seed(&self.seed, &self.calls), 0, 1, e.as_mut_ptr()) }, "psfp_samp_d");
        matz_from_rows(m, 1, &e)
    }

    /// mp_perturbation.rs:304-336: one preimage (use `samp_p_batch` for throughput)
    fn samp_p(&self, a: &MatZq, td: &Self::Trapdoor, u: &MatZq) -> MatZ {
        let (n, _, _, m) = self.dims();
        self.ensure_key(a, td);
        assert!(u.get_num_rows() == n && u.get_num_columns() == 1, "u must be an n x 1 column (mp_perturbation.rs:318)");
        let uv = matzq_to_rows(u);
        let mut e = vec![0i64; m as usize];
        check(unsafe { ffi::psfp_samp_p(self.handle, next_seed(&self.seed, &self.calls), 0, 1, uv.as_ptr(), e.as_mut_ptr()) }, "psfp_samp_p");
        matz_from_rows(m, 1, &e)
    }

    /// mp_perturbation.rs:366-369; panics like the reference's assert! when sigma is not in the domain
    fn f_a(&self, a: &MatZq, sigma: &MatZ) -> MatZq {
        let (n, _, _, m) = self.dims();
        assert!(sigma.get_num_rows() == m && sigma.get_num_columns() == 1, "sigma must be a column vector of length m");
        // PSF::f_a takes the public matrix alone (mp_perturbation.rs:366): a verifier that never saw a trapdoor can call it.  If the caller's A is
        // not the one the handle holds, it is installed as a public-key-only key (psfp_load_key(A, NULL, NULL)); a later samp_p re-installs its tuple.
        let pa = print_matzq(a);
        let same = match *self.held.borrow() {
            Held::Full(ha, _, _) | Held::Public(ha) => ha == pa,
            Held::Nothing => false,
        };
        if !same {
            let av = matzq_to_rows(a);
            *self.held.borrow_mut() = Held::Nothing;
            check(unsafe { ffi::psfp_load_key(self.handle, av.as_ptr(), std::ptr::null(), std::ptr::null()) }, "psfp_load_key");
            *self.held.borrow_mut() = Held::Public(pa);
        }
        let e = matz_to_rows_i64(sigma);
        let mut u = vec![0u64; n as usize];
        check(unsafe { ffi::psfp_f_a(self.handle, 1, e.as_ptr(), u.as_mut_ptr()) }, "psfp_f_a");
        matzq_from_rows(n, 1, &self.params.gp.q, &u)
    }

    /// mp_perturbation.rs:396-402
    fn check_domain(&self, sigma: &MatZ) -> bool {
        if !sigma.is_column_vector() {
            return false;
        }
        let e = matz_to_rows_i64(sigma);
        let mut ok = [0u8; 1];
        check(unsafe { ffi::psfp_check_domain(self.handle, 1, e.as_ptr(), e.len(), ok.as_mut_ptr()) }, "psfp_check_domain");
        ok[0] != 0
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// PSFGPV (gpv.rs:53-57, impl PSF :59-225)
// ------------------------------------------------------------------------------------------------------------------------

pub struct GpuPSFGPV {
    pub params: PSFGPV,
    handle: *mut ffi::psfgpv_handle,
    seed: Cell<u64>,
    calls: Cell<u64>,
    /// fingerprints of the (A, basis, GSO) the handle holds (A entry by entry, the two d x d matrices sampled: `Held`) instead of a deep comparison of two
    /// d x d matrices and a second copy of them (d = 6208 at n = 256, q = 3329)
    held: RefCell<Option<(u64, u64, u64)>>,
}

impl GpuPSFGPV {
    pub fn new(params: PSFGPV, device: i32, seed: u64) -> Self {
        let c = ffi::psfgpv_params { gp: gp_to_c(&params.gp), s: f64::from(&params.s), device, flags: 0 };
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::psfgpv_create(&c, &mut handle) }, "psfgpv_create");
        Self { params, handle, seed: Cell::new(seed), calls: Cell::new(0), held: RefCell::new(None) }
    }

    pub fn reseed(&self, seed: u64) {
        self.seed.set(seed);
        self.calls.set(0);
    }

    fn dims(&self) -> (i64, i64) {
        let gp = gp_to_c(&self.params.gp);
        (gp.n as i64, (gp.m_bar + gp.n * gp.k) as i64)
    }

    /// Uploads (A, basis, GSO): once per key; later calls with the same tuple cost a fingerprint check.
    pub fn install_key(&self, a: &MatZq, basis: &MatZ, gso: &MatQ) {
        // the ABI takes both matrices transposed: row i = basis vector i (psf_mi355x.h, PSFGPV section)
        let (av, bt, gt) = (matzq_to_rows(a), matz_to_rows_t_i32(basis), matq_to_rows_t(gso));
        *self.held.borrow_mut() = None;
        check(unsafe { ffi::psfgpv_load_key(self.handle, av.as_ptr(), bt.as_ptr(), gt.as_ptr()) }, "psfgpv_load_key");
        *self.held.borrow_mut() = Some((print_matzq(a), print_basis(basis), print_matq(gso)));
    }

    fn ensure_key(&self, a: &MatZq, basis: &MatZ, gso: &MatQ) {
        let want = Some((print_matzq(a), print_basis(basis), print_matq(gso)));
        if *self.held.borrow() != want {
            self.install_key(a, basis, gso);
        }
    }

    /// B independent `samp_p` calls with one key (row b of `targets` = one syndrome).
    pub fn samp_p_batch(&self, a: &MatZq, td: &(MatZ, MatQ), targets: &MatZq) -> MatZ {
        let (n, m) = self.dims();
        self.ensure_key(a, &td.0, &td.1);
        let b = targets.get_num_rows();
        assert_eq!(targets.get_num_columns(), n);
        let u = matzq_to_rows(targets);
        let mut e = vec![0i64; (b * m) as usize];
        check(unsafe { ffi::psfgpv_samp_p(self.handle, next_seed(&self.seed, &self.calls), 0, b as usize, u.as_ptr(), e.as_mut_ptr()) }, "psfgpv_samp_p");
        matz_from_rows(b, m, &e)
    }

    /// `samp_p_batch` without waiting (`psfgpv_samp_p_async`): at most two batches in flight per handle, the rows of batch i cross PCIe while batch i + 1
    /// walks.  `PendingBatch::into_matz` (or dropping the batch) waits.
    pub fn samp_p_batch_async<'a>(&'a self, a: &MatZq, td: &(MatZ, MatQ), targets: &MatZq) -> PendingBatch<'a> {
        let (n, m) = self.dims();
        self.ensure_key(a, &td.0, &td.1);
        let b = targets.get_num_rows();
        assert_eq!(targets.get_num_columns(), n);
        let u = matzq_to_rows(targets);
        let mut e = vec![0i64; (b * m) as usize];
        let owner = Owner::Gpv(self);
        let ticket = owner.next_ticket();
        check(unsafe { ffi::psfgpv_samp_p_async(self.handle, next_seed(&self.seed, &self.calls), 0, b as usize, u.as_ptr(), e.as_mut_ptr()) }, "psfgpv_samp_p_async");
        PendingBatch { owner, ticket, rows: b, cols: m, _u: u, e, waited: Cell::new(false) }
    }

    /// every asynchronous batch of this handle has completed (`psfgpv_wait`)
    pub fn wait_batches(&self) {
        check(unsafe { ffi::psfgpv_wait(self.handle) }, "psfgpv_wait");
    }
}

impl Drop for GpuPSFGPV {
    fn drop(&mut self) {
        unsafe { ffi::psfgpv_destroy(self.handle) }
    }
}

impl PSF for GpuPSFGPV {
    type A = MatZq;
    type Trapdoor = (MatZ, MatQ);
    type Domain = MatZ;
    type Range = MatZq;

    /// gpv.rs:83-94
    fn trap_gen(&self) -> (MatZq, (MatZ, MatQ)) {
        let (n, m) = self.dims();
        check(unsafe { ffi::psfgpv_trap_gen(self.handle, next_seed(&self.seed, &self.calls)) }, "psfgpv_trap_gen");
        let mut a = vec![0u64; (n * m) as usize];
        let mut bt = vec![0i32; (m * m) as usize];
        let mut gt = vec![0f64; (m * m) as usize];
        check(unsafe { ffi::psfgpv_export_key(self.handle, a.as_mut_ptr(), std::ptr::null_mut(), bt.as_mut_ptr(), gt.as_mut_ptr()) }, "psfgpv_export_key");
        let a_mat = matzq_from_rows(n, m, &self.params.gp.q, &a);
        // rows of the ABI's matrices are the reference's columns
        let mut basis = MatZ::new(m, m);
        for i in 0..m {
            for j in 0..m {
                basis.set_entry(j, i, Z::from(bt[(i * m + j) as usize] as i64)).unwrap();
            }
        }
        let gso = matq_from_rows(m, m, &gt, true);
        *self.held.borrow_mut() = Some((print_matzq(&a_mat), print_basis(&basis), print_matq(&gso)));
        (a_mat, (basis, gso))
    }

    /// gpv.rs:113-116
    fn samp_d(&self) -> MatZ {
        let (_, m) = self.dims();
        let mut e = vec![0i64; m as usize];
        check(unsafe { ffi::psfgpv_samp_d(self.handle, next_seed(&self.seed, &self.calls), 0, 1, e.as_mut_ptr()) }, "psfgpv_samp_d");
        matz_from_rows(m, 1, &e)
    }

    /// gpv.rs:152-161
    fn samp_p(&self, a: &MatZq, td: &(MatZ, MatQ), u: &MatZq) -> MatZ {
        let (n, m) = self.dims();
        self.ensure_key(a, &td.0, &td.1);
        assert!(u.get_num_rows() == n && u.get_num_columns() == 1);
        let uv = matzq_to_rows(u);
        let mut e = vec![0i64; m as usize];
        check(unsafe { ffi::psfgpv_samp_p(self.handle, next_seed(&self.seed, &self.calls), 0, 1, uv.as_ptr(), e.as_mut_ptr()) }, "psfgpv_samp_p");
        matz_from_rows(m, 1, &e)
    }

    /// gpv.rs:190-193
    fn f_a(&self, a: &MatZq, sigma: &MatZ) -> MatZq {
        let (n, m) = self.dims();
        assert!(sigma.get_num_rows() == m && sigma.get_num_columns() == 1);
        let pa = print_matzq(a);
        let held = *self.held.borrow();
        match held {
            Some((ha, _, _)) if ha == pa => {}
            Some((_, hb, hg)) => {
                // another public matrix on a handle that holds a trapdoor: the basis and its GSO stay (read back from the device), A is replaced
                let (mut bt, mut gt) = (vec![0i32; (m * m) as usize], vec![0f64; (m * m) as usize]);
                check(unsafe { ffi::psfgpv_export_key(self.handle, std::ptr::null_mut(), std::ptr::null_mut(), bt.as_mut_ptr(), gt.as_mut_ptr()) }, "psfgpv_export_key");
                let av = matzq_to_rows(a);
                *self.held.borrow_mut() = None;
                check(unsafe { ffi::psfgpv_load_key(self.handle, av.as_ptr(), bt.as_ptr(), gt.as_ptr()) }, "psfgpv_load_key");
                *self.held.borrow_mut() = Some((pa, hb, hg));
            }
            None => panic!("f_a before trap_gen / samp_p: the handle holds no key"),
        }
        let e = matz_to_rows_i64(sigma);
        let mut u = vec![0u64; n as usize];
        check(unsafe { ffi::psfgpv_f_a(self.handle, 1, e.as_ptr(), u.as_mut_ptr()) }, "psfgpv_f_a");
        matzq_from_rows(n, 1, &self.params.gp.q, &u)
    }

    /// gpv.rs:219-224
    fn check_domain(&self, sigma: &MatZ) -> bool {
        if !sigma.is_column_vector() {
            return false;
        }
        let e = matz_to_rows_i64(sigma);
        let mut ok = [0u8; 1];
        check(unsafe { ffi::psfgpv_check_domain(self.handle, 1, e.as_ptr(), e.len(), ok.as_mut_ptr()) }, "psfgpv_check_domain");
        ok[0] != 0
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// PSFGPVRing (gpv_ring.rs:62-67, impl PSF :69-284)
// ------------------------------------------------------------------------------------------------------------------------

pub struct GpuPSFGPVRing {
    pub params: PSFGPVRing,
    handle: *mut ffi::psfring_handle,
    seed: Cell<u64>,
    calls: Cell<u64>,
    installed: RefCell<Option<(MatPolynomialRingZq, MatPolyOverZ, MatPolyOverZ)>>,
}

impl GpuPSFGPVRing {
    pub fn new(params: PSFGPVRing, device: i32, seed: u64) -> Self {
        let c = ffi::psfring_params { gp: gp_ring_to_c(&params.gp), s: f64::from(&params.s), s_td: f64::from(&params.s_td), device, flags: 0 };
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::psfring_create(&c, &mut handle) }, "psfring_create");
        Self { params, handle, seed: Cell::new(seed), calls: Cell::new(0), installed: RefCell::new(None) }
    }

    pub fn reseed(&self, seed: u64) {
        self.seed.set(seed);
        self.calls.set(0);
    }

    /// (ring degree n, gadget length k)
    fn dims(&self) -> (i64, i64) {
        let gp = gp_ring_to_c(&self.params.gp);
        (gp.n as i64, gp.k as i64)
    }

    fn ensure_key(&self, a: &MatPolynomialRingZq, r: &MatPolyOverZ, e: &MatPolyOverZ) {
        if let Some((ia, ir, ie)) = self.installed.borrow().as_ref() {
            if ia == a && ir == r && ie == e {
                return;
            }
        }
        let (n, _) = self.dims();
        let (av, rv, ev) = (matpolyring_to_rows(a, n), matpoly_to_rows(r, n), matpoly_to_rows(e, n));
        check(unsafe { ffi::psfring_load_key(self.handle, av.as_ptr(), rv.as_ptr(), ev.as_ptr()) }, "psfring_load_key");
        *self.installed.borrow_mut() = Some((a.clone(), r.clone(), e.clone()));
    }
}

impl Drop for GpuPSFGPVRing {
    fn drop(&mut self) {
        unsafe { ffi::psfring_destroy(self.handle) }
    }
}

impl PSF for GpuPSFGPVRing {
    type A = MatPolynomialRingZq;
    type Trapdoor = (MatPolyOverZ, MatPolyOverZ);
    type Domain = MatPolyOverZ;
    type Range = MatPolynomialRingZq;

    /// gpv_ring.rs:91-98
    fn trap_gen(&self) -> (MatPolynomialRingZq, (MatPolyOverZ, MatPolyOverZ)) {
        let (n, k) = self.dims();
        check(unsafe { ffi::psfring_trap_gen(self.handle, next_seed(&self.seed, &self.calls)) }, "psfring_trap_gen");
        let mut a = vec![0u64; ((k + 2) * n) as usize];
        let mut r = vec![0i64; (k * n) as usize];
        let mut e = vec![0i64; (k * n) as usize];
        check(unsafe { ffi::psfring_export_key(self.handle, a.as_mut_ptr(), r.as_mut_ptr(), e.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) }, "psfring_export_key");
        let a_i64: Vec<i64> = a.iter().map(|&v| v as i64).collect();
        let a_poly = matpoly_from_rows(1, k + 2, n, &a_i64);
        let a_mat = MatPolynomialRingZq::from((&a_poly, &self.params.gp.modulus));
        let (r_mat, e_mat) = (matpoly_from_rows(1, k, n, &r), matpoly_from_rows(1, k, n, &e));
        *self.installed.borrow_mut() = Some((a_mat.clone(), r_mat.clone(), e_mat.clone()));
        (a_mat, (r_mat, e_mat))
    }

    /// gpv_ring.rs:118-122
    fn samp_d(&self) -> MatPolyOverZ {
        let (n, k) = self.dims();
        let mut sigma = vec![0i64; ((k + 2) * n) as usize];
        check(unsafe { ffi::psfring_samp_d(self.handle, next_seed(&self.seed, &self.calls), 0, 1, sigma.as_mut_ptr()) }, "psfring_samp_d");
        matpoly_from_rows(k + 2, 1, n, &sigma)
    }

    /// gpv_ring.rs:160-212 (short basis, elimination and Gram-Schmidt vectors are built once per key inside the library)
    fn samp_p(&self, a: &MatPolynomialRingZq, td: &(MatPolyOverZ, MatPolyOverZ), u: &MatPolynomialRingZq) -> MatPolyOverZ {
        let (n, k) = self.dims();
        self.ensure_key(a, &td.0, &td.1);
        let uv = matpolyring_to_rows(u, n);
        assert_eq!(uv.len() as i64, n, "u is one element of R_q");
        let mut sigma = vec![0i64; ((k + 2) * n) as usize];
        check(unsafe { ffi::psfring_samp_p(self.handle, next_seed(&self.seed, &self.calls), 0, 1, uv.as_ptr(), sigma.as_mut_ptr()) }, "psfring_samp_p");
        matpoly_from_rows(k + 2, 1, n, &sigma)
    }

    /// gpv_ring.rs:243-247
    fn f_a(&self, a: &MatPolynomialRingZq, sigma: &MatPolyOverZ) -> MatPolynomialRingZq {
        let (n, _) = self.dims();
        match self.installed.borrow().as_ref() {
            Some((ia, _, _)) if ia == a => {}
            Some((_, ir, ie)) => {
                let (av, rv, ev) = (matpolyring_to_rows(a, n), matpoly_to_rows(ir, n), matpoly_to_rows(ie, n));
                check(unsafe { ffi::psfring_load_key(self.handle, av.as_ptr(), rv.as_ptr(), ev.as_ptr()) }, "psfring_load_key");
            }
            None => panic!("f_a before trap_gen / samp_p: the handle holds no key"),
        }
        let s = matpoly_to_rows(sigma, n);
        let mut u = vec![0u64; n as usize];
        check(unsafe { ffi::psfring_f_a(self.handle, 1, s.as_ptr(), u.as_mut_ptr()) }, "psfring_f_a");
        let u_i64: Vec<i64> = u.iter().map(|&v| v as i64).collect();
        MatPolynomialRingZq::from((&matpoly_from_rows(1, 1, n, &u_i64), &self.params.gp.modulus))
    }

    /// gpv_ring.rs:274-283
    fn check_domain(&self, sigma: &MatPolyOverZ) -> bool {
        let (n, _) = self.dims();
        if sigma.get_num_columns() != 1 {
            return false;
        }
        let s = matpoly_to_rows(sigma, n);
        let mut ok = [0u8; 1];
        check(unsafe { ffi::psfring_check_domain(self.handle, 1, s.as_ptr(), s.len(), ok.as_mut_ptr()) }, "psfring_check_domain");
        ok[0] != 0
    }
}

#[cfg(test)]
mod test_drop_in {
    //! The reference's own PSF tests (mp_perturbation.rs:433-448, gpv.rs:254-268, gpv_ring.rs:318-334), pointed at the GPU types.
    use super::*;

    #[test]
    fn perturbation_samp_p_is_a_preimage() {
        for (n, q) in [(5, 256), (6, 128)] {
            let psf = GpuPSFPerturbation::new(
                PSFPerturbation { gp: GadgetParameters::init_default(n, q), r: Q::from(n).log(2).unwrap(), s: Q::from(25) },
                0,
                1,
            );
            let (a, td) = psf.trap_gen();
            let domain_sample = psf.samp_d();
            let range_fa = psf.f_a(&a, &domain_sample);
            let preimage = psf.samp_p(&a, &td, &range_fa);
            assert_eq!(range_fa, psf.f_a(&a, &preimage));
            assert!(psf.check_domain(&preimage));
        }
    }

    #[test]
    fn gpv_samp_p_is_a_preimage() {
        let psf = GpuPSFGPV::new(PSFGPV { gp: GadgetParameters::init_default(5, 256), s: Q::from(10) }, 0, 2);
        let (a, td) = psf.trap_gen();
        let range_fa = psf.f_a(&a, &psf.samp_d());
        let preimage = psf.samp_p(&a, &td, &range_fa);
        assert_eq!(range_fa, psf.f_a(&a, &preimage));
        assert!(psf.check_domain(&preimage));
    }

    #[test]
    fn ring_samp_p_is_a_preimage() {
        let psf = GpuPSFGPVRing::new(
            PSFGPVRing { gp: GadgetParametersRing::init_default(8, 512), s: Q::from(100), s_td: Q::from(1.005_f64) },
            0,
            3,
        );
        let (a, td) = psf.trap_gen();
        let range_fa = psf.f_a(&a, &psf.samp_d());
        let preimage = psf.samp_p(&a, &td, &range_fa);
        assert_eq!(range_fa, psf.f_a(&a, &preimage));
        assert!(psf.check_domain(&preimage));
    }
}
