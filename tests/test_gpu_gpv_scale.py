"""PSFGPV / PSFGPVRing at BASELINE's full sizes (C2: n = 256, q = 3329, d = 6208; C4: R_q of degree 256, d = 3584): what bitwise
parity with a shared key cannot see.

Every large parity test loads the DEVICE's Gram-Schmidt data into the oracle, A e = u holds for any integer z and check_domain is a
loose norm bound -- so a Gram-Schmidt pass that lost orthogonality over its thousands of dependent rank-1 updates (MatQ::gso,
gpv.rs:91; inside MatPolyOverZ::sample_d, gpv_ring.rs:205) would ship unnoticed.  This file closes that hole the way
tests/test_gpu_cholesky_scale.py closed it for the Cholesky factor:

 (a) the device's own B~ at full dimension: rows mutually orthogonal (>= 2.5e5 sampled pairs incl. (first, last)), B = M B~ with M unit
     lower triangular (mu_ji = <b_j, b~_i> / |b~_i|^2) -- together these two say B~ IS the Gram-Schmidt orthogonalisation of B -- and the
     leading 512 vectors against the oracle's per-vector chain (vector i reads only vectors <= i);
 (b) the distribution of full-size batches with one fixed syndrome: per-coordinate variance s^2 / 2 pi in every band of 128 rows, means,
     cross-band correlations at the 1 / sqrt(B) level (D_{Lambda_u^perp(A), s} is spherical above the smoothing parameter, GPV08 Thm 4.1).
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY_SEED = 3


def _build(name):
    import tools_amd as T
    if name == "c2":
        n, q, s, B = 256, 3329, 1024.0, 1024                      # BASELINE.json configs[1]
        psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
        psf.trap_gen(KEY_SEED, export=False)
        _, (bt, gt) = psf.export_key()
        d = psf.m
    elif name == "c2s240":
        n, q, s, B = 256, 3329, 240.0, 1024                       # the bench's own rule 30 log2 n (benches/psf.rs:32), SURVEY 8d's second point
        psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
        psf.trap_gen(KEY_SEED, export=False)
        _, (bt, gt) = psf.export_key()
        d = psf.m
    else:
        n, q, B = 256, 3329, 4096                                 # BASELINE.json configs[3]
        s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4          # compute_s, gpv_ring.rs:296-298
        psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
        psf.trap_gen(KEY_SEED)
        _, _, _, bt, gt = psf.export_key()
        d = psf.d
    return dict(name=name, psf=psf, bt=bt, gt=gt, d=d, n=n, q=q, s=s, B=B)


@pytest.fixture(scope="module", params=["c2", "c4"])
def keyed(request):
    k = _build(request.param)
    yield k
    k["psf"].close()


def test_gram_schmidt_rows_are_orthogonal_at_full_dimension(keyed):
    gt, d = keyed["gt"], keyed["d"]
    rng = np.random.default_rng(11)
    sel = sorted(set([0, 1, 63, 64, 127, 128, d // 2, d - 129, d - 128, d - 65, d - 64, d - 2, d - 1]) | set(rng.choice(d, 500, replace=False).tolist()))
    sub = gt[sel]
    nrm = np.sqrt((sub * sub).sum(axis=1))
    assert (nrm > 0).all() and np.isfinite(nrm).all()            # (norms far below 1 are legitimate: at C2 the shortest b~ are ~ 0.026)
    G = (sub @ sub.T) / np.outer(nrm, nrm)
    np.fill_diagonal(G, 0.0)
    assert G.shape[0] ** 2 - G.shape[0] >= 250_000
    worst = np.abs(G).max()
    assert worst < 1e-10, (worst, np.unravel_index(np.abs(G).argmax(), G.shape))
    first_last = abs(float(gt[0] @ gt[d - 1])) / (np.linalg.norm(gt[0]) * np.linalg.norm(gt[d - 1]))
    assert first_last < 1e-10, first_last
    # the last 64 vectors (the first block the walk samples, and the one that saw every update) against ALL vectors
    tail = gt[d - 64:]
    nt = np.sqrt((tail * tail).sum(axis=1))
    nall = np.sqrt((gt * gt).sum(axis=1))
    Gt = (tail @ gt.T) / np.outer(nt, nall)
    for r in range(64):
        Gt[r, d - 64 + r] = 0.0
    assert np.abs(Gt).max() < 1e-10, np.abs(Gt).max()


def test_basis_is_unit_lower_triangular_times_gram_schmidt(keyed):
    bt, gt, d = keyed["bt"], keyed["gt"], keyed["d"]
    rng = np.random.default_rng(12)
    rows = sorted(set([0, 1, 63, 64, 65, 127, 128, d // 2, d - 65, d - 64, d - 2, d - 1]) | set(rng.choice(d, 52, replace=False).tolist()))
    norm2 = (gt * gt).sum(axis=1)
    mu = (bt[rows].astype(np.float64) @ gt.T) / norm2          # mu[r][i] = <b_row, b~_i> / |b~_i|^2
    for r, j in enumerate(rows):
        mu[r, j] = 1.0
        mu[r, j + 1:] = 0.0
    recon = mu @ gt
    scale = float(np.abs(bt).max())
    err = np.abs(recon - bt[rows]).max()
    assert err < 1e-9 * scale, (err, scale)
    # and nothing of b_j lives beyond its own index: <b_j, b~_i> = 0 for i > j
    above = (bt[rows].astype(np.float64) @ gt.T) / np.sqrt(norm2)
    for r, j in enumerate(rows):
        above[r, :j + 1] = 0.0
    rown = np.sqrt((bt[rows].astype(np.float64) ** 2).sum(axis=1))
    assert (np.abs(above) / rown[:, None]).max() < 1e-10


def test_leading_gram_schmidt_vectors_equal_the_oracle_chain(keyed, oracle):
    bt, gt = keyed["bt"], keyed["gt"]
    lead = 512
    ref = oracle.gso_rows_leading(bt[:lead])
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(gt[:lead], ref, rtol=0, atol=1e-9 * scale)
    n2d, n2r = (gt[:lead] ** 2).sum(axis=1), (ref ** 2).sum(axis=1)
    np.testing.assert_allclose(n2d, n2r, rtol=1e-11, atol=0)


def _full_size_distribution(k):
    import torch
    psf, d, n, B, s = k["psf"], k["d"], k["n"], k["B"], k["s"]
    dev = torch.device("cuda:0")
    u = torch.empty((B, n), dtype=torch.int64, device=dev)
    e = torch.empty((B, d), dtype=torch.int64, device=dev)
    psf.uniform_targets_dev(u.data_ptr(), 1, seed=9)
    u[1:] = u[0]                                               # one fixed syndrome: every row is a draw from the same coset Gaussian
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=2025)
    torch.cuda.synchronize()
    assert psf.last_status() == 0
    u2 = torch.empty_like(u)
    ok = torch.empty((B,), dtype=torch.uint8, device=dev)
    psf.f_a_dev(e.data_ptr(), u2.data_ptr(), ok.data_ptr(), B)
    torch.cuda.synchronize()
    assert bool((u2 == u).all().item()) and bool(ok.all().item())
    x = e.to(torch.float64)
    sigma2 = s * s / (2 * math.pi)
    ratio = (x.var(dim=0, unbiased=True) / sigma2).cpu().numpy()
    mean = x.mean(dim=0).cpu().numpy()
    return e, ratio, mean, sigma2


def _check_distribution(k, e, ratio, mean, sigma2):
    d, B = k["d"], k["B"]
    # single coordinates: relative std of a variance estimate from B draws is sqrt(2 / B); d coordinates -> a 4.5 sigma tail
    one = math.sqrt(2.0 / B)
    assert np.abs(ratio - 1).max() < 6.0 * one, (ratio.min(), ratio.max())
    assert np.abs(mean).max() < 6 * math.sqrt(sigma2 / B) + 1.0
    # bands of 128 rows (two blocks of the walk): a wrong update of the rows below a block would show as a band off by a constant
    nb = (d + 127) // 128
    band = np.array([ratio[p * 128:(p + 1) * 128].mean() for p in range(nb)])
    assert np.abs(band - 1).max() < 5.5 * one / math.sqrt(128), (int(np.abs(band - 1).argmax()), band.min(), band.max())
    assert abs(ratio.mean() - 1) < 5.0 * one / math.sqrt(d)
    # correlations between coordinates of different blocks / panels, incl. first and last, stay at the 1 / sqrt(B) noise level
    idx = [0, 63, 64, 127, 128, 511, 512, d // 2, d - 513, d - 129, d - 128, d - 65, d - 64, d - 1]
    sub = e[:, idx].to(dtype=__import__("torch").float64).cpu().numpy()
    corr = np.corrcoef(sub.T)
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() < 5.5 / math.sqrt(B), np.abs(corr).max()


@pytest.mark.timeout(900)
def test_full_size_preimages_have_the_right_variance_in_every_band(keyed):
    e, ratio, mean, sigma2 = _full_size_distribution(keyed)
    _check_distribution(keyed, e, ratio, mean, sigma2)


@pytest.mark.timeout(900)
def test_c2_at_the_bench_rule_s_240_is_valid_and_still_spherical_in_the_mean():
    """SURVEY 8d's second C2 point: s = 30 log2 n = 240 (benches/psf.rs:32).  |b~|_max of the short basis is ~ 178 at C2, so s = 240 is
    BELOW the smoothing parameter of the lattice: the reference's own bench runs there, every preimage must still satisfy A e = u and
    check_domain, but the marginals need not be spherical -- only the invariants and gross scale are asserted."""
    k = _build("c2s240")
    try:
        e, ratio, mean, sigma2 = _full_size_distribution(k)          # asserts A e = u and check_domain on every row
        assert 0.5 < ratio.mean() < 1.5, ratio.mean()
        assert np.abs(mean).max() < 8 * math.sqrt(sigma2 * ratio.max() / k["B"]) + 1.0
    finally:
        k["psf"].close()
