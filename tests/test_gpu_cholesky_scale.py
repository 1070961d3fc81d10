"""The device Cholesky factor sqrt(Sigma_2) (mp_perturbation.rs:111-139) at BASELINE's headline size -- 241 panels of 128 --
checked WITHOUT trusting any other device stage:

  * residual: for random probe vectors y, L (L^t y) must equal Sigma_2 y with Sigma_2 y evaluated from R alone,
    Sigma_2 = (r^2 / 2 pi) ((s^2 - 1) I - (b^2 + 1) T T^t), T = [R; I]  (:116-135);
  * rows 0..2047 of the factor (16 panels) against the oracle's unblocked Cholesky-Banachiewicz recurrence of the leading
    block (the leading principal block of a Cholesky factor is the factor of the leading block);
  * end to end: the per-coordinate variance of 4096 full-size preimages, panel by panel -- a wrong factor anywhere (a bad
    trailing update at panel 37, say) changes the covariance of the perturbation and with it the variance of e there.
A e = u and check_domain cannot see any of this: they hold for every perturbation."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, Q, R_PAR, S_PAR = 512, 2**30, 9.0, 512.0        # BASELINE.json configs[2] (bench.py "c3")


@pytest.fixture(scope="module")
def c3():
    import tools_amd as T
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(N, Q), R_PAR, S_PAR)
    psf.trap_gen(3, export=False)
    yield psf
    psf.close()


def sigma2_times(R, y, r, s, base=2):
    """Sigma_2 y from R alone: c ((s^2 - 1) y - (b^2 + 1) T (T^t y)), T = [R; I]."""
    mb, w = R.shape
    c = (1.0 / (2 * math.pi)) * (r * r)
    t = y[mb:].copy()                               # T^t y = R^t y_top + y_bot
    for i0 in range(0, mb, 1024):
        i1 = min(i0 + 1024, mb)
        t += R[i0:i1].astype(np.float64).T @ y[i0:i1]
    Tt = np.empty_like(y)
    for i0 in range(0, mb, 1024):
        i1 = min(i0 + 1024, mb)
        Tt[i0:i1] = R[i0:i1].astype(np.float64) @ t
    Tt[mb:] = t
    return c * ((s * s - 1.0) * y - (base * base + 1) * Tt)


@pytest.mark.timeout(1500)
def test_factor_reproduces_sigma2_on_probe_vectors(c3):
    psf = c3
    m = psf.m
    _, R = psf.export_A_R()
    rng = np.random.default_rng(5)
    y = rng.standard_normal((m, 8))
    # every panel is probed on its own as well: a unit-ish vector supported on the LAST panel sees only the last rows of L
    y[:, 7] = 0.0
    y[m - 100:, 7] = rng.standard_normal(100)
    blocks = []
    step = 1024
    w = np.zeros_like(y)                             # w = L^t y
    for row0 in range(0, m, step):
        nr = min(step, m - row0)
        packed = psf.export_sqrt_sigma2_rows(row0, nr)
        blk = np.zeros((nr, row0 + nr))
        off = 0
        for r in range(nr):
            ln = row0 + r + 1
            blk[r, :ln] = packed[off:off + ln]
            off += ln
        assert np.isfinite(blk).all()
        blocks.append(blk)
        w[:row0 + nr] += blk.T @ y[row0:row0 + nr]
    v = np.zeros_like(y)                             # v = L w
    for bi, blk in enumerate(blocks):
        row0 = bi * step
        v[row0:row0 + blk.shape[0]] = blk @ w[:blk.shape[1]]
    ref = sigma2_times(R, y, R_PAR, S_PAR)
    for c in range(y.shape[1]):
        err = np.abs(v[:, c] - ref[:, c]).max() / np.abs(ref[:, c]).max()
        assert err < 1e-10, (c, err)
    # per panel of 128 rows, relative to that panel's own magnitude (dense probes only: their entries have one scale everywhere)
    for p0 in range(0, m, 128):
        sl = slice(p0, min(p0 + 128, m))
        err = np.abs(v[sl, :7] - ref[sl, :7]).max() / np.abs(ref[sl, :7]).max()
        assert err < 1e-9, (p0, err)


@pytest.mark.timeout(1500)
def test_leading_rows_match_the_unblocked_recurrence(c3, oracle):
    psf = c3
    m0 = 2048                                        # 16 device panels
    _, R = psf.export_A_R()
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(N, Q), R_PAR, S_PAR, with_L=False)
    rc, Lref = orc.sqrt_sigma_2_leading(R, S_PAR, m0)
    assert rc == 0
    Ldev = psf.export_sqrt_sigma2_rows(0, m0)
    scale = np.abs(Lref).max()
    np.testing.assert_allclose(Ldev, Lref, rtol=0, atol=1e-9 * scale)
    # and a few hundred rows bit-for-bit close in relative terms on the diagonal (the entries that carry sqrt)
    diag = np.array([i * (i + 1) // 2 + i for i in range(m0)])
    np.testing.assert_allclose(Ldev[diag], Lref[diag], rtol=1e-12, atol=0)


@pytest.mark.timeout(1500)
def test_full_size_preimages_have_the_right_variance_in_every_panel(c3):
    import torch
    psf = c3
    B, m = 4096, psf.m
    dev = torch.device("cuda:0")
    # device memory and copies only -- the moments are taken on the host, so the test does not wait on torch's own kernels being paged in on a fresh box
    u1 = torch.empty((1, psf.n), dtype=torch.int64, device=dev)
    psf.uniform_targets_dev(u1.data_ptr(), 1, seed=9)
    u = torch.from_numpy(np.repeat(u1.cpu().numpy(), B, axis=0)).to(dev)   # one fixed syndrome: every row is a draw from the same coset Gaussian
    e = torch.empty((B, m), dtype=torch.int64, device=dev)
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=2024)
    torch.cuda.synchronize()
    assert psf.last_status() == 0
    sigma2 = (S_PAR * R_PAR) ** 2 / (2 * math.pi)
    eh = e.cpu().numpy()
    del e
    var = np.empty(m)
    mean = np.empty(m)
    for c0 in range(0, m, 4096):                      # column blocks: no 1 GB float64 copy of e
        x = eh[:, c0:c0 + 4096].astype(np.float64)
        mean[c0:c0 + 4096] = x.mean(axis=0)
        var[c0:c0 + 4096] = x.var(axis=0, ddof=1)
    ratio = var / sigma2
    # single coordinates: relative std of a variance estimate from B draws is sqrt(2/B) = 2.2 %; 30801 coordinates -> 4.5 sigma tail
    assert np.abs(ratio - 1).max() < 0.12, (ratio.min(), ratio.max())
    assert np.abs(mean).max() < 6 * math.sqrt(sigma2 / B) + 1.0
    # panels of 128 coordinates: 2.2 % / sqrt(128) = 0.2 % each
    npan = (m + 127) // 128
    pan = np.array([ratio[p * 128:(p + 1) * 128].mean() for p in range(npan)])
    assert np.abs(pan - 1).max() < 0.012, (int(np.abs(pan - 1).argmax()), pan.min(), pan.max())
    # correlations between coordinates of different panels, incl. the last one, stay at the 1/sqrt(B) noise level
    idx = [0, 127, 128, 4095, 15440, 15441, 20000, m - 129, m - 128, m - 1]
    sub = eh[:, idx].astype(np.float64)
    corr = np.corrcoef(sub.T)
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() < 5.5 / math.sqrt(B), np.abs(corr).max()


@pytest.mark.parametrize("n,q", [(8, 64), (20, 257), (40, 2**20)])
def test_the_three_cholesky_forms_agree(monkeypatch, exp_lib, oracle, n, q):
    """PSF_CHOL selects the factorisation: on the key's chunk stream (the default for large keys: no dense matrix), left-looking on a dense matrix with
    the LDS-staged GEMM (the default below 16 GB), right-looking (rounds 1-2).  Same key seed: A and R are bitwise equal, the factors agree with each
    other and with the oracle's unblocked recurrence within rounding, and each reproduces Sigma_2; m runs from one partial panel to several panels."""
    import tools_amd as T
    r, s = 4.0, 120.0
    gp = T.GadgetParameters.init_default(n, q)
    got = {}
    for form in ("stream", "gemm", "right", "hybrid"):                 # "hybrid" = no switch: the default below 16 GB (Sigma_2 dense at once, factorisation on the chunk stream)
        if form == "hybrid":
            monkeypatch.delenv("PSF_CHOL", raising=False)
        else:
            monkeypatch.setenv("PSF_CHOL", form)
        psf = T.PSFPerturbation(gp, r, s)
        A, (R, Lp, _) = psf.trap_gen(6)
        got[form] = (A, R, Lp)
        psf.close()
    A, R, L0 = got["stream"]
    for form in ("gemm", "right", "hybrid"):
        assert (got[form][0] == A).all() and (got[form][1] == R).all()
        np.testing.assert_allclose(got[form][2], L0, rtol=0, atol=1e-10 * np.abs(L0).max())
    assert (got["hybrid"][2] == L0).all()                             # the hybrid copies its panels out of the same integers: the stream form's factor bit for bit
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    rc, Lref = orc.compute_sqrt_sigma_2(R, s)
    assert rc == 0
    np.testing.assert_allclose(L0, Lref, rtol=0, atol=1e-10 * np.abs(Lref).max())
