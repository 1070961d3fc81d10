"""End-to-end distribution checks of samp_p on the GPU (the reference has none).

MP12's perturbation method must output a SPHERICAL discrete Gaussian of parameter s*r over the coset
{e : A e = u}: every coordinate has standard deviation s r / sqrt(2 pi) and distinct coordinates are uncorrelated --
which only holds if Sigma_2, its Cholesky factor, the perturbation, the gadget sampler (parameter r sqrt(b^2+1)) and
the recombination e = p + [R; I] z all fit together (mp_perturbation.rs:111-139, :304-336)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_perturbation_preimages_are_spherical():
    import tools_amd as T
    n, q, r, s, B = 8, 64, 3.0, 25.0, 24000
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    A, (R, _, _) = psf.trap_gen(3)
    u = np.tile(np.array([[5, 60, 0, 17, 33, 1, 2, 63]], dtype=np.uint64), (B, 1))     # one fixed syndrome
    e = psf.samp_p(u, seed=11).astype(np.float64)
    sigma = s * r / math.sqrt(2 * math.pi)
    std = e.std(axis=0)
    assert np.abs(std / sigma - 1).max() < 0.05, (std.min(), std.max(), sigma)
    assert np.abs(e.mean(axis=0)).max() < 5 * sigma / math.sqrt(B) + 1.0
    # the two halves of e (the A_bar part that receives R z, and the gadget part) must not differ in scale
    mb = psf.m_bar
    assert abs(std[:mb].mean() / std[mb:].mean() - 1) < 0.01
    corr = np.corrcoef(e.T)
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() < 6 / math.sqrt(B)
    # squared norm concentrates around m sigma^2
    nrm2 = (e**2).sum(axis=1)
    assert abs(nrm2.mean() / (psf.m * sigma**2) - 1) < 0.02


def test_perturbation_preimages_are_spherical_across_cholesky_panels():
    """The same property at a size whose sqrt(Sigma_2) spans five 128-row panels (m = 537), so the blocked device Cholesky
    (diagonal block, TRSM, MFMA SYRK) and the multi-row-block triangular product are inside the loop: a wrong trailing update
    shows up as a wrong variance / correlation of e in the later panels."""
    import tools_amd as T
    n, q, r, s, B = 32, 256, 5.0, 120.0, 30000
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    psf.trap_gen(5, export=False)
    assert psf.m == 537
    u = np.tile((np.arange(n, dtype=np.uint64) * 37 + 11) % q, (B, 1))
    e = psf.samp_p(u, seed=13).astype(np.float64)
    assert (psf.f_a(e.astype(np.int64)) == u).all()
    sigma = s * r / math.sqrt(2 * math.pi)
    std = e.std(axis=0)
    assert np.abs(std / sigma - 1).max() < 0.03, (std.min(), std.max(), sigma)     # 1/sqrt(2B) = 0.4 %
    for p0 in range(0, psf.m, 128):                                                # each Cholesky panel on its own
        pan = (std[p0:p0 + 128] ** 2).mean() / sigma**2
        assert abs(pan - 1) < 0.006, (p0, pan)
    assert np.abs(e.mean(axis=0)).max() < 5.5 * sigma / math.sqrt(B) + 1.0
    corr = np.corrcoef(e.T)
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() < 6.2 / math.sqrt(B), np.abs(corr).max()             # 144k pairs: 4.7 sigma tail
    nrm2 = (e**2).sum(axis=1)
    assert abs(nrm2.mean() / (psf.m * sigma**2) - 1) < 0.005


def test_gpv_preimages_have_the_right_scale():
    import tools_amd as T
    n, q, s, B = 6, 128, 60.0, 12000      # s above the smoothing parameter of the short basis so the output is ~ spherical
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    psf.trap_gen(2, export=False)
    u = np.tile(np.array([[1, 100, 7, 64, 0, 127]], dtype=np.uint64), (B, 1))
    e = psf.samp_p(u, seed=5).astype(np.float64)
    sigma = s / math.sqrt(2 * math.pi)
    std = e.std(axis=0)
    assert np.abs(std / sigma - 1).max() < 0.06, (std.min(), std.max(), sigma)
    corr = np.corrcoef(e.T)
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() < 6 / math.sqrt(B)


def test_ring_preimages_have_the_right_scale():
    """PSFGPVRing (gpv_ring.rs:160-212) at a small degree: one fixed syndrome, many preimages; the coefficient embedding of the preimage is a
    spherical Gaussian of parameter s over its coset (s = 4 x compute_s is above the smoothing parameter of the embedded short basis)."""
    import tools_amd as T
    n, q, B = 8, 257, 12000
    s = 4 * ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4                   # 4 x compute_s(n), gpv_ring.rs:296-298
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
    psf.trap_gen(6)
    u = np.tile(np.array([[1, 200, 7, 64, 0, 127, 256, 33]], dtype=np.uint64), (B, 1))
    sg = psf.samp_p(u, seed=5)
    assert (psf.f_a(sg) == u).all() and psf.check_domain(sg).all()
    e = sg.reshape(B, -1).astype(np.float64)
    sigma = s / math.sqrt(2 * math.pi)
    std = e.std(axis=0)
    assert np.abs(std / sigma - 1).max() < 0.05, (std.min(), std.max(), sigma)          # 1 / sqrt(2 B) = 0.65 %, d coordinates
    assert abs((std**2).mean() / sigma**2 - 1) < 0.006
    assert np.abs(e.mean(axis=0)).max() < 5.5 * sigma / math.sqrt(B) + 1.0
    corr = np.corrcoef(e.T)
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() < 6 / math.sqrt(B), np.abs(corr).max()


def test_samp_d_marginals():
    import tools_amd as T
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(5, 32), 2.0, 20.0)
    e = psf.samp_d(seed=1, B=20000).astype(np.float64)        # D_{Z^m, s r}, mp_perturbation.rs:264-267
    sigma = 40.0 / math.sqrt(2 * math.pi)
    assert np.abs(e.std(axis=0) / sigma - 1).max() < 0.04
    assert np.abs(e).max() <= 6 * 40.0 + 1


def test_empty_and_ragged_batches():
    import tools_amd as T
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(5, 32), 2.0, 25.0)
    psf.trap_gen(1)
    assert psf.samp_p(np.zeros((0, 5), dtype=np.uint64)).shape == (0, psf.m)
    assert psf.samp_d(B=0).shape == (0, psf.m)
    u = np.arange(5 * 129, dtype=np.uint64).reshape(129, 5) % 32        # 129 = one column block + 1
    e = psf.samp_p(u, seed=9)
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all()
    for b in (0, 127, 128):                                              # rows do not depend on the batch they ride in
        assert (psf.samp_p(u[b], seed=9, first_index=b) == e[b]).all()
    # index beyond 2^32 selects a different (but still valid) stream
    e_hi = psf.samp_p(u[:3], seed=9, first_index=2**33)
    assert (psf.f_a(e_hi) == u[:3]).all() and not (e_hi == e[:3]).all()


def test_two_handles_in_two_threads_do_not_interfere():
    """One handle per thread is the rule (as for the reference's PSF objects); two of them working at the same time must give
    what each gives alone."""
    import threading
    import numpy as np
    import tools_amd as T
    from oracle import oracle as O
    cfgs = [(8, 64, 3.0, 25.0, 300), (6, 128, float(np.log2(6)), 25.0, 257)]
    handles, targets, alone = [], [], []
    for n, q, r, s, B in cfgs:
        psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
        psf.trap_gen(4)
        u = O.uniform_targets(3, B, n, q)
        handles.append(psf); targets.append(u)
        alone.append([psf.samp_p(u, seed=70 + i) for i in range(6)])
    together = [[None] * 6 for _ in cfgs]

    def work(k):
        for i in range(6):
            together[k][i] = handles[k].samp_p(targets[k], seed=70 + i)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(cfgs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(len(cfgs)):
        for i in range(6):
            assert (together[k][i] == alone[k][i]).all()


def test_host_pointer_path_slices_large_batches_without_changing_a_bit(monkeypatch, exp_lib):
    """psfp_samp_p cuts big batches into slices so that PCIe overlaps compute; PSF_HOST_SLICE forces slicing at a small size."""
    import numpy as np
    import tools_amd as T
    from oracle import oracle as O
    n, q = 8, 64
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), 3.0, 25.0)
    psf.trap_gen(2)
    u = O.uniform_targets(4, 1000, n, q)
    monkeypatch.setenv("PSF_HOST_SLICE", "100000")
    whole = psf.samp_p(u, seed=21, first_index=7)
    monkeypatch.setenv("PSF_HOST_SLICE", "256")          # 256, 256, 256, 232
    sliced = psf.samp_p(u, seed=21, first_index=7)
    assert (whole == sliced).all()
    big = O.uniform_targets(5, 3300, n, q)               # default policy: two halves from 3072 rows on
    monkeypatch.delenv("PSF_HOST_SLICE")
    a = psf.samp_p(big, seed=22)
    monkeypatch.setenv("PSF_HOST_SLICE", "100000")
    b = psf.samp_p(big, seed=22)
    assert (a == b).all() and (psf.f_a(a) == big).all()
