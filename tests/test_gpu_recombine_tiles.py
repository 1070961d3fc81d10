"""The recombination e = p + [R; I] z (mp_perturbation.rs:328-335) picks its kernel by shape and, on the device, by the digit planes in use: 256 x 256 tiles
(k_recombine_mfma_big) for one plane and batches that are multiples of 256 with m_bar >= 512, the 128 x 128 kernel otherwise or when a second plane is in
use.  Every route against the oracle, whole batch, with m_bar set by hand (a public field of GadgetParameters, gadget_parameters.rs:44-52) so that the shapes
stay small: partial tiles along m_bar, batch sizes on both sides of the dispatch, and a wide gadget Gaussian that raises the second plane."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # (n, k, m_bar, base, q, r, s, B, expect_hi)
    (8, 10, 512, 2, 1024, 3.0, 400.0, 256, False),      # big tiles, exactly two row tiles
    (8, 10, 700, 2, 1024, 3.0, 450.0, 512, False),      # big tiles, m_bar not a multiple of 256 (rows of R beyond m_bar are zero, stores are masked)
    (8, 10, 700, 2, 1024, 3.0, 450.0, 384, False),      # batch not a multiple of 256: the 128 x 128 kernel
    (8, 2, 512, 32, 1024, 6.0, 12000.0, 256, True),      # big tiles are launched, see the second plane and leave the call to the 128 x 128 kernel
    (8, 10, 300, 2, 1024, 3.0, 300.0, 256, False),      # m_bar below the big-tile threshold
    # 5 ... 64 preimages: k_recombine_wg (64 x 64 tiles over all of K through an LDS-DMA ring; K a multiple of 128), one to four fragments of 16 preimages
    (8, 10, 512, 2, 1024, 3.0, 400.0, 5, False),
    (8, 10, 700, 2, 1024, 3.0, 450.0, 16, False),        # m_bar = 10 tiles of 64 + 60 rows
    (8, 10, 700, 2, 1024, 3.0, 450.0, 40, False),        # three fragments on eight waves
    (8, 10, 300, 2, 1024, 3.0, 300.0, 64, False),
    (24, 16, 1000, 2, 65536, 3.0, 700.0, 33, False),     # K = 384: three slots, fewer than the ring holds
    (64, 2, 512, 32, 1024, 6.0, 12000.0, 48, True),      # the second plane: a second pass over the ring
    (64, 2, 512, 32, 1024, 6.0, 12000.0, 9, True),
    (8, 10, 700, 2, 1024, 3.0, 450.0, 100, False),       # 65 ... 128 preimages: two column groups of 64, the second one ragged
    (64, 2, 512, 32, 1024, 6.0, 12000.0, 128, True),     # ... with the second plane
    (8, 10, 300, 2, 1024, 3.0, 300.0, 65, False),
    (8, 10, 700, 2, 1024, 3.0, 450.0, 300, False),       # five column groups, the last one of 44 preimages
    (8, 10, 700, 2, 1024, 3.0, 450.0, 500, False),       # beyond 448: the 256 x 256 tiles with a ragged last tile
    (64, 2, 512, 32, 1024, 6.0, 12000.0, 600, True),     # ... and the second plane: they leave the call to the 128 x 128 kernel
    (8, 10, 700, 2, 1024, 3.0, 450.0, 1000, False),
    (64, 2, 512, 32, 1024, 6.0, 12000.0, 448, True),
    (8, 2, 512, 32, 1024, 6.0, 12000.0, 16, True),       # K = 64 is not a multiple of 128: the 128 x 128 kernel with K splits
]


@pytest.mark.parametrize("n,k,m_bar,base,q,r,s,B,expect_hi", CASES)
def test_every_recombination_route_matches_the_oracle(oracle, n, k, m_bar, base, q, r, s, B, expect_hi):
    import tools_amd as T
    gp = T.GadgetParameters(n, k, m_bar, base, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = psf.trap_gen(11)
    orc = oracle.PSFPerturbation(oracle.GadgetParams(n, k, m_bar, base, q), r, s)
    orc.load_key(A, R, Lp)
    u = oracle.uniform_targets(9, B, n, q)
    e = psf.samp_p(u, seed=31, first_index=5)
    assert psf.last_status() == 0
    e_ref = orc.samp_p(31, u, first_index=5)
    assert (e == e_ref).all()
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all()
    st = psf.samp_p_stages(u, seed=31, first_index=5)
    assert (st["e"] == e_ref).all()
    assert (np.abs(st["z"]).max() > 127) == expect_hi                    # the wide case really uses the second plane, the others do not
    psf.close()
