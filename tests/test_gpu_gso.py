"""The blocked Gram-Schmidt of the device (psf_gemm_kernels.hpp: FP64-MFMA GEMMs, panels of 128, re-orthogonalised; MatQ::gso of gpv.rs:88-91)
through its own ABI entry psf_gso_rows, on bases the key generators never produce: every panel-edge shape, rectangular inputs, and rows that
are nearly dependent (Gram-Schmidt vectors thousands of times shorter than their basis vectors -- what broke the round-2 loop at C2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(bt, gt, oracle, tol=1e-9):
    ref = oracle.gso_rows_leading(bt)
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(gt, ref, rtol=0, atol=tol * scale)
    nrm = np.sqrt((gt * gt).sum(axis=1))
    G = (gt @ gt.T) / np.outer(nrm, nrm)
    np.fill_diagonal(G, 0.0)
    assert np.abs(G).max() < 1e-11, np.abs(G).max()
    return nrm


@pytest.mark.parametrize("rows,width", [(1, 1), (3, 7), (23, 23), (127, 127), (128, 128), (129, 129), (130, 257), (300, 300), (385, 512), (700, 700)])
def test_random_bases_of_every_panel_shape(oracle, rows, width):
    from tools_amd import gadget
    rng = np.random.default_rng(rows * 1000 + width)
    bt = rng.integers(-50, 51, size=(rows, width)).astype(np.int32)
    bt[np.arange(rows), np.arange(rows)] += 400            # comfortably independent
    gt = gadget.gso_rows(bt)
    _check(bt, gt, oracle)


def test_nearly_dependent_rows_keep_their_orthogonality(oracle):
    """Rows that are small multiples of an earlier row plus a tiny vector, in runs inside and across panels: the Gram-Schmidt vectors of those rows are ~ 1000 times
    shorter than the rows themselves (|b| / |b~| of C2's short basis is ~ 4000)."""
    from tools_amd import gadget
    rng = np.random.default_rng(7)
    d = 520
    bt = rng.integers(-40, 41, size=(d, d)).astype(np.int64)
    bt[np.arange(d), np.arange(d)] += 300
    for run in (range(5, 40), range(120, 140), range(250, 262), (383, 384, 385), (511,), (519,)):
        anchor = run[0] - 1                                   # every row of the run is a small multiple of the row in front of the run, plus a tiny vector
        for j in run:
            small = np.zeros(d, dtype=np.int64)
            small[rng.integers(0, d, 3)] = rng.integers(-1, 2, 3)
            small[j] += 1
            bt[j] = (2 + j % 3) * bt[anchor] + small
    assert np.abs(bt).max() < 2**31
    bt = bt.astype(np.int32)
    gt = gadget.gso_rows(bt)
    nrm = _check(bt, gt, oracle, tol=1e-9)
    ratio = np.sqrt((bt.astype(np.float64) ** 2).sum(axis=1)) / nrm
    assert ratio.max() > 500, ratio.max()                   # the case really is ill-conditioned


def test_dependent_rows_are_reported(oracle):
    from tools_amd import gadget, PsfError
    bt = np.arange(36, dtype=np.int32).reshape(6, 6)         # rank 2
    with pytest.raises(PsfError):
        gadget.gso_rows(bt)


def test_gso_is_reproducible_bit_for_bit():
    """every rank regenerates the key from the seed: the split-K partial sums are added in a fixed order, no atomics"""
    from tools_amd import gadget
    rng = np.random.default_rng(5)
    bt = rng.integers(-99, 100, size=(400, 400)).astype(np.int32)
    bt[np.arange(400), np.arange(400)] += 500
    a, b = gadget.gso_rows(bt), gadget.gso_rows(bt)
    assert (a == b).all()
