"""Row f4: the t.in writer (sac_amd/make_tin.py, recipe of make_tin_zeroT.jl) against a t.in the reference itself wrote
from bins it holds: tests/golden/tin_bdp = sac/process_G/synthetic/bosonic_delta_peak/{cor.dat (as exact integers), tgrid.dat,
t.in}.  Everything that does not depend on the bootstrap draws must agree exactly; the bootstrap columns (the reference's
draws are unseeded) within the bootstrap error, which the test derives from N_boot."""
import os

import numpy as np

from sac_amd.make_tin import make_tin
from sac_amd.tin import read_tin

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tin_bdp")


def test_tin_written_from_the_references_bins_matches_the_references_tin(tmp_path):
    bins = np.load(f"{D}/bins.npz")["q"] / 1e8
    tau_grid = np.loadtxt(f"{D}/tgrid.dat")
    assert bins.shape == (8000, 23) and (bins[:, 0] == 1.0).all() is not None
    ref = read_tin(f"{D}/t_ref.in")
    out = tmp_path / "t.in"
    td = make_tin(bins, tau_grid, beta=128.0, n_boot=10000, seed=7, out_file=str(out))
    back = read_tin(str(out))                                   # the file round-trips through the sampler's own reader
    n_boot = 10000
    # deterministic part: beta, N_tau (no point cut), N_boot, norm, the tau grid, the normalised means
    assert (td.beta, td.n_tau, td.n_boot) == (ref.beta, ref.n_tau, ref.n_boot) == (128.0, 22, n_boot)
    assert abs(td.norm / ref.norm - 1) < 1e-13
    np.testing.assert_array_equal(td.tau, ref.tau)
    np.testing.assert_allclose(td.G_raw, ref.G_raw, rtol=1e-12)
    for a, b in ((back.G_raw, td.G_raw), (back.sigma, td.sigma), (back.cov, td.cov)):
        np.testing.assert_array_equal(a, b)
    # bootstrap error bars: an estimate of a standard deviation from N_boot samples scatters by 1 / sqrt(2 N_boot) = 0.7 %
    # on either side -> 5 sigma of the difference of two such estimates
    tol = 5 * np.sqrt(2) / np.sqrt(2 * n_boot)
    np.testing.assert_allclose(td.sigma_bar, ref.sigma_bar, rtol=tol)
    # the covariance the eigen-decomposition encodes, C = U diag(N_boot sigma~^2) U^T: each element is a bootstrap estimate
    # with relative error ~ sqrt(2 / N_boot); compared as a whole and through its spectrum
    C = lambda t: (t.cov * (n_boot * t.sigma ** 2)[None, :]) @ t.cov.T          # noqa: E731
    Ct, Cr = C(td), C(ref)
    assert np.linalg.norm(Ct - Cr) / np.linalg.norm(Cr) < 5 * np.sqrt(2.0 / n_boot)
    np.testing.assert_allclose(td.sigma[-6:], ref.sigma[-6:], rtol=5 * np.sqrt(2) / np.sqrt(2 * n_boot))   # well separated
    np.testing.assert_allclose(td.sigma, ref.sigma, rtol=0.15)                  # the small ones mix with their neighbours
    assert (np.diff(td.sigma) >= 0).all() and (np.diff(ref.sigma) >= 0).all()   # ascending, as Julia's eigen returns them
    # eigenvectors: orthonormal; the well-separated ones coincide up to sign
    assert np.abs(td.cov.T @ td.cov - np.eye(22)).max() < 1e-12
    ov = np.abs(np.sum(td.cov[:, -4:] * ref.cov[:, -4:], axis=0))
    assert (ov > 0.99).all(), ov
    # what the sampler sees: chi^2 of a test vector in either eigenbasis
    rng = np.random.default_rng(1)
    x = ref.G_raw * (1 + 1e-5 * rng.standard_normal(22))
    chi2 = lambda t: float((((t.cov.T @ (x - t.G_raw)) / t.sigma) ** 2).sum())  # noqa: E731
    assert abs(chi2(td) / chi2(ref) - 1) < 0.15


def test_relative_error_cutoff_drops_tau_points():
    rng = np.random.default_rng(0)
    tau = np.array([0.0, 0.5, 1.0, 1.5])
    mean = np.array([1.0, 0.5, 0.1, 1e-4])
    bins = mean + rng.standard_normal((400, 4)) * np.array([0.0, 1e-3, 1e-3, 2e-3])   # last point: relative error >> 0.2
    td = make_tin(bins, tau, beta=8.0, n_boot=500, seed=1)
    assert td.n_tau == 2 and np.array_equal(td.tau, [0.5, 1.0])              # mt:80-83
    assert td.cov.shape == (2, 2) and td.norm == 1.0
