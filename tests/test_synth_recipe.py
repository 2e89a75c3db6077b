"""Row f4, the synthetic-data recipe (sac/process_G/synthetic/generate_synth.jl:534-697) restated in sac_amd/synth.py, against
what the reference itself generated: tests/golden/tin_bdp = sac/process_G/synthetic/bosonic_delta_peak/{aw.dat (the true
spectrum), tgrid.dat, cor.dat (8000 bins)} (tools/make_golden_tin.py).  The reference's noise draws are unseeded, so the bins
are compared through their statistics: the bin mean against the restated G0(tau) within the standard error, the bin covariance
against the covariance add_noise's mixing implies within the sampling error of a covariance estimate."""
import os

import numpy as np

from sac_amd.synth import add_noise, make_G_tau, noise_covariance

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tin_bdp")
REL_SIGMA, XI, BETA = 1e-3, 1.0, 128.0      # the run's settings: sigma = 10^-3 (cor.dat's scatter), xi = 1 (run_synth.jl:26)


def _load():
    aw = np.load(f"{D}/aw.npz")
    omega, S = aw["omega"], aw["S"]
    # rows 0-1: the delta peak at omega_0 = 1 as a spike of height A0 / spacing (A0 = 0.7); the rest: the continuum
    assert omega[0] == omega[1] == 1.0 and S[0] == 0.0
    A0 = S[1] * (omega[2] - omega[1])
    tau = np.loadtxt(f"{D}/tgrid.dat")
    bins = np.load(f"{D}/bins.npz")["q"] / 1e8
    return omega[2:], S[2:], A0, tau, bins


def test_bin_mean_is_the_restated_G_of_the_references_spectrum():
    omega, S, A0, tau, bins = _load()
    assert abs(A0 - 0.7) < 1e-12 and tau[0] == 0.0 and bins.shape == (8000, 23)
    # the continuum starts one grid step above the peak: the missing sliver [1, 1.0004] carries S(1.0004) * 4e-4
    omega = np.concatenate([[1.0], omega])
    S = np.concatenate([[S[0]], S])
    G0 = make_G_tau(omega, S, tau, BETA, "bosonic", peaks=[(1.0, A0)])
    assert abs(G0[0] * np.pi - 1.0) < 2e-4                      # normalised spectrum, K(omega, 0) = 1: G0(0) = 1 / pi
    sigma = REL_SIGMA * 2 * G0[0]
    err = sigma / np.sqrt(len(bins))                            # standard error of a bin mean (every tau point has variance sigma^2)
    z = (bins.mean(axis=0) - G0) / err
    print("z", np.round(z, 2))
    # tau-correlated noise: the 23 z values are not independent draws; |z| < 4 each and an rms of order one
    assert np.abs(z).max() < 4.0, z
    assert np.sqrt((z ** 2).mean()) < 2.0, z
    # the scatter itself: sigma per tau point (plus the 1e-8 rounding of cor.dat, var 8e-18)
    sd = bins.std(axis=0, ddof=1)
    np.testing.assert_allclose(sd, np.sqrt(sigma ** 2 + 1e-16 / 12), rtol=0.05)


def test_bin_covariance_is_add_noises_mixing():
    omega, S, A0, tau, bins = _load()
    G0 = bins.mean(axis=0)
    want = noise_covariance(G0, tau, REL_SIGMA, XI)
    got = np.cov(bins.T)
    # an element of a covariance estimate from N samples scatters by sqrt((C_ii C_jj + C_ij^2) / N)
    n = len(bins)
    tol = 5 * np.sqrt((np.outer(np.diag(want), np.diag(want)) + want ** 2) / n)
    assert (np.abs(got - want) < tol + 2e-17).all(), np.abs(got - want).max()
    # neighbouring tau points are strongly correlated (xi = 1 against spacings of 0.1 ... 1.1), distant ones are not
    corr = got / np.sqrt(np.outer(np.diag(got), np.diag(got)))
    assert corr[1, 2] > 0.9 and abs(corr[1, 22]) < 0.05


def test_own_draws_follow_the_same_statistics():
    omega, S, A0, tau, bins = _load()
    G0 = make_G_tau(np.concatenate([[1.0], omega]), np.concatenate([[S[0]], S]), tau, BETA, "bosonic", peaks=[(1.0, A0)])
    mine = add_noise(G0, tau, REL_SIGMA, XI, np.random.default_rng(3), n_bins=8000)
    assert mine.shape == bins.shape
    sigma = REL_SIGMA * 2 * G0[0]
    np.testing.assert_allclose(mine.std(axis=0, ddof=1), sigma, rtol=0.05)
    cm, cr = np.cov(mine.T), np.cov(bins.T)
    assert np.abs(cm - cr).max() < 8 * np.sqrt(2.0 / 8000) * sigma ** 2
