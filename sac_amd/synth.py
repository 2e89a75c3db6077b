"""Seeded synthetic G(tau) problems in the t.in contract (BASELINE.md section 3 workload).

Recipe: true spectrum = delta peak + Gaussian continuum -> G(tau) through the bosonic kernel
-> N_b bins with relative, exponentially tau-correlated noise
(sac/process_G/synthetic/generate_synth.jl:657-687) -> bootstrap, covariance, eigenbasis
(sac/process_G/make_tin_zeroT.jl:92-166).  Own numpy code; only the recipe is the reference's.
"""
import numpy as np

from .tin import TimeData


def kernel_values(kind, omega, tau, beta):
    """K(omega, tau) of sac_free.jl:220-239; omega (n,), tau (n_tau,) -> (n, n_tau)."""
    w = np.asarray(omega, dtype=np.float64)[:, None]
    t = np.asarray(tau, dtype=np.float64)[None, :]
    if kind == "finiteT":
        num, den = -w * t, -beta * w
        m = np.maximum(np.maximum(num, den), 0.0)
        return np.exp(num - m) / (np.exp(-m) + np.exp(den - m))
    if kind == "zeroT":
        return np.exp(-w * t)
    if kind == "bosonic":
        return (np.exp(-w * t) + np.exp(-w * (beta - t))) / (1.0 + np.exp(-beta * w))
    raise ValueError("Invalid Kernel type.")


def synthetic_timedata(n_tau=64, beta=128.0, seed=12345, n_bins=1000, n_boot=10000,
                       rel_noise=1e-5, xi=1.0, peak_omega=1.0, peak_weight=0.7,
                       cont_center=3.0, cont_width=0.8, tau_max=8.0, kernel_type="bosonic"):
    """Returns a TimeData like read_G! would after make_tin: bosonic delta peak + Gaussian continuum.

    The tau grid is quadratic on (0, tau_max] (dense at small tau, cf. the reference's
    synthetic/bosonic_delta_peak/tgrid.dat: 0.1 ... 7.9 non-uniform)."""
    rng = np.random.default_rng(seed)
    i = np.arange(1, n_tau + 1, dtype=np.float64)
    tau = np.round(tau_max * (0.25 * i / n_tau + 0.75 * (i / n_tau) ** 2), 8)
    # true spectrum on a fine grid (bosonic: S(w) at w >= 0, kernel carries the mirror image)
    w = np.linspace(0.0, 12.0, 24001)
    cont = np.exp(-0.5 * ((w - cont_center) / cont_width) ** 2)
    cont *= (1.0 - peak_weight) / np.trapezoid(cont, w)
    K = kernel_values(kernel_type, w, tau, beta)
    G0 = peak_weight * kernel_values(kernel_type, [peak_omega], tau, beta)[0] + np.trapezoid(cont[:, None] * K, w, axis=0)
    G_zero = 1.0      # normalised spectrum: bosonic K(w, 0) = 1, so G(0) = 1
    # correlated relative noise, generate_synth.jl:672-687
    sig = rel_noise * (2.0 * G_zero)
    tt = np.concatenate([[0.0], tau])
    C = np.exp(-np.abs(tt[:, None] - tt[None, :]) / xi)
    C = C / np.sqrt((C ** 2).sum(axis=1))[:, None]
    noise = (rng.standard_normal((n_bins, n_tau + 1)) * sig) @ C.T
    bins = np.concatenate([[G_zero], G0])[None, :] + noise          # (n_bins, n_tau+1)
    # bootstrap: make_tin_zeroT.jl:92-118 (multinomial counts = N_b draws with replacement)
    counts = rng.multinomial(n_bins, np.full(n_bins, 1.0 / n_bins), size=n_boot).astype(np.float64)
    samples = np.vstack([bins.mean(axis=0)[None, :], counts @ bins / n_bins])
    norm = samples[0, 0]
    samples = samples / samples[:, :1]
    mean, boots = samples[0, 1:], samples[1:, 1:]
    dev = boots - mean[None, :]
    sigma_bar = np.sqrt((dev ** 2).sum(axis=0) / n_boot)            # make_tin_zeroT.jl:131
    cov = dev.T @ dev                                               # compute_cov!, 156-166
    evals, evecs = np.linalg.eigh(cov)
    evecs = evecs / np.linalg.norm(evecs, axis=0)[None, :]
    eig_sigma = np.sqrt(np.maximum(evals, 1e-300) / n_boot)         # 146
    return TimeData(beta, n_boot, norm, tau, mean, sigma_bar, eig_sigma, evecs)


# --------------------------------------------------------------------------------------------
# Row f4: the reference's synthetic-data recipe restated (sac/process_G/synthetic/generate_synth.jl,
# "gs:NNN"): G(tau) of a tabulated spectrum, then bins with tau-correlated noise.
# --------------------------------------------------------------------------------------------

def make_G_tau(omega, S, tau, beta, kernel_type="bosonic", peaks=()):
    """make_G_tau (gs:534-654) for a spectrum given as a table: G0(tau_j) = (1 / pi) [ int S(w) K(w, tau_j) dw +
    sum_p A_p K(w_p, tau_j) ] (gs:638: G0_val / pi; delta peaks enter as A0 K(w0, tau), gs:596-599).  The reference integrates
    its analytic spectrum with QuadGK at rtol 1e-13; here the table it wrote (aw.dat, spacing 4e-4) is integrated by the
    trapezoid rule -- error ~ 1e-8 relative, the rounding of cor.dat.  tau includes tau = 0 (gs:585: j = 0..N_tau)."""
    omega, S, tau = np.asarray(omega, dtype=np.float64), np.asarray(S, dtype=np.float64), np.asarray(tau, dtype=np.float64)
    K = kernel_values(kernel_type, omega, tau, beta)                    # [len(omega), len(tau)]
    G = np.trapezoid(S[:, None] * K, omega, axis=0)
    for w_p, A_p in peaks:
        G = G + A_p * kernel_values(kernel_type, [w_p], tau, beta)[0]
    return G / np.pi


def noise_mixing(tau, xi):
    """The matrix of add_noise (gs:672-685): row j = exp(-|tau_i - tau_j| / xi) / sqrt(sum_i exp(...)^2)."""
    tau = np.asarray(tau, dtype=np.float64)
    C = np.exp(-np.abs(tau[:, None] - tau[None, :]) / xi)
    return C / np.sqrt((C ** 2).sum(axis=1))[:, None]


def add_noise(G0, tau, rel_sigma, xi, rng, n_bins=1):
    """add_noise (gs:657-687), default branch: sigma = rel_sigma * 2 G0[tau = 0] (gs:669); white noise of that size per tau
    point, mixed with the exponential tau correlation; every point keeps variance sigma^2.  Returns [n_bins, len(tau)]."""
    G0 = np.asarray(G0, dtype=np.float64)
    sig = rel_sigma * (G0[0] * 2)
    white = rng.standard_normal((n_bins, len(G0))) * sig
    return G0[None, :] + white @ noise_mixing(tau, xi).T


def noise_covariance(G0, tau, rel_sigma, xi):
    """Covariance of the bins add_noise draws: sigma^2 M M^T with M = noise_mixing."""
    M = noise_mixing(tau, xi)
    return (rel_sigma * 2 * G0[0]) ** 2 * (M @ M.T)
