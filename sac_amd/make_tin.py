"""Data preparation: QMC bins of G(tau) -> t.in, the recipe of sac/process_G/make_tin_zeroT.jl
("mt:NNN" = line NNN of sac/process_G/synthetic/bosonic_delta_peak/make_tin_zeroT.jl; make_tin.jl differs only in
normalising with G(0) + G(beta)).

  read_data      mt:36-62    bins[N_b, N_tau + 1], column 0 = G(0)
  compute_means  mt:64-90    bootstrap error bars of the un-normalised means; tau points with G < 0 or a relative error above
                             `cutoff` are dropped
  bootstrap      mt:92-118   sample 0 = the mean of all bins, samples 1..N_boot = means of N_b bins drawn with replacement;
                             with norm=1 every sample is divided by its own G(0)
  cov_matrix     mt:122-153  mean, error bar, covariance of the normalised samples, its eigen-decomposition, the file

The reference draws with Julia's unseeded rand(1:N_b); here the draws come from a seeded numpy generator (multinomial
counts = N_b draws with replacement), so the bootstrap columns agree with a reference-written t.in within the bootstrap
error, everything else exactly (tests/test_make_tin.py)."""
import numpy as np

from .tin import TimeData, write_tin


def _bootstrap(bins, n_boot, rng, norm):
    """mt:92-118.  Returns samples[n_boot + 1, N_tau + 1] and Gnorm = the mean G(0)."""
    n_b = bins.shape[0]
    samples = np.empty((n_boot + 1, bins.shape[1]))
    samples[0] = bins.sum(axis=0)                                                # mt:96-98
    for a in range(0, n_boot, 1000):                                             # mt:101-106, in blocks to bound memory
        b = min(a + 1000, n_boot)
        counts = rng.multinomial(n_b, np.full(n_b, 1.0 / n_b), size=b - a).astype(np.float64)
        samples[1 + a:1 + b] = counts @ bins
    samples /= n_b                                                               # mt:108
    g_norm = samples[0, 0]                                                       # mt:111
    if norm:
        samples = samples / samples[:, :1]                                       # mt:112-116
    return samples, g_norm


def make_tin(bins, tau_grid, beta, n_boot=10000, seed=0, cutoff=0.2, out_file=None):
    """bins[N_b, N_tau + 1] (column 0 = G(0)), tau_grid[N_tau + 1] (entry 0 = 0) -> TimeData (and the file when out_file).
    Follows run() of mt:169-187: read_data, compute_means, cov_matrix."""
    bins = np.asarray(bins, dtype=np.float64)
    tau_grid = np.asarray(tau_grid, dtype=np.float64)
    if bins.ndim != 2 or bins.shape[1] != len(tau_grid):
        raise ValueError("make_tin: bins must be [N_b, len(tau_grid)]")
    rng = np.random.default_rng(seed)
    # compute_means (mt:64-90): which tau points survive
    s0, _ = _bootstrap(bins, n_boot, rng, norm=False)
    g_bar = s0[0]
    sig = np.sqrt(((s0[1:] - s0[0]) ** 2).sum(axis=0) / n_boot)
    with np.errstate(divide="ignore", invalid="ignore"):
        keep = ~((g_bar < 0) | ((sig / g_bar) > cutoff))                         # mt:80-83
    # cov_matrix (mt:122-153) on a fresh, normalised bootstrap
    s1, g_norm = _bootstrap(bins, n_boot, rng, norm=True)
    mean = s1[0, 1:]
    sigma = np.sqrt(((s1[1:, 1:] - mean) ** 2).sum(axis=0) / n_boot)             # mt:131
    sel = keep[1:]
    mean, sigma = mean[sel], sigma[sel]                                          # mt:134-135
    kept = s1[:, keep]                                                           # compute_cov!, mt:158: the kept rows, G(0) first
    dev = kept[1:, 1:] - kept[0, 1:]
    cov = dev.T @ dev                                                            # mt:159-164
    evals, evecs = np.linalg.eigh(cov)                                           # mt:138 (ascending, as Julia's eigen)
    evecs = evecs / np.linalg.norm(evecs, axis=0)[None, :]                       # mt:149
    eig_sigma = np.sqrt(np.maximum(evals, 0.0) / n_boot)                         # mt:146
    tau = tau_grid[keep][1:]                                                     # mt:146
    if out_file is not None:
        write_tin(out_file, float(beta), n_boot, float(g_norm), tau, mean, sigma, eig_sigma, evecs)
    return TimeData(float(beta), n_boot, float(g_norm), tau, mean, sigma, eig_sigma, evecs)
