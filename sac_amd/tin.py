"""`t.in` input contract (written by the reference's make_tin*.jl, read by read_G!).

    line 1          : beta  N_tau  N_boot  norm
    lines 2..N_tau+1: tau_i  G_i  sigma_i  sigma~_i = sqrt(eigval_i / N_boot)
    N_tau blocks    : a line with the block index j, then N_tau lines = eigenvector j

Mirrors read_G! of sac/free/sac_free.jl:99-131 and the writer make_tin_zeroT.jl:143-153.
"""
import numpy as np


class TimeData:
    """What read_G! leaves on the SAC struct (free:125-129)."""

    def __init__(self, beta, n_boot, norm, tau, G_raw, sigma_bar, sigma, cov):
        self.beta, self.n_boot, self.norm = float(beta), float(n_boot), float(norm)
        self.tau = np.asarray(tau, dtype=np.float64)
        self.n_tau = len(self.tau)
        self.G_raw = np.asarray(G_raw, dtype=np.float64)
        self.sigma_bar = np.asarray(sigma_bar, dtype=np.float64)
        self.sigma = np.asarray(sigma, dtype=np.float64)        # column 4 (free:109)
        self.cov = np.asarray(cov, dtype=np.float64)            # eigenvectors as columns
        self.sigma_inv = 1.0 / self.sigma                       # free:110
        self.G = self.cov.T @ self.G_raw                        # free:123
        half = self.tau <= (self.beta // 2)                     # free:119-120
        self.omega_window = float(np.log(1.0 / self.G_raw[half][-1]) / self.tau[half][-1])  # free:121


def read_tin(path):
    """read_G! (free:99-131).  Accepts both on-disk dialects (Julia writedlm, Fortran fixed width)."""
    with open(path) as f:
        rows = [ln.split() for ln in f if ln.strip()]
    beta, n_tau, n_boot, norm = (float(x) for x in rows[0][:4])
    n_tau = int(n_tau)
    body = np.array([[float(x) for x in r[:4]] for r in rows[1:n_tau + 1]])
    cov = np.zeros((n_tau, n_tau))
    start = n_tau + 1
    for j in range(n_tau):
        head = rows[start + j * (n_tau + 1)]
        if int(float(head[0])) != j + 1:                        # free:115
            raise ValueError(f"{path}: eigenvector block {j + 1} is headed {head[0]!r}")
        blk = rows[start + j * (n_tau + 1) + 1: start + (j + 1) * (n_tau + 1)]
        cov[:, j] = [float(r[0]) for r in blk]
    return TimeData(beta, n_boot, norm, body[:, 0], body[:, 1], body[:, 2], body[:, 3], cov)


def write_tin(path, beta, n_boot, norm, tau, G, sigma_bar, eig_sigma, eigvecs):
    """make_tin_zeroT.jl:143-153."""
    n = len(tau)
    with open(path, "w") as f:
        f.write(f"{float(beta)!r} {float(n)!r} {float(n_boot)!r} {float(norm)!r}\n")
        for i in range(n):
            f.write(f"{float(tau[i])!r} {float(G[i])!r} {float(sigma_bar[i])!r} {float(eig_sigma[i])!r}\n")
        for j in range(n):
            f.write(f"{j + 1}\n")
            for v in eigvecs[:, j]:
                f.write(f"{float(v)!r}\n")
