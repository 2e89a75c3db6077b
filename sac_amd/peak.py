"""Host side of the `peak` parameterisation: mirror of sac/peak/sac_peak.jl above the C ABI.

"peak:NNN" = /root/reference/sac/peak/sac_peak.jl line NNN.  Everything per-move runs in
libsac_hip.so (SAC_KIND_PEAK); this module builds what the reference builds once on the host
(read_G! additions, initialize!: the four delta groups, amplitudes, Kp/Kn tables).
"""
import numpy as np

from .lib import KIND_PEAK, SacHandle
from .synth import kernel_values


def peak_window_and_ratio(td, symm, kernel_type):
    """What peak's read_G! adds to free's (peak:117-127): omega_window from tau <= beta/4 and the
    guess of the +/- peak weight ratio."""
    q = td.tau <= (td.beta // 4)
    window = float(np.log(1.0 / td.G_raw[q][-1]) / td.tau[q][-1])
    if symm == 1 or kernel_type == "bosonic" or td.tau.max() <= td.beta // 2:
        ratio = np.inf
    else:
        ratio = td.G_raw[0] / td.G_raw[-1]
    return window, ratio


class PeakProblem:
    """What initialize! (peak:149-285) leaves on the SAC struct."""

    def __init__(self, td, n_omega, n_p, A_0, omega_m, d_omega, omega_0=0.0, fix_edge=0, symm=0,
                 kernel_type="finiteT", approx_ratio=None):
        if kernel_type not in ("finiteT", "zeroT", "bosonic"):
            raise ValueError("Invalid Kernel type.")                 # peak:245
        if kernel_type == "bosonic":
            symm = 0                                                 # peak:1460-1461
        self.td, self.symm, self.kernel_type, self.fix_edge = td, symm, kernel_type, int(fix_edge)
        self.n_omega, self.n_p, self.A_0, self.d_omega = int(n_omega), int(n_p), float(A_0), float(d_omega)
        omega_window, ratio = peak_window_and_ratio(td, symm, kernel_type)
        if approx_ratio is not None:
            ratio = approx_ratio
        self.approx_ratio = ratio
        self.wi_0 = int(np.floor(omega_0 / d_omega))                 # peak:151-152
        self.wi_m = int(np.ceil(omega_m / d_omega))
        self.window = omega_window / d_omega                         # peak:154
        self.n_grid = self.wi_m + 1
        self.sample_neg = not (symm == 1 or kernel_type == "bosonic")
        N, P = self.n_omega, self.n_p
        self.groups = dict(peak_p=slice(0, P), cont_p=slice(P, P + N), peak_n=slice(P + N, 2 * P + N),
                           cont_n=slice(2 * P + N, 2 * P + 2 * N))   # peak:166-169
        g = self.groups
        wi = np.zeros(2 * (N + P), dtype=np.int64)
        A = np.zeros(2 * (N + P))
        inv_r = 0.0 if np.isinf(ratio) else 1.0 / ratio
        A0_p = A_0 / (1 + inv_r)                                     # peak:172-173
        A0_n = 0.0 if np.isinf(ratio) else A_0 / (1 + ratio)
        start = int(np.floor(self.wi_0)) if self.fix_edge == 1 else int(np.floor(self.window))   # peak:178-184
        wi[g["peak_p"]] = start
        wi[g["peak_n"]] = start
        A[g["peak_p"]] = A0_p / P                                    # peak:186-187
        A[g["peak_n"]] = A0_n / P
        i = np.arange(1, N + 1)
        spread = np.floor((self.window / N) * i).astype(np.int64)    # peak:194-199
        wi[g["cont_p"]] = int(np.floor(wi[g["peak_p"]].max() * 1.0)) + spread
        wi[g["cont_n"]] = int(np.floor(wi[g["peak_n"]].max() * 1.0)) + spread
        Ac_p = (1 - A_0) / (1 + inv_r)                               # peak:207-208
        Ac_n = 0.0 if np.isinf(ratio) else (1 - A_0) / (1 + ratio)
        A[g["cont_p"]] = (i / i.sum()) * Ac_p                        # peak:210-211
        A[g["cont_n"]] = (i / i.sum()) * Ac_n
        if not self.sample_neg:                                      # peak:214-227
            A[g["peak_p"]] = A_0 / P
            A[g["cont_p"]] = (1 - A_0) / N
            A[g["peak_n"]] = 0
            A[g["cont_n"]] = 0
            if symm == 1:
                A = A / (2 * A.sum())
            wi[g["peak_n"]] = 0
            wi[g["cont_n"]] = 0
        else:
            A = A / A.sum()                                          # peak:229
        self.wi, self.A = wi, A
        self.Kp, self.Kn = self._kernel_tables()

    def _kernel_tables(self):
        """peak:248-276: half-offset grid omega = (i + 1/2) d_omega, rotated into the eigenbasis."""
        td = self.td
        Kp = np.zeros((self.n_grid, td.n_tau))
        Kn = np.zeros((self.n_grid, td.n_tau)) if self.sample_neg else None
        for a in range(0, self.n_grid, 65536):
            w = (np.arange(a, min(a + 65536, self.n_grid)) + 0.5) * self.d_omega     # peak:253
            kv = kernel_values(self.kernel_type, w, td.tau, td.beta)
            if self.symm == 1:
                kv = kv + kernel_values(self.kernel_type, -w, td.tau, td.beta)       # peak:258-262
            Kp[a:a + len(w)] = kv @ td.cov
            if self.sample_neg:
                Kn[a:a + len(w)] = kernel_values(self.kernel_type, -w, td.tau, td.beta) @ td.cov   # peak:270-272
        return Kp, Kn

    def make_handle(self, n_chains=1, seed=0, **kw):
        """SAC(...) + initialize! + calc_Gbar/calc_chi2 + default windows (peak:1482-1507)."""
        h = SacHandle(self.td.n_tau, self.n_omega, self.n_grid, self.wi_0, self.wi_m, n_chains=n_chains,
                      symm=self.symm, seed=seed, kind=KIND_PEAK, n_peak=self.n_p, fix_edge=self.fix_edge,
                      bosonic=int(self.kernel_type == "bosonic"), **kw)
        h.upload_problem_peak(self.Kp, self.Kn, self.td.G, self.td.sigma_inv)
        h.set_state(self.wi, self.A, np.full(11, self.window / 10))  # peak:1507
        h.init_gbar()
        return h
