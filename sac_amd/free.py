"""Host side of the `free` parameterisation: mirror of sac/free/sac_free.jl above the C ABI.

Names, argument meaning and order of operations follow the reference ("free:NNN" =
/root/reference/sac/free/sac_free.jl line NNN).  Everything per-move runs in libsac_hip.so;
this module builds what the reference builds once on the host (grid, initial ladder, K table).
"""
import numpy as np

from .lib import SacHandle
from .synth import kernel_values


class FreeProblem:
    """What initialize! (free:134-218) leaves on the SAC struct."""

    def __init__(self, td, n_omega, omega_0, omega_m, d_omega, par=1, symm=0, kernel_type="bosonic"):
        if kernel_type not in ("finiteT", "zeroT", "bosonic"):
            raise ValueError("Invalid Kernel type.")                 # free:195
        if kernel_type == "bosonic":                                 # free:1172-1174
            omega_0, symm = 0.0, 0
        elif symm == 1:                                              # free:136-138, 1176-1178
            omega_0 = 0.0
        self.td, self.par, self.symm, self.kernel_type = td, par, symm, kernel_type
        self.n_omega, self.d_omega = int(n_omega), float(d_omega)
        self.wi_0 = int(np.floor(omega_0 / d_omega))                 # free:140
        self.wi_m = int(np.ceil(omega_m / d_omega))                  # free:141
        self.window = td.omega_window / d_omega                      # free:143
        if self.wi_0 < 0:                                            # free:148-156
            self.n_grid, start = self.wi_m - self.wi_0 + 1, self.wi_0
        else:
            self.n_grid, start = self.wi_m + 1, 0
        step = (self.wi_m - self.wi_0) // self.n_omega
        i = np.arange(1, self.n_omega + 1, dtype=np.int64)
        self.wi = start + step * i                                   # free:166
        A = np.ones(self.n_omega) if par in (1, 2) else i.astype(np.float64)   # free:168-172
        self.A = A / (2 * A.sum()) if symm == 1 else A / A.sum()     # free:176-183
        self.K = self._kernel_table()

    def _kernel_table(self):
        """free:199-213: K(omega_i, tau) rotated into the covariance eigenbasis, tau contiguous."""
        td = self.td
        grid = np.arange(self.wi_0, self.wi_m + 1, dtype=np.int64)
        K = np.zeros((self.n_grid, td.n_tau))
        for a in range(0, len(grid), 65536):
            w = grid[a:a + 65536] * self.d_omega                     # free:202
            kv = kernel_values(self.kernel_type, w, td.tau, td.beta)
            if self.symm == 1:
                kv = kv + kernel_values(self.kernel_type, -w, td.tau, td.beta)   # free:206-208
            K[a:a + len(w)] = kv @ td.cov                            # free:212 (U^T K[:, i])
        return K

    def make_handle(self, n_chains=1, seed=0, **kw):
        """SAC(...) + read_G! + initialize! + calc_Gbar/calc_chi2 + default windows (free:1191-1218)."""
        h = SacHandle(self.td.n_tau, self.n_omega, self.n_grid, self.wi_0, self.wi_m,
                      n_chains=n_chains, par=self.par, symm=self.symm, seed=seed, **kw)
        h.upload_problem(self.K, self.td.G, self.td.sigma_inv)
        h.set_state(self.wi, self.A, np.full(4, self.window))
        h.init_gbar()
        return h
