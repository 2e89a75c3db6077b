"""Python face of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under sac_amd/ does.  The hot loops live in
oracle/sac_oracle.c (plain C); the file-format and table-construction steps of
the reference are restated here in numpy.  "free:NNN" cites
/root/reference/sac/free/sac_free.jl line NNN, "cvcalc:NNN" cites
sac/cross_validation/calc_cv.jl.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libsac_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int64)
    vp = C.c_void_p
    L.ora_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.ora_log_u01.restype = C.c_double
    L.ora_log_u01.argtypes = [C.c_double]
    L.ora_chi2.restype = C.c_double
    L.ora_chi2.argtypes = [dp, dp, dp, C.c_int]
    L.ora_free_create.restype = vp
    L.ora_free_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                  C.c_int64, dp, dp, dp]
    L.ora_free_destroy.argtypes = [vp]
    L.ora_free_set_state.argtypes = [vp, ip, dp, dp]
    L.ora_free_get_state.argtypes = [vp, ip, dp, dp]
    L.ora_free_init_gbar.argtypes = [vp]
    L.ora_free_set_gbar.argtypes = [vp, dp, C.c_double, C.c_double]
    L.ora_free_get_gbar.argtypes = [vp, dp, dp, dp]
    L.ora_free_use_philox.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64]
    L.ora_free_use_trace.argtypes = [vp, dp, C.c_int64]
    L.ora_free_trace_pos.restype = C.c_int64
    L.ora_free_trace_pos.argtypes = [vp]
    L.ora_free_sweep_counter.restype = C.c_uint64
    L.ora_free_sweep_counter.argtypes = [vp]
    L.ora_free_set_log.argtypes = [vp, vp, C.c_int64]
    L.ora_free_log_len.restype = C.c_int64
    L.ora_free_log_len.argtypes = [vp]
    L.ora_free_set_hist_mode.argtypes = [vp, C.c_int]
    L.ora_free_run_updates.argtypes = [vp, C.c_double]
    L.ora_free_adjust_windows.argtypes = [vp, C.c_int64, C.c_double]
    L.ora_free_sample.argtypes = [vp, C.c_int64, C.c_double]
    L.ora_free_get_accept_rates.argtypes = [vp, dp]
    L.ora_free_sampled_chi2.restype = C.c_double
    L.ora_free_sampled_chi2.argtypes = [vp]
    L.ora_free_get_spec.argtypes = [vp, dp]
    L.ora_free_get_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


MOVE_REC = np.dtype([("chi2_new", "<f8"), ("flags", "<i4"), ("move", "<i4")])


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().ora_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def log_u01(u):
    return lib().ora_log_u01(float(u))


def chi2(gbar, g, sinv):
    gbar = np.ascontiguousarray(gbar, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    sinv = np.ascontiguousarray(sinv, dtype=np.float64)
    return lib().ora_chi2(_dp(gbar), _dp(g), _dp(sinv), len(g))


# --------------------------------------------------------------------------
# File format and table construction (numpy restatement)
# --------------------------------------------------------------------------

def read_tin(path):
    """free:99-131 read_G! (identical copy: cvcalc:3-27).

    Returns dict(beta, n_tau, n_boot, norm, tau, G (raw), sigma, sigma_inv, cov (U),
    G_D (= U^T G), omega_window)."""
    with open(path) as f:
        rows = [ln.split() for ln in f if ln.strip()]
    beta, n_tau, n_boot, norm = (float(x) for x in rows[0][:4])
    n_tau = int(n_tau)
    body = rows[1:n_tau + 1]
    tau = np.array([float(r[0]) for r in body])
    G = np.array([float(r[1]) for r in body])
    sigma = np.array([float(r[3]) for r in body])        # column 4 (free:109)
    cov = np.zeros((n_tau, n_tau))
    start = n_tau + 1
    for j in range(n_tau):
        assert int(float(rows[start + j * (n_tau + 1)][0])) == j + 1      # free:115
        blk = rows[start + j * (n_tau + 1) + 1: start + j * (n_tau + 1) + 1 + n_tau]
        cov[:, j] = [float(r[0]) for r in blk]           # free:116
    half = tau <= (beta // 2)                            # free:119-120 (Julia `÷` on floats = floor div)
    omega_window = np.log(1.0 / G[half][-1]) / tau[half][-1]   # free:121
    G_D = cov.T @ G                                      # free:123
    return dict(beta=beta, n_tau=n_tau, n_boot=n_boot, norm=norm, tau=tau, G=G, sigma=sigma,
                sigma_inv=1.0 / sigma, cov=cov, G_D=G_D, omega_window=omega_window)


def kernel_values(kind, omega, tau, beta):
    """free:220-239.  omega: (n,), tau: (n_tau,) -> (n, n_tau)."""
    w = np.asarray(omega, dtype=np.float64)[:, None]
    t = np.asarray(tau, dtype=np.float64)[None, :]
    if kind == "finiteT":                                 # free:220-230
        num = -w * t
        den = -beta * w
        m = np.maximum(np.maximum(num, den), 0.0)
        return np.exp(num - m) / (np.exp(-m) + np.exp(den - m))
    if kind == "zeroT":                                   # free:233-235
        return np.exp(-w * t)
    if kind == "bosonic":                                 # free:237-239
        return (np.exp(-w * t) + np.exp(-w * (beta - t))) / (1.0 + np.exp(-beta * w))
    raise ValueError("Invalid Kernel type.")              # free:195


def initialize(tin, n_omega, omega_0, omega_m, d_omega, par=1, symm=0, kernel_type="bosonic"):
    """free:134-218 initialize!: grid bounds, initial delta ladder, amplitudes, K table.

    Returns dict(wi0, wim, n_grid, wi (int64[n_omega]), A, K ([n_grid, n_tau] C-order, i.e.
    tau contiguous per omega like Julia's column-major K_array[n_tau, n_grid]), window)."""
    if symm == 1:
        omega_0 = 0.0
    wi0 = int(np.floor(omega_0 / d_omega))                # free:140
    wim = int(np.ceil(omega_m / d_omega))                 # free:141
    window = tin["omega_window"] / d_omega                # free:143
    if wi0 < 0:                                           # free:148-156
        n_grid = wim - wi0 + 1
        start = wi0
    else:
        n_grid = wim + 1
        start = 0
    step = (wim - wi0) // n_omega
    i = np.arange(1, n_omega + 1, dtype=np.int64)
    wi = start + step * i                                 # free:166
    A = np.ones(n_omega) if par in (1, 2) else i.astype(np.float64)   # free:168-172
    A = A / (2 * A.sum()) if symm == 1 else A / A.sum()   # free:176-183
    grid = np.arange(wi0, wim + 1, dtype=np.int64)
    omega = grid * d_omega                                # free:202
    K = np.zeros((n_grid, tin["n_tau"]))
    chunk = 65536
    U = tin["cov"]
    for a in range(0, len(grid), chunk):
        w = omega[a:a + chunk]
        kv = kernel_values(kernel_type, w, tin["tau"], tin["beta"])
        if symm == 1:
            kv = kv + kernel_values(kernel_type, -w, tin["tau"], tin["beta"])   # free:206-208
        K[a:a + len(w)] = kv @ U                          # free:212  (U^T K[:, i]) per column
    return dict(wi0=wi0, wim=wim, n_grid=n_grid, wi=wi, A=A, K=K, window=window)


def calc_cv(tin_path, gsac_path):
    """cvcalc:29-58: chi2 of every G_SAC row against another t.in, divided by N_tau."""
    t = read_tin(tin_path)
    rows = np.loadtxt(gsac_path, delimiter=",", ndmin=2)
    thetas, gsac = rows[:, 0], rows[:, 1:]
    out = np.empty(len(thetas))
    for i in range(len(thetas)):
        g_t = t["cov"].T @ gsac[i]                        # cvcalc:37
        out[i] = chi2(g_t, t["G_D"], t["sigma_inv"])      # cvcalc:45-51
    return thetas, out / t["n_tau"]


# --------------------------------------------------------------------------
# The sampler
# --------------------------------------------------------------------------

class OracleFree:
    """One reference Markov chain (sac_free.jl SAC struct, sampler fields)."""

    def __init__(self, K, G, sigma_inv, n_omega, wi0, wim, par=1, symm=0):
        self.L = lib()
        self.K = np.ascontiguousarray(K, dtype=np.float64)      # kept alive: C borrows it
        self.n_grid, self.n_tau = self.K.shape
        self.n_omega = int(n_omega)
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        self.h = self.L.ora_free_create(par, symm, self.n_tau, self.n_omega, self.n_grid,
                                        int(wi0), int(wim), _dp(self.K), _dp(G), _dp(sigma_inv))
        if not self.h:
            raise ValueError("ora_free_create failed")
        self._log = None
        self._trace = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_free_destroy(self.h)
            self.h = None

    def set_state(self, wi=None, A=None, windows=None):
        wi = None if wi is None else np.ascontiguousarray(wi, dtype=np.int64)
        A = None if A is None else np.ascontiguousarray(A, dtype=np.float64)
        windows = None if windows is None else np.ascontiguousarray(windows, dtype=np.float64)
        self.L.ora_free_set_state(self.h, None if wi is None else _ip(wi),
                                  None if A is None else _dp(A),
                                  None if windows is None else _dp(windows))

    def get_state(self):
        wi = np.empty(self.n_omega, dtype=np.int64)
        A = np.empty(self.n_omega)
        w = np.empty(4)
        self.L.ora_free_get_state(self.h, _ip(wi), _dp(A), _dp(w))
        return wi, A, w

    def init_gbar(self):
        self.L.ora_free_init_gbar(self.h)

    def set_gbar(self, gbar, chi2, chi2_min):
        gbar = np.ascontiguousarray(gbar, dtype=np.float64)
        self.L.ora_free_set_gbar(self.h, _dp(gbar), chi2, chi2_min)

    def get_gbar(self):
        g = np.empty(self.n_tau)
        c = C.c_double()
        m = C.c_double()
        self.L.ora_free_get_gbar(self.h, _dp(g), C.byref(c), C.byref(m))
        return g, c.value, m.value

    def use_philox(self, seed, chain, sweep0=0):
        self.L.ora_free_use_philox(self.h, seed, chain, sweep0)

    def use_trace(self, trace):
        self._trace = np.ascontiguousarray(trace, dtype=np.float64)
        self.L.ora_free_use_trace(self.h, _dp(self._trace), len(self._trace))

    def trace_pos(self):
        return self.L.ora_free_trace_pos(self.h)

    def sweep_counter(self):
        return self.L.ora_free_sweep_counter(self.h)

    def enable_log(self, cap):
        self._log = np.zeros(cap, dtype=MOVE_REC)
        self.L.ora_free_set_log(self.h, self._log.ctypes.data, cap)

    def log(self):
        n = min(self.L.ora_free_log_len(self.h), len(self._log))
        return self._log[:n]

    def set_hist_mode(self, mode):
        self.L.ora_free_set_hist_mode(self.h, mode)

    def run_updates(self, theta):
        self.L.ora_free_run_updates(self.h, theta)

    def adjust_windows(self, steps, theta):
        self.L.ora_free_adjust_windows(self.h, steps, theta)

    def sample(self, steps, theta):
        self.L.ora_free_sample(self.h, steps, theta)

    def accept_rates(self):
        a = np.empty(6)
        self.L.ora_free_get_accept_rates(self.h, _dp(a))
        return a

    def sampled_chi2(self):
        return self.L.ora_free_sampled_chi2(self.h)

    def spec(self):
        s = np.empty(self.n_grid)
        self.L.ora_free_get_spec(self.h, _dp(s))
        return s

    def counters(self):
        c = (C.c_uint64 * 4)()
        self.L.ora_free_get_counters(self.h, c)
        return [int(x) for x in c]


# --------------------------------------------------------------------------
# peak parameterisation ("peak:NNN" = sac/peak/sac_peak.jl line NNN)
# --------------------------------------------------------------------------

def _peak_lib():
    L = lib()
    if getattr(L, "_peak_ready", False):
        return L
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p
    L.ora_peak_create.restype = vp
    L.ora_peak_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                  C.c_int, dp, dp, dp, dp]
    L.ora_peak_destroy.argtypes = [vp]
    L.ora_peak_set_state.argtypes = [vp, ip, dp, dp]
    L.ora_peak_get_state.argtypes = [vp, ip, dp, dp]
    L.ora_peak_init.argtypes = [vp]
    L.ora_peak_set_bounds.argtypes = [vp, dp]
    L.ora_peak_get_bounds.argtypes = [vp, dp]
    L.ora_peak_get_gbar.argtypes = [vp, dp, dp, dp]
    L.ora_peak_use_philox.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64]
    L.ora_peak_set_indiv_update.argtypes = [vp, C.c_int]
    L.ora_peak_set_log.argtypes = [vp, vp, C.c_int64]
    L.ora_peak_log_len.restype = C.c_int64
    L.ora_peak_log_len.argtypes = [vp]
    L.ora_peak_sweep_counter.restype = C.c_uint64
    L.ora_peak_sweep_counter.argtypes = [vp]
    L.ora_peak_run_updates.argtypes = [vp, C.c_double, C.c_int]
    L.ora_peak_adjust_windows.argtypes = [vp, C.c_int64, C.c_double]
    L.ora_peak_sample.argtypes = [vp, C.c_int64, C.c_double, C.c_int]
    L.ora_peak_get_accept_rates.argtypes = [vp, dp]
    L.ora_peak_sampled_chi2.restype = C.c_double
    L.ora_peak_sampled_chi2.argtypes = [vp]
    L.ora_peak_get_edge_res.argtypes = [vp, dp]
    L.ora_peak_set_edge_res.argtypes = [vp, dp]
    L.ora_peak_zero_spec.argtypes = [vp]
    L.ora_peak_get_spec.argtypes = [vp, C.c_int, dp]
    L.ora_peak_get_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    L._peak_ready = True
    return L


def read_tin_peak(path, symm=0, kernel_type="finiteT"):
    """peak:97-138 read_G!: as free's, but omega_window from tau <= beta/4 and the +/- peak weight
    ratio guess G(tau_1)/G(tau_N) (peak:117-127)."""
    t = read_tin(path)
    tau, G, beta = t["tau"], t["G"], t["beta"]
    q = tau <= (beta // 4)
    t["omega_window"] = np.log(1.0 / G[q][-1]) / tau[q][-1]          # peak:117-119
    if symm == 1 or kernel_type == "bosonic" or tau.max() <= beta // 2:
        t["approx_ratio"] = np.inf                                   # peak:122-124
    else:
        t["approx_ratio"] = G[0] / G[-1]                             # peak:126
    return t


def initialize_peak(tin, n_omega, n_p, A_0, omega_m, d_omega, omega_0=0.0, fix_edge=0, symm=0,
                    kernel_type="finiteT"):
    """peak:149-285 initialize!.  Returns dict(wi0, wim, n_grid, wi, A, Kp, Kn, window)."""
    wi0 = int(np.floor(omega_0 / d_omega))                           # peak:151
    wim = int(np.ceil(omega_m / d_omega))
    window = tin["omega_window"] / d_omega                           # peak:154
    N, P = n_omega, n_p
    wi = np.zeros(2 * (N + P), dtype=np.int64)
    A = np.zeros(2 * (N + P))
    pp, cp, pn, cn = slice(0, P), slice(P, P + N), slice(P + N, 2 * P + N), slice(2 * P + N, 2 * P + 2 * N)
    r = tin["approx_ratio"]
    inv_r = 0.0 if np.isinf(r) else 1.0 / r
    A0_p = A_0 / (1 + inv_r)                                         # peak:172-173
    A0_n = 0.0 if np.isinf(r) else A_0 / (1 + r)
    start = int(np.floor(wi0)) if fix_edge == 1 else int(np.floor(window))   # peak:178-184
    wi[pp] = start
    wi[pn] = start
    A[pp] = A0_p / P                                                 # peak:186-187
    A[pn] = A0_n / P
    wi_pp, wi_np = wi[pp].max(), wi[pn].max()
    i = np.arange(1, N + 1)
    spread = np.floor((window / N) * i).astype(np.int64)             # peak:198-199
    wi[cp] = int(np.floor(wi_pp * 1.0)) + spread
    wi[cn] = int(np.floor(wi_np * 1.0)) + spread
    Ac_p = (1 - A_0) / (1 + inv_r)                                   # peak:207-208
    Ac_n = 0.0 if np.isinf(r) else (1 - A_0) / (1 + r)
    A[cp] = (i / i.sum()) * Ac_p                                     # peak:210-211
    A[cn] = (i / i.sum()) * Ac_n
    if symm == 1 or kernel_type == "bosonic":                        # peak:214-227
        A[pp] = A_0 / P
        A[cp] = (1 - A_0) / N
        A[pn] = 0
        A[cn] = 0
        if symm == 1:
            A = A / (2 * A.sum())
        wi[pn] = 0
        wi[cn] = 0
    else:
        A = A / A.sum()                                              # peak:229
    n_grid = wim + 1
    n_tau = tin["n_tau"]
    Kp = np.zeros((n_grid, n_tau))
    Kn = np.zeros((n_grid, n_tau))
    U = tin["cov"]
    for a in range(0, n_grid, 65536):
        w = (np.arange(a, min(a + 65536, n_grid)) + 0.5) * d_omega   # peak:253 half-offset grid
        kv = kernel_values(kernel_type, w, tin["tau"], tin["beta"])
        if symm == 1:
            kv = kv + kernel_values(kernel_type, -w, tin["tau"], tin["beta"])     # peak:258-262
            Kp[a:a + len(w)] = kv @ U
        else:
            Kp[a:a + len(w)] = kv @ U                                # peak:265-266
            if kernel_type != "bosonic":
                Kn[a:a + len(w)] = kernel_values(kernel_type, -w, tin["tau"], tin["beta"]) @ U   # peak:270-272
    return dict(wi0=wi0, wim=wim, n_grid=n_grid, wi=wi, A=A, Kp=Kp, Kn=Kn, window=window)


class OraclePeak:
    """One reference Markov chain of sac_peak.jl."""

    def __init__(self, Kp, Kn, G, sigma_inv, n_omega, n_p, wi0, wim, symm=0, bosonic=0, fix_edge=0):
        self.L = _peak_lib()
        self.Kp = np.ascontiguousarray(Kp, dtype=np.float64)
        self.Kn = np.ascontiguousarray(Kn, dtype=np.float64)
        self.n_grid, self.n_tau = self.Kp.shape
        self.n_omega, self.n_p = int(n_omega), int(n_p)
        self.n_tot = 2 * (self.n_omega + self.n_p)
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        self.h = self.L.ora_peak_create(self.n_tau, self.n_omega, self.n_p, int(wi0), int(wim), int(symm),
                                        int(bosonic), int(fix_edge), _dp(self.Kp), _dp(self.Kn), _dp(G),
                                        _dp(sigma_inv))
        if not self.h:
            raise ValueError("ora_peak_create failed")
        self._log = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_peak_destroy(self.h)
            self.h = None

    def set_state(self, wi=None, A=None, windows=None):
        wi = None if wi is None else np.ascontiguousarray(wi, dtype=np.int64)
        A = None if A is None else np.ascontiguousarray(A, dtype=np.float64)
        windows = None if windows is None else np.ascontiguousarray(windows, dtype=np.float64)
        self.L.ora_peak_set_state(self.h, None if wi is None else _ip(wi), None if A is None else _dp(A),
                                  None if windows is None else _dp(windows))

    def get_state(self):
        wi = np.empty(self.n_tot, dtype=np.int64)
        A = np.empty(self.n_tot)
        w = np.empty(11)
        self.L.ora_peak_get_state(self.h, _ip(wi), _dp(A), _dp(w))
        return wi, A, w

    def init(self):
        self.L.ora_peak_init(self.h)

    def bounds(self):
        b = np.empty(4)
        self.L.ora_peak_get_bounds(self.h, _dp(b))
        return b

    def get_gbar(self):
        g = np.empty(self.n_tau)
        c, m = C.c_double(), C.c_double()
        self.L.ora_peak_get_gbar(self.h, _dp(g), C.byref(c), C.byref(m))
        return g, c.value, m.value

    def use_philox(self, seed, chain, sweep0=0):
        self.L.ora_peak_use_philox(self.h, seed, chain, sweep0)

    def set_indiv_update(self, on):
        self.L.ora_peak_set_indiv_update(self.h, int(on))

    def enable_log(self, cap):
        self._log = np.zeros(cap, dtype=MOVE_REC)
        self.L.ora_peak_set_log(self.h, self._log.ctypes.data, cap)

    def log(self):
        return self._log[:min(self.L.ora_peak_log_len(self.h), len(self._log))]

    def sweep_counter(self):
        return self.L.ora_peak_sweep_counter(self.h)

    def run_updates(self, theta, transfer=True):
        self.L.ora_peak_run_updates(self.h, theta, int(transfer))

    def adjust_windows(self, steps, theta):
        self.L.ora_peak_adjust_windows(self.h, steps, theta)

    def sample(self, steps, theta, bin=False):
        self.L.ora_peak_sample(self.h, steps, theta, int(bin))

    def accept_rates(self):
        a = np.empty(11)
        self.L.ora_peak_get_accept_rates(self.h, _dp(a))
        return a

    def sampled_chi2(self):
        return self.L.ora_peak_sampled_chi2(self.h)

    def edge_res(self):
        e = np.empty(6)
        self.L.ora_peak_get_edge_res(self.h, _dp(e))
        return e.reshape(3, 2)

    def set_edge_res(self, e):
        e = np.ascontiguousarray(e, dtype=np.float64).reshape(6)
        self.L.ora_peak_set_edge_res(self.h, _dp(e))

    def zero_spec(self):
        self.L.ora_peak_zero_spec(self.h)

    def spec(self, which):
        s = np.empty(self.n_grid)
        self.L.ora_peak_get_spec(self.h, which, _dp(s))
        return s

    def counters(self):
        c = (C.c_uint64 * 5)()
        self.L.ora_peak_get_counters(self.h, c)
        return [int(x) for x in c]
