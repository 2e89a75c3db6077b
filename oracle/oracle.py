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


# --------------------------------------------------------------------------
# edge parameterisation (sac/edge/sac_edge.jl)
# --------------------------------------------------------------------------

class _EdgeCfg(C.Structure):
    _fields_ = [("n_tau", C.c_int32), ("n_e", C.c_int32), ("n_omega", C.c_int32), ("n_rl", C.c_int32),
                ("double_in", C.c_int32), ("fix_edge", C.c_int32), ("wi0", C.c_int64), ("n_grid", C.c_int64),
                ("A_c", C.c_double), ("A_r", C.c_double), ("d0", C.c_double), ("w_floor", C.c_double * 2),
                ("w_m", C.c_double), ("dw", C.c_double), ("c", C.c_double * 2)]


def _edge_lib():
    L = lib()
    if getattr(L, "_edge_ready", False):
        return L
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p
    L.ora_sac_log.restype = C.c_double
    L.ora_sac_log.argtypes = [C.c_double]
    L.ora_sac_exp.restype = C.c_double
    L.ora_sac_exp.argtypes = [C.c_double]
    L.ora_edge_create.restype = vp
    L.ora_edge_create.argtypes = [C.POINTER(_EdgeCfg), C.POINTER(dp), C.POINTER(dp), C.POINTER(dp), dp, dp]
    L.ora_edge_destroy.argtypes = [vp]
    L.ora_edge_set_state.argtypes = [vp, dp, dp, dp, ip, dp]
    L.ora_edge_get_state.argtypes = [vp, dp, dp, dp, ip, dp]
    L.ora_edge_init_gbar.argtypes = [vp]
    L.ora_edge_profile.argtypes = [vp, C.c_int, C.c_double, C.c_double, dp]
    L.ora_edge_get_gbar.argtypes = [vp, dp, dp, dp]
    L.ora_edge_use_philox.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64]
    L.ora_edge_set_log.argtypes = [vp, vp, C.c_int64]
    L.ora_edge_log_len.restype = C.c_int64
    L.ora_edge_log_len.argtypes = [vp]
    L.ora_edge_sweep_counter.restype = C.c_uint64
    L.ora_edge_sweep_counter.argtypes = [vp]
    L.ora_edge_violations.restype = C.c_int64
    L.ora_edge_violations.argtypes = [vp]
    L.ora_edge_get_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ora_edge_run_updates.argtypes = [vp, C.c_int64, C.c_double]
    L.ora_edge_run_bins.argtypes = [vp, C.c_int64, C.c_int64, C.c_double]
    L.ora_edge_get_results.argtypes = [vp, dp, dp]
    L.ora_edge_get_accept_rates.argtypes = [vp, C.c_int, C.c_int, dp]
    L.ora_edge_get_spec.argtypes = [vp, C.c_int, C.c_int, dp]
    L.ora_edge_get_avg.argtypes = [vp, dp, dp]
    L._edge_ready = True
    return L


def sac_log(x):
    return _edge_lib().ora_sac_log(float(x))


def sac_exp(x):
    return _edge_lib().ora_sac_exp(float(x))


def read_tin_edge(path):
    """edge:130-164 read_G!: free's fields plus edge_guess from the log-slope of G at tau <= beta/4."""
    t = read_tin(path)
    q = t["tau"] <= (t["beta"] // 4)                      # edge:150-151
    Gh, th = t["G"][q], t["tau"][q]
    t["edge_guess"] = float(-(np.log(Gh[-1]) - np.log(Gh[-2])) / (th[-1] - th[-2]))   # edge:153
    return t


EDGE_MODES = ("single_edge", "double_edge_in", "double_edge_out", "double_edge_symm")


class EdgeTables:
    """init_kernal! (edge:167-242): K, dK, d2K on the grid w_i = i dw, i = wi0 .. wim, in the
    covariance eigenbasis, for side rl = 0 (K(w)) and rl = 1 (K(-w)); symm: K(w) + K(-w).
    Columns are produced on demand (the same arithmetic as the full tables) so that known-answer tests
    on the reference's multi-million-column grids stay cheap; full() materialises them."""

    def __init__(self, tin, mode, kernel_type, omega_0, omega_m, d_omega):
        if mode not in EDGE_MODES:
            raise ValueError("mode")
        self.tin, self.mode, self.kernel_type, self.dw = tin, mode, kernel_type, float(d_omega)
        self.wi0 = int(np.floor(omega_0 / d_omega))       # edge:169-170
        self.wim = int(np.ceil(omega_m / d_omega))
        self.n_grid = self.wim - self.wi0 + 1
        self.n_rl = 2 if mode in ("double_edge_in", "double_edge_out") else 1

    def _K(self, rl, col):
        """K_array[:, col+1, rl+1] for 0-based columns `col` (any shape) -> (..., n_tau)."""
        t = self.tin
        col = np.asarray(col)
        w = ((col + self.wi0) * self.dw).ravel()          # edge:195
        if self.mode == "double_edge_symm":
            kv = kernel_values(self.kernel_type, w, t["tau"], t["beta"]) + \
                kernel_values(self.kernel_type, -w, t["tau"], t["beta"])          # edge:223
        else:
            kv = kernel_values(self.kernel_type, w if rl == 0 else -w, t["tau"], t["beta"])
        return (kv @ t["cov"]).reshape(col.shape + (t["n_tau"],))                 # transpose(cov) * K[:, i]

    def columns(self, rl, col):
        """(K, dK, d2K) at 0-based columns col (1-D int array), edge:229-236 including its end rules:
        first column forward difference, the column with 1-based index wim+1 backward difference
        (the last column only when wi0 == 0), zeros where the reference leaves zeros."""
        col = np.asarray(col, dtype=np.int64)
        last = self.n_grid - 1
        c = np.clip(col, 0, last)
        Kc = self._K(rl, c)
        Km = self._K(rl, np.clip(c - 1, 0, last))
        Kp = self._K(rl, np.clip(c + 1, 0, last))
        dK = (Kp - Km) / (2 * self.dw)
        d2K = (Kp - (2 * Kc) + Km) / (2 * self.dw ** 2)
        edge_cols = (c == 0) | (c == last)
        dK[edge_cols] = 0.0
        d2K[edge_cols] = 0.0
        first = c == 0
        dK[first] = ((Kp - Kc) / self.dw)[first]          # edge:234
        special = c == self.wim                           # edge:235 (1-based wim+1)
        if self.wim <= last:
            dK[special] = ((Kc - Km) / self.dw)[special]
        return Kc, dK, d2K

    def taylor(self, rl, w):
        """K(w) by the reference's lookup (edge:595-606)."""
        w = np.asarray(w, dtype=np.float64)
        wi = np.floor(w / self.dw).astype(np.int64)
        d = w - wi * self.dw
        K, dK, d2K = self.columns(rl, wi - self.wi0)
        return (K + dK * d[:, None]) + d2K * (d ** 2)[:, None]

    def full(self):
        """[(K, dK, d2K) per side], each [n_grid, n_tau] C-order."""
        out = []
        for rl in range(self.n_rl):
            parts = [self.columns(rl, np.arange(a, min(a + 65536, self.n_grid)))
                     for a in range(0, self.n_grid, 65536)]
            out.append(tuple(np.ascontiguousarray(np.concatenate([p[i] for p in parts])) for i in range(3)))
        return out


def edge_amplitudes(n_e, n_omega, c, n0, eps0, A_c, A_r, rl):
    """set_amplitudes! (edge:354-390) for one side: array of N_w + 1 (Julia rows 1 .. N_w+1)."""
    A = np.zeros(n_omega + 1)
    eps1 = (np.log(2) * c) ** 2 * eps0                    # edge:357
    i = np.arange(1, n_e + 1)
    x = np.log(i / n0)
    root = np.sqrt((c * x) ** 2 + eps1)
    ln_A = c * x + root if c < 0 else c * x - root        # edge:360-364
    A[1:n_e + 1] = np.exp(ln_A / 2)
    A[n_e + 1:] = 1.0                                     # edge:370-374
    if rl == 0:
        f1, f2 = A_r / (1 + A_c), A_c / (1 + A_c)         # edge:378-384
    else:
        f1, f2 = (1 - A_r) / (1 + A_c), 0.0
    A[1:n_e + 1] *= f1 / A[1:n_e + 1].sum()               # edge:386
    if n_omega > n_e:
        A[n_e + 1:] *= f2 / A[n_e + 1:].sum()             # edge:388
    return A


class EdgeInit:
    """init_struct! + init_config_dual! (edge:269-583) in numpy: the deterministic 50 x 50 scan for the
    initial edge.  Arrays are [N_w+1, 2] (Julia shape, 0-based rows here)."""

    def __init__(self, tin, tables, n_e, n_c, p, A_c, A_r, omega_0, omega_m, fix_edge_value=0.0, d0=0.0):
        self.tin, self.tab = tin, tables
        self.mode = tables.mode
        self.n_e, self.n_omega = int(n_e), int(n_e + n_c)
        self.A_c, self.A_r, self.d0 = float(A_c), float(A_r), float(d0)
        self.c = 1 - 2 * p                                # edge:1971
        self.w_m = float(omega_m)
        self.fix_edge = 1 if fix_edge_value != 0 else 0
        self.rls = list(range(tables.n_rl))
        N1 = self.n_omega + 1
        self.w = np.zeros((N1, 2))
        self.d = np.zeros((N1, 2))
        self.A = np.zeros((N1, 2))
        self.dw = np.zeros((N1, 2))
        self.nm = np.zeros((N1, 2), dtype=np.int64)
        # omega_floor (edge:1999-2015)
        if self.mode == "double_edge_in":
            self.w_floor = [omega_0, -omega_m]
        elif self.fix_edge == 1:
            self.w_floor = [fix_edge_value, 1.5 * fix_edge_value]
        else:
            self.w_floor = [omega_0, omega_0]
        self.n0 = np.array([n_e, n_e]) * 0.75             # set_amplitudes!(m = 0), edge:341-348
        self.eps0 = np.array([0.5, 0.5])
        self.dn0 = np.array([2.0, 2.0])
        self.deps0 = np.array([0.05, 0.05])
        for rl in self.rls:
            self.A[:, rl] = edge_amplitudes(self.n_e, self.n_omega, self.c, self.n0[rl], self.eps0[rl],
                                            self.A_c, self.A_r, rl)

    def config_jk(self, j, k):
        """init_config_jk! (edge:399-481)."""
        w, d, NE, N, wm = self.w, self.d, self.n_e, self.n_omega, self.w_m
        g = np.sign(self.tin["edge_guess"])
        k0 = 55
        if self.fix_edge == 1:
            w[1, :] = w[0, :]
            dd = abs((wm * (k + 5) / k0) - w[1, 0])
        elif self.mode == "single_edge":
            w[1, 0] = w[0, 0] + (g * ((j - 1) * wm / 500))
            dd = abs((wm * (k + 5) / k0) - w[1, 0])
        elif self.mode == "double_edge_in":
            w[1, 0] = w[0, 0] + (g * (j * wm / 500))
            w[1, 1] = w[1, 0] - ((k + 5) * wm / k0)
            dd = abs(-w[1, 1] - w[1, 0]) * 0.5
        elif self.mode == "double_edge_out":
            w[1, 0] = w[0, 0] + (j * wm / 500)
            w[1, 1] = w[0, 1] + (k * wm / 500)
            dd = abs((wm * 0.5) - w[1, 0])
        else:
            w[1, 0] = w[0, 0] + (j * wm / 500)
            dd = abs((wm * (k + 5) / k0) - w[1, 0])
        d[1:NE + 1, :] = np.sqrt(np.arange(1, NE + 1))[:, None]              # edge:447
        d[2:NE + 1, :] *= dd / d[2:NE + 1, :].sum()                         # edge:448 (both columns)
        for i in range(2, NE + 1):                                           # edge:451-455
            for rl in self.rls:
                w[i, rl] = w[i - 1, rl] + d[i, rl]
        if NE < N:                                                           # edge:458-473
            w[NE + 1:, 1] = w[2, 1]
            w[NE + 1, 0] = w[2, 0]
            if self.mode == "double_edge_in":
                dd = abs(-w[1, 1] - w[NE + 1, 0]) * 0.5
            else:
                dd = abs(w[NE, 0] - w[NE + 1, 0]) * 0.5
            flat = d.reshape(-1, order="F")                                  # linear index N_e+2:end
            flat[NE + 1:] = dd / (N - NE)
            d[:, :] = flat.reshape(d.shape, order="F")
            for i in range(NE + 2, N + 1):
                w[i, 0] = w[i - 1, 0] + dd / (N - NE)

    def gbar(self):
        """calc_Gbar!(self, 1) (edge:591-622)."""
        g = self.tab.taylor(0, self.w[:, 0]).T @ self.A[:, 0]
        if self.mode in ("double_edge_in", "double_edge_out"):
            g = g + self.tab.taylor(1, self.w[:, 1]).T @ self.A[:, 1]
        return g

    def chi2(self):
        return float((((self.gbar() - self.tin["G_D"]) * self.tin["sigma_inv"]) ** 2).sum())

    def config_dual(self):
        """init_config_dual! (edge:487-583).  Returns (j_opt, k_opt, chi2_min)."""
        NE, N = self.n_e, self.n_omega
        j_max, k_max = 50, 50
        if self.fix_edge == 1:
            self.w[0, :] = self.w_floor
            j_max = 1
        else:
            self.w[0, :] = 0.0
        best, k_opt, j_opt = 1e20, 1, 1
        for k in range(1, k_max + 1):
            for j in range(1, j_max + 1):
                self.config_jk(j, k)
                if self.mode == "double_edge_in":
                    if self.w[1, 0] > -self.w[NE, 1]:
                        continue
                    if self.w[:, 0].max() > -self.w[1, 1]:
                        continue
                c2 = self.chi2()
                if c2 < best:
                    k_opt, j_opt, best = k, j, c2
        self.config_jk(j_opt, k_opt)
        for rl in self.rls:                                                  # edge:567-578
            self.nm[1:NE - 1, rl] = 1 + (N // 20)
            for i in range(2, NE):
                self.nm[i - 1, rl] = min(self.nm[i - 1, rl], NE - i - 1)
            self.nm[NE + 1:, rl] = -1
            self.dw[1, rl] = self.d[2, rl]
            for i in range(2, N + 1):
                self.dw[i, rl] = self.d[i, rl]
        return j_opt, k_opt, best

    def scalars(self):
        return np.concatenate([self.n0, self.eps0, self.dn0, self.deps0])


class OracleEdge:
    """One reference Markov chain of sac_edge.jl (state arrays in the reference's [N_w+1, 2] shape)."""

    def __init__(self, tables_full, G, sigma_inv, n_e, n_omega, wi0, dw, w_m, w_floor, c, A_c=0.0, A_r=1.0,
                 d0=0.0, double_in=0, fix_edge=0):
        self.L = _edge_lib()
        self.n_rl = len(tables_full)
        self._keep = [[np.ascontiguousarray(t, dtype=np.float64) for t in side] for side in tables_full]
        self.n_grid, self.n_tau = self._keep[0][0].shape
        self.n_e, self.n_omega = int(n_e), int(n_omega)
        dp = C.POINTER(C.c_double)
        arrs = [(dp * self.n_rl)(*[_dp(side[i]) for side in self._keep]) for i in range(3)]
        cfg = _EdgeCfg(self.n_tau, self.n_e, self.n_omega, self.n_rl, int(double_in), int(fix_edge), int(wi0),
                       self.n_grid, float(A_c), float(A_r), float(d0), (C.c_double * 2)(*w_floor), float(w_m),
                       float(dw), (C.c_double * 2)(float(c), float(c)))
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        self.h = self.L.ora_edge_create(C.byref(cfg), arrs[0], arrs[1], arrs[2], _dp(G), _dp(sigma_inv))
        if not self.h:
            raise ValueError("ora_edge_create failed")
        self._log = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_edge_destroy(self.h)
            self.h = None

    @staticmethod
    def _f(a):
        return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).ravel()

    def set_state(self, w=None, A=None, dw=None, nm=None, scal=None):
        """w, A, dw, nm: [N_w+1, 2]; scal: n0[2], eps0[2], dn0[2], deps0[2]."""
        w, A, dw = self._f(w), self._f(A), self._f(dw)
        nm = None if nm is None else np.ascontiguousarray(np.asarray(nm, dtype=np.int64).T).ravel()
        scal = None if scal is None else np.ascontiguousarray(scal, dtype=np.float64)
        self.L.ora_edge_set_state(self.h, *[None if a is None else _dp(a) for a in (w, A, dw)],
                                  None if nm is None else _ip(nm), None if scal is None else _dp(scal))

    def get_state(self):
        n1 = self.n_omega + 1
        w, A, dw, scal = np.empty(2 * n1), np.empty(2 * n1), np.empty(2 * n1), np.empty(8)
        nm = np.empty(2 * n1, dtype=np.int64)
        self.L.ora_edge_get_state(self.h, _dp(w), _dp(A), _dp(dw), _ip(nm), _dp(scal))
        r = lambda a: a.reshape(2, n1).T.copy()
        return dict(w=r(w), A=r(A), dw=r(dw), nm=r(nm), scal=scal)

    def init_gbar(self):
        self.L.ora_edge_init_gbar(self.h)

    def profile(self, rl, n0, eps0):
        out = np.empty(self.n_e)
        self.L.ora_edge_profile(self.h, rl, n0, eps0, _dp(out))
        return out

    def get_gbar(self):
        g = np.empty(self.n_tau)
        c, m = C.c_double(), C.c_double()
        self.L.ora_edge_get_gbar(self.h, _dp(g), C.byref(c), C.byref(m))
        return g, c.value, m.value

    def use_philox(self, seed, chain, sweep0=0):
        self.L.ora_edge_use_philox(self.h, seed, chain, sweep0)

    def enable_log(self, cap):
        self._log = np.zeros(cap, dtype=MOVE_REC)
        self.L.ora_edge_set_log(self.h, self._log.ctypes.data, cap)

    def log(self):
        return self._log[:min(self.L.ora_edge_log_len(self.h), len(self._log))]

    def sweep_counter(self):
        return self.L.ora_edge_sweep_counter(self.h)

    def violations(self):
        return self.L.ora_edge_violations(self.h)

    def counters(self):
        c = (C.c_uint64 * 5)()
        self.L.ora_edge_get_counters(self.h, c)
        return list(c)

    def run_updates(self, steps, theta):
        self.L.ora_edge_run_updates(self.h, steps, theta)

    def run_bins(self, steps, bins, theta):
        self.L.ora_edge_run_bins(self.h, steps, bins, theta)

    def results(self):
        r, a = np.empty(21), np.empty(6)
        self.L.ora_edge_get_results(self.h, _dp(r), _dp(a))
        return dict(chi2_res=r[:3].copy(), edge_res=r[3:9].reshape(3, 2).copy(), n0_res=r[9:15].reshape(3, 2).copy(),
                    wn0_res=r[15:21].reshape(3, 2).copy(), amp_ar=a.reshape(3, 2).copy())

    def accept_rates(self, m, rl):
        a = np.empty(self.n_omega + 1)
        self.L.ora_edge_get_accept_rates(self.h, m, rl, _dp(a))
        return a

    def spec(self, which, rl):
        o = np.empty(self.n_grid)
        self.L.ora_edge_get_spec(self.h, which, rl, _dp(o))
        return o

    def avg(self):
        n1 = self.n_omega + 1
        w, A = np.empty(2 * n1), np.empty(2 * n1)
        self.L.ora_edge_get_avg(self.h, _dp(w), _dp(A))
        return w.reshape(2, n1).T.copy(), A.reshape(2, n1).T.copy()
