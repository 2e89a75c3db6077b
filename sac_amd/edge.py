"""Host side of the `edge` parameterisation: mirror of sac/edge/sac_edge.jl above the C ABI.

"edge:NNN" = /root/reference/sac/edge/sac_edge.jl line NNN.  Every Monte-Carlo sweep runs in
libsac_hip.so (sac_edge_*); this module builds what the reference builds once on the host (read_G!
edge_guess, init_kernal!: K / dK / d2K tables, set_amplitudes!, the deterministic 50 x 50 scan
init_config_dual!) and mirrors the schedule and output routines (anneal, anneal_and_sample,
anneal_and_scan, write_res, write_spec, run).
"""
import datetime
import math
import os
import shutil

import numpy as np

from .free import _jl_round, jl_float
from .lib import SacEdgeHandle
from .synth import kernel_values
from .tin import read_tin

MODES = ("single_edge", "double_edge_in", "double_edge_out", "double_edge_symm")


def edge_guess(td):
    """edge:150-153: minus the log-slope of G at the last two tau <= beta/4."""
    q = td.tau <= (td.beta // 4)
    G, t = td.G_raw[q], td.tau[q]
    return float(-(np.log(G[-1]) - np.log(G[-2])) / (t[-1] - t[-2]))


class EdgeKernel:
    """init_kernal! (edge:167-242): K, dK, d2K [n_grid, n_tau] per side, eigenbasis of the covariance."""

    def __init__(self, td, mode, kernel_type, omega_0, omega_m, d_omega):
        if mode not in MODES:
            raise ValueError(f"mode {mode!r}")
        if kernel_type not in ("finiteT", "zeroT", "bosonic"):
            raise ValueError("Invalid Kernel type.")                        # edge:191
        self.td, self.mode, self.kernel_type, self.dw = td, mode, kernel_type, float(d_omega)
        self.wi0 = int(math.floor(omega_0 / d_omega))                       # edge:169-170
        self.wim = int(math.ceil(omega_m / d_omega))
        self.n_grid = self.wim - self.wi0 + 1
        if self.wi0 > 0:
            raise ValueError("omega_0 > 0: the reference's dK_array[:, wi_m+1] (edge:235) is out of bounds")
        self.n_rl = 2 if mode in ("double_edge_in", "double_edge_out") else 1
        self.tables = [self._build(rl) for rl in range(self.n_rl)]

    def _values(self, rl, lo, hi):
        w = np.arange(lo, hi) * self.dw                                     # edge:195 (i * dw, i absolute)
        td = self.td
        if self.mode == "double_edge_symm":                                 # edge:223
            kv = kernel_values(self.kernel_type, w, td.tau, td.beta) + kernel_values(self.kernel_type, -w, td.tau, td.beta)
        else:
            kv = kernel_values(self.kernel_type, w if rl == 0 else -w, td.tau, td.beta)
        return kv @ td.cov                                                  # transpose(cov) * K[:, i]

    def _build(self, rl):
        n, nt, dw = self.n_grid, self.td.n_tau, self.dw
        K = np.empty((n, nt))
        step = 1 << 16
        for a in range(0, n, step):
            b = min(a + step, n)
            K[a:b] = self._values(rl, self.wi0 + a, self.wi0 + b)
        dK = np.zeros((n, nt))
        d2K = np.zeros((n, nt))
        for a in range(1, n - 1, step):                                     # edge:230-233
            b = min(a + step, n - 1)
            dK[a:b] = (K[a + 1:b + 1] - K[a - 1:b - 1]) / (2 * dw)
            d2K[a:b] = (K[a + 1:b + 1] - (2 * K[a:b]) + K[a - 1:b - 1]) / (2 * dw ** 2)
        dK[0] = (K[1] - K[0]) / dw                                          # edge:234
        dK[self.wim] = (K[self.wim] - K[self.wim - 1]) / dw                 # edge:235 (1-based wi_m + 1)
        return K, dK, d2K

    def lookup(self, rl, w):
        """calc_Gbar!'s Taylor lookup (edge:595-606) for an array of frequencies -> [len, n_tau]."""
        w = np.asarray(w, dtype=np.float64)
        wi = np.floor(w / self.dw).astype(np.int64)
        d = w - wi * self.dw
        col = np.clip(wi - self.wi0, 0, self.n_grid - 1)
        K, dK, d2K = self.tables[rl]
        return (K[col] + dK[col] * d[:, None]) + d2K[col] * (d ** 2)[:, None]


def set_amplitudes(n_e, n_omega, c, n0, eps0, A_c, A_r, rl):
    """set_amplitudes! (edge:354-390), one side: N_w + 1 amplitudes (row 0 = the unused leading slot)."""
    A = np.zeros(n_omega + 1)
    eps1 = (math.log(2) * c) ** 2 * eps0
    i = np.arange(1, n_e + 1)
    x = np.log(i / n0)
    root = np.sqrt((c * x) ** 2 + eps1)
    A[1:n_e + 1] = np.exp((c * x + root if c < 0 else c * x - root) / 2)
    A[n_e + 1:] = 1.0
    f1, f2 = (A_r / (1 + A_c), A_c / (1 + A_c)) if rl == 0 else ((1 - A_r) / (1 + A_c), 0.0)
    A[1:n_e + 1] *= f1 / A[1:n_e + 1].sum()
    if n_omega > n_e:
        A[n_e + 1:] *= f2 / A[n_e + 1:].sum()
    return A


class EdgeProblem:
    """init_struct! + init_config_dual! (edge:269-583): kernel tables and the initial configuration."""

    def __init__(self, td, mode, kernel_type, n_e, n_c, p, A_c, A_r, omega_0, omega_m, d_omega, fix_edge=0.0, d0=0.0):
        self.td, self.mode, self.kernel_type = td, mode, kernel_type
        self.n_e, self.n_omega = int(n_e), int(n_e) + int(n_c)
        self.A_c, self.A_r, self.d0, self.p = float(A_c), float(A_r), float(d0), float(p)
        self.c = 1 - 2 * p                                                  # edge:1971
        self.w_0, self.w_m, self.dw = float(omega_0), float(omega_m), float(d_omega)
        self.fix_edge = 1 if fix_edge != 0 else 0
        if mode == "double_edge_in":                                        # edge:1999-2015
            self.w_floor = [omega_0, -omega_m]
        elif self.fix_edge == 1:
            self.w_floor = [fix_edge, 1.5 * fix_edge]
        else:
            self.w_floor = [omega_0, omega_0]
        self.edge_guess = edge_guess(td)
        self.kernel = EdgeKernel(td, mode, kernel_type, omega_0, omega_m, d_omega)
        self.rls = list(range(self.kernel.n_rl))
        N1 = self.n_omega + 1
        self.w, self.d, self.A = np.zeros((N1, 2)), np.zeros((N1, 2)), np.zeros((N1, 2))
        self.dwin = np.zeros((N1, 2))
        self.nm = np.zeros((N1, 2), dtype=np.int64)
        self.n0 = np.array([n_e, n_e]) * 0.75                               # set_amplitudes!(m = 0)
        self.eps0 = np.array([0.5, 0.5])
        self.dn0 = np.array([2.0, 2.0])
        self.deps0 = np.array([0.05, 0.05])
        for rl in self.rls:
            self.A[:, rl] = set_amplitudes(self.n_e, self.n_omega, self.c, self.n0[rl], self.eps0[rl], A_c, A_r, rl)
        self.j_opt, self.k_opt, self.chi2_init = self._config_dual()

    def _config_jk(self, j, k):                                             # edge:399-481
        w, d, NE, N, wm = self.w, self.d, self.n_e, self.n_omega, self.w_m
        g, k0 = np.sign(self.edge_guess), 55
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
        d[1:NE + 1, :] = np.sqrt(np.arange(1, NE + 1))[:, None]
        d[2:NE + 1, :] *= dd / d[2:NE + 1, :].sum()                         # both columns (edge:448)
        for i in range(2, NE + 1):
            for rl in self.rls:
                w[i, rl] = w[i - 1, rl] + d[i, rl]
        if NE < N:                                                          # edge:458-473
            w[NE + 1:, 1] = w[2, 1]
            w[NE + 1, 0] = w[2, 0]
            dd = abs(-w[1, 1] - w[NE + 1, 0]) * 0.5 if self.mode == "double_edge_in" else abs(w[NE, 0] - w[NE + 1, 0]) * 0.5
            flat = d.reshape(-1, order="F")                                 # d_array[N_e+2:end], linear index
            flat[NE + 1:] = dd / (N - NE)
            d[:, :] = flat.reshape(d.shape, order="F")
            for i in range(NE + 2, N + 1):
                w[i, 0] = w[i - 1, 0] + dd / (N - NE)

    def gbar(self):
        g = self.kernel.lookup(0, self.w[:, 0]).T @ self.A[:, 0]
        if self.kernel.n_rl == 2:
            g = g + self.kernel.lookup(1, self.w[:, 1]).T @ self.A[:, 1]
        return g

    def chi2(self):
        return float((((self.gbar() - self.td.G) * self.td.sigma_inv) ** 2).sum())

    def _config_dual(self):                                                 # edge:487-583
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
                self._config_jk(j, k)
                if self.mode == "double_edge_in":
                    if self.w[1, 0] > -self.w[NE, 1] or self.w[:, 0].max() > -self.w[1, 1]:
                        continue
                c2 = self.chi2()
                if c2 < best:
                    k_opt, j_opt, best = k, j, c2
        self._config_jk(j_opt, k_opt)
        for rl in self.rls:
            self.nm[1:NE - 1, rl] = 1 + (N // 20)
            for i in range(2, NE):
                self.nm[i - 1, rl] = min(self.nm[i - 1, rl], NE - i - 1)
            self.nm[NE + 1:, rl] = -1
            self.dwin[1, rl] = self.d[2, rl]
            self.dwin[2:N + 1, rl] = self.d[2:N + 1, rl]
        return j_opt, k_opt, best

    def scalars(self):
        return np.concatenate([self.n0, self.eps0, self.dn0, self.deps0])

    def make_handle(self, n_chains=1, seed=0, device_tables=False, **kw):
        """device_tables: K, dK, d2K built on the GPU (sac_edge_build_tables) instead of uploading the host-built
        ones (the host tables still serve the initial scan, edge:399-583)."""
        k = self.kernel
        h = SacEdgeHandle(self.td.n_tau, self.n_e, self.n_omega, k.n_grid, k.wi0, k.dw, self.w_m, self.w_floor, self.c,
                          n_rl=k.n_rl, double_in=int(self.mode == "double_edge_in"), fix_edge=self.fix_edge,
                          A_c=self.A_c, A_r=self.A_r, d0=self.d0, n_chains=n_chains, seed=seed, **kw)
        if device_tables:
            td = self.td
            for rl in range(k.n_rl):
                h.build_tables(rl, k.kernel_type, td.tau, td.beta, td.cov, k.wi0, k.dw,
                               mirror=int(k.mode == "double_edge_symm"), negate=int(rl == 1))
            h.upload_data(td.G, td.sigma_inv)
        else:
            h.upload_problem(k.tables, self.td.G, self.td.sigma_inv)
        nm = np.where(self.nm < 0, 0, self.nm)                              # the -1 markers (edge:572) are never read
        h.set_state(self.w, self.A, self.dwin, nm, self.scalars())
        h.init_gbar()
        return h


class EdgeSAC:
    """The reference's SAC struct + schedule functions for the edge parameterisation (edge:1452-2065).
    With n_chains > 1 every printed statistic is the mean over chains; chi2_min, which the reference's criteria take from
    its ONE chain (edge:1682, 1718), is by default the mean over chains of each chain's own minimum (chi2_min_mode=
    "chain_mean": Theta_opt then does not drift with the number of chains), or the minimum over all chains ("min")."""

    def __init__(self, N_e, N_c, omega_0, omega_m, d_omega_h, d_omega, p, A_c, A_r, theta_0, f_anneal, N_anneal,
                 a_criterion, anneal_steps, sample_steps, bins, G_file, output_folder, fix_edge, kernel_type, mode,
                 n_chains=1, seed=0, tol=1e-3, device=0, timedata=None, comm=None, spread_sampling=False,
                 chi2_min_mode="chain_mean"):
        if chi2_min_mode not in ("chain_mean", "min"):
            raise ValueError("chi2_min_mode must be 'chain_mean' or 'min'")
        self.chi2_min_mode = chi2_min_mode
        self.comm = comm
        # no counterpart in the one-chain reference: the final sampling's sweeps per bin are shared out over all chains
        self.spread_sampling = bool(spread_sampling)
        self.is_root = comm is None or comm.rank == 0
        self.N_e, self.N_c = int(N_e), int(N_c)
        self.omega_0, self.omega_m, self.d_omega_h, self.d_omega = float(omega_0), float(omega_m), float(d_omega_h), float(d_omega)
        self.p, self.A_c, self.A_r = float(p), float(A_c), float(A_r)
        self.theta_0, self.f_anneal, self.N_anneal, self.a_criterion = float(theta_0), float(f_anneal), int(N_anneal), float(a_criterion)
        self.anneal_steps, self.sample_steps, self.bins = int(anneal_steps), int(sample_steps), int(bins)
        self.G_file, self.output_folder = G_file, output_folder
        self.fix_edge, self.kernel_type, self.mode = float(fix_edge), str(kernel_type), str(mode)
        self.n_chains, self.seed, self.tol, self.device, self.timedata = n_chains, seed, tol, device, timedata
        self.chi2_anneal = np.zeros(self.N_anneal)
        self.h = None

    @classmethod
    def from_infile(cls, path="in_edge.in", A_c_in=None, A_r_in=None, p_in=None, scan=False, **kw):
        """edge:1884-1962."""
        with open(path) as f:
            rows = [ln.split() for ln in f]
        N_e, N_c = int(float(rows[0][0])), int(float(rows[0][1]))
        w0, wm, dwh, dw = (float(x) for x in rows[1][:4])                   # edge:1888: omega_0, omega_m, dw_h, dw
        p, A_c, A_r = (float(x) for x in rows[2][:3])
        th0, fa, Na, a = (float(x) for x in rows[3][:4])
        asteps, ssteps, bins = (int(float(x)) for x in rows[4][:3])
        G_file, out = rows[5][0], rows[5][1]
        fix_edge, ktype = float(rows[6][0]), rows[6][1]
        mode = rows[7][0]
        A_c = A_c if A_c_in is None else A_c_in
        A_r = A_r if A_r_in is None else A_r_in
        p = p if p_in is None else p_in
        if mode == "single_edge":                                           # edge:1909-1921
            out += "_single/"
            A_r = 1.0
        elif mode == "double_edge_in":
            out += "_double_in/"
        elif mode == "double_edge_out":
            out += "_double_out/"
            w0 = 0.0
        elif mode == "double_edge_symm":
            out += "_double_symm/"
            w0, A_r = 0.0, 0.5
        else:
            raise ValueError(f"mode {mode!r}")
        settings = []
        if fix_edge != 0:
            settings.append("fixed")
        if A_c > 0.0:                                                       # edge:1931-1942
            if N_c == 0:
                N_c = int(math.ceil(0.5 * N_e))
            assert A_c < 1
        if A_c == 0.0:
            N_c = 0
        settings.append(f"Nw{N_e}")
        if scan:
            settings.append("scan")
        out += "_".join(settings)
        out += "/Ac_%.3f" % A_c + "/p_%.3f" % p
        if mode in ("double_edge_in", "double_edge_out"):
            out += "/Ar_%.3f" % A_r
        if ktype == "bosonic":                                              # edge:1964-1969
            if mode != "single_edge":
                raise ValueError("Only mode that can be run with the bosonic kernel is single_edge. Exiting.")
            w0 = 0.0
        return cls(N_e, N_c, w0, wm, dwh, dw, p, A_c, A_r, th0, fa, Na, a, asteps, ssteps, bins, G_file, out,
                   fix_edge, ktype, mode, **kw)

    # ---- files ---------------------------------------------------------------------------
    def _path(self, name):
        return os.path.join(self.output_folder, name)

    def write_log(self, message, mode="a"):                                 # edge:1850-1860
        if not self.is_root:
            return
        now = datetime.datetime.now().isoformat(timespec="milliseconds")
        with open(self._path("log.txt"), mode) as f:
            f.write(f"{now} - {message}\n")

    def init_outfiles(self):                                                # edge:305-334
        if not self.is_root:
            return
        head = ("j,theta,chi2_min,chi2_avg,chi2_sigma,edge_R,wn0_R,n0_R,eps_R,edge_R_sigma,wn0_R_sigma,n0_R_sigma,"
                "edge_L,wn0_L,n0_L,eps_L,edge_L_sigma,wn0_L_sigma,n0_L_sigma,wc0,w_cf\n")
        for name in ("anneal.csv", "sample.csv"):
            with open(self._path(name), "w") as f:
                f.write(head)
        for name in ("accept_rate_1.csv", "accept_rate_2.csv"):
            with open(self._path(name), "w") as f:
                f.write("j,ar_cont,∆w_cont,ar_cont_amp,ar_edge_s,∆w_edge,ar_edge_m,n_clust,ar_edge_amp\n")

    # ---- initialisation (edge:2038-2049) -------------------------------------------------
    def initialize(self):
        os.makedirs(self.output_folder, exist_ok=True)
        td = self.timedata if self.timedata is not None else read_tin(self.G_file)
        self.td = td
        self.write_log("Initializing Kernel.")
        self.prob = EdgeProblem(td, self.mode, self.kernel_type, self.N_e, self.N_c, self.p, self.A_c, self.A_r,
                                self.omega_0, self.omega_m, self.d_omega, fix_edge=self.fix_edge)
        pr = self.prob
        self.write_log("Finding Best Initial Configuration.")
        self.write_log(f"Rightward Edge: [{jl_float(pr.w[1, 0])}, {jl_float(pr.w[pr.n_e, 0])}] ({pr.j_opt}, {pr.k_opt})")
        if self.mode in ("double_edge_in", "double_edge_out"):              # edge:555-557
            self.write_log(f"Leftward Edge: [{jl_float(-pr.w[1, 1])}, {jl_float(-pr.w[pr.n_e, 1])}]")
        if self.h is not None:
            self.h.close()
        offset = 0 if self.comm is None else self.comm.rank * self.n_chains
        self.h = pr.make_handle(n_chains=self.n_chains, seed=self.seed, device=self.device, chain_offset=offset)
        if self.comm is not None:
            self.comm.attach(self.h)                                        # RCCL groups: the merge goes behind the C ABI
        self.chi2_min = self.chi2_min_all = self.h.get_gbar(0)[2]           # all chains start from the same configuration
        self.write_log("Initialization Finished.")

    # ---- stage -----------------------------------------------------------------------------
    def run_bins(self, steps, bins, theta):                                 # edge:1357
        self.h.run_bins(steps, bins, theta)
        self.res = self.h.results(-1)                                       # over all ranks already when the handle merges
        if self.comm is not None and not self.h.merged_over_ranks:          # host-side groups: one all-reduce per stage
            r = self.res
            keys = ("chi2_res", "edge_res", "n0_res", "wn0_res", "amp_ar", "n0", "eps0", "dn0", "deps0", "chi2", "chi2_min_mean")
            m = self.comm.merge_named(self.n_chains, {k: r[k] for k in keys}, mins={"chi2_min": r["chi2_min"]},
                                      sums={"violations": r["violations"], "chains_below_target": r["chains_below_target"]})
            self.res = dict(r, **m)
            self.res["n_chains_total"] = m["n_chains"]
            self.res["chains_below_target"] = int(round(m["chains_below_target"]))
        self.chi2_min_all = self.res["chi2_min"]
        self.chi2_min = self.res["chi2_min_mean"] if self.chi2_min_mode == "chain_mean" else self.res["chi2_min"]
        if self.res["violations"]:
            with open(self._path("error.txt"), "a") as f:                   # the reference exit()s here (edge:668)
                f.write(f"{jl_float(theta)} ordering / bound violations in {self.res['violations']} chain-bins\n")
            raise RuntimeError("check_w: ordering or bound violation (see error.txt)")

    def _move_means(self):
        """Chain means over ALL chains of all ranks of the per-delta windows, the cluster sizes and the continuum's extent
        (write_res, edge:1454-1506): one call, summed on the device (sac_edge_get_move_means)."""
        dw, nm, ext = self.h.move_means()
        if self.comm is not None and not self.h.merged_over_ranks:          # host-side groups
            m = self.comm.merge_named(self.n_chains, {"dw": dw, "nm": nm, "ext": ext})
            dw, nm, ext = m["dw"], m["nm"], m["ext"]
        return dw, nm, ext

    def _hist(self, which, rl):
        h = self.h.histogram(which, rl)
        if self.comm is None or self.h.merged_over_ranks:
            return h
        tot = self.comm.sum_vector(np.concatenate([h * self.n_chains, [float(self.n_chains)]]))
        return tot[:-1] / tot[-1]

    def _accept_rates(self, m, rl):
        a = self.h.accept_rates(m, rl)
        if self.comm is None or self.h.merged_over_ranks:
            return a
        return self.comm.merge_named(self.n_chains, {"a": a})["a"]

    def _avg(self):
        w, A = self.h.avg(-1)
        if self.comm is None or self.h.merged_over_ranks:
            return w, A
        m = self.comm.merge_named(self.n_chains, {"w": w, "A": A})
        return m["w"], m["A"]

    def write_res(self, j, theta):                                          # edge:1452-1510
        r, NE, N, nt = self.res, self.N_e, self.prob.n_omega, self.td.n_tau
        dwm, nmm, ext = self._move_means()                                  # collective for host-side groups: every rank
        ar = {rl: (self._accept_rates(0, rl), self._accept_rates(1, rl)) for rl in self.prob.rls}
        if not self.is_root:
            return
        w_c0, w_cf = (float(ext[0]), float(ext[1])) if self.A_c > 0 else (0.0, 0.0)
        vals = [theta, self.chi2_min / nt, r["chi2_res"][1] / nt, r["chi2_res"][2] / nt,
                r["edge_res"][1, 0], r["wn0_res"][1, 0], r["n0_res"][1, 0], r["eps0"][0],
                r["edge_res"][2, 0], r["wn0_res"][2, 0], r["n0_res"][2, 0],
                -r["edge_res"][1, 1], -r["wn0_res"][1, 1], r["n0_res"][1, 1], r["eps0"][1],
                r["edge_res"][2, 1], r["wn0_res"][2, 1], r["n0_res"][2, 1], w_c0, w_cf]
        with open(self._path("anneal.csv"), "a") as f:
            f.write(f"{j}," + ",".join(jl_float(_jl_round(v, 8)) for v in vals) + "\n")
        for rl in self.prob.rls:
            ar1, ar2 = ar[rl]
            nc = N - NE
            with np.errstate(divide="ignore", invalid="ignore"):
                a1 = np.float64(ar1[NE + 1:].sum()) / nc                    # Julia: 0/0 -> NaN without continuum
                a2 = np.float64(dwm[NE + 1:, rl].sum()) / nc
            a3 = r["amp_ar"][1, rl]
            a4 = ar1[1:NE + 1].sum() / NE
            a5 = dwm[1:NE + 1, rl].sum() / NE
            a6 = ar2[2:NE - 1].sum() / (NE - 3)
            a7 = nmm[2:NE - 1, rl].sum() / (NE - 3)
            a8 = r["amp_ar"][0, rl]
            with open(self._path(f"accept_rate_{rl + 1}.csv"), "a") as f:
                f.write(f"{j}," + ",".join(jl_float(_jl_round(v, 8)) for v in (a1, a2, a3, a4, a5, a6, a7, a8)) + "\n")

    def write_spec(self, n, total_steps, rl):                               # edge:1514-1637
        k = self.prob.kernel
        wi0, wim, dw, dwh, NE = k.wi0, k.wim, self.d_omega, self.d_omega_h, self.N_e
        ngrid = wim + 1 - wi0
        hist = np.zeros((ngrid + 2, 2))
        spec_bin = np.stack([self._hist(0, rl), self._hist(1, rl)], axis=1) / total_steps
        w_avg, A_avg = self._avg()
        if not self.is_root:
            return
        spec_bin = spec_bin * (self.td.norm * np.pi)
        conv = int(math.ceil(dwh / dw))
        N_h = int(math.floor((wim - wi0) / conv))
        for i in range(1, N_h + 1):                                         # edge:1536-1537 (1-based window)
            hist[i - 1, :] = spec_bin[(i - 1) * conv + 1:i * conv + 1, :].sum(axis=0)
        hist /= dwh
        j1 = j2 = -1
        for i in range(0, N_h + 1):                                         # edge:1541-1548
            j1 = i + 1
            if hist[i, :].max() > 1e-10:
                j1 -= 1
                break
        j1 = max(1, j1)
        for i in range(N_h, -1, -1):                                        # edge:1550-1555
            j2 = i + 1
            if hist[i, :].max() > 1e-10:
                break
        fl = lambda v: jl_float(float(v))
        with open(self._path(f"sw{n:03d}_{rl + 1}.dat"), "w") as f:
            f.write("omega,S,S_edge,S_cont\n")
            f.write(f"{fl(dwh * (j1 - 1 - 0.5) + self.omega_0)},0,0,0\n")
            for i in range(j1, j2 + 1):
                w = dwh * (i - 0.5) + self.omega_0
                f.write(f"{fl(w)},{fl(hist[i - 1, 0] + hist[i - 1, 1])},{fl(hist[i - 1, 0])},{fl(hist[i - 1, 1])}\n")
            f.write(f"{fl(dwh * (j2 + 1 - 0.5) + self.omega_0)},0,0,0\n")
        # spectral density on the self-generated grid of the averaged deltas (edge:1573-1635)
        wb, Ab = w_avg[:, rl] / total_steps, A_avg[:, rl] / total_steps     # 0-based rows = Julia index - 1
        jj = NE - 1
        i2 = 0
        for i in range(1, jj + 1):
            gap = wb[i + 1] - wb[i]
            hist[i, 0] = (self.td.norm * np.pi) / gap
            hist[i, 0] *= 0.5 * (Ab[i] + Ab[i + 1])
            i1 = int(math.ceil(wb[i] / dw)) - wi0
            i2 = int(math.ceil(wb[i + 1] / dw)) - 1 - wi0
            hist[i, 1] = spec_bin[max(i1 - 1, 0):max(i2, 0), 1].sum() / gap
        dwh2 = wb[jj + 1] - wb[jj]
        conv2 = int(math.floor((dwh2 / dw) + 0.5))
        kk = jj
        while True:                                                         # edge:1601-1616
            i1 = i2 + 1
            i2 = i1 + conv2
            if i2 > (wim - wi0):
                break
            kk += 1
            hist[kk - 1, 0] = 0
            hist[kk - 1, 1] = spec_bin[i1 - 1:i2, 1].sum() / dwh2
            if hist[kk - 1, 1] < 1.0e-10:
                kk -= 1
                break
        with open(self._path(f"dw{n:03d}_{rl + 1}.dat"), "w") as f:
            f.write("omega,S,S_edge,S_cont\n")
            f.write(f"{fl(wb[1])},0,0,0\n")
            w = 0.0
            for i in range(1, jj + 1):
                w = 0.5 * (wb[i] + wb[i + 1])
                f.write(f"{fl(w)},{fl(hist[i, 0] + hist[i, 1])},{fl(hist[i, 0])},{fl(hist[i, 1])}\n")
            for i in range(jj + 1, kk + 1):
                w += dwh2
                f.write(f"{fl(w)},{fl(hist[i, 0] + hist[i, 1])},{fl(hist[i, 0])},{fl(hist[i, 1])}\n")
            f.write(f"{fl(w + dwh2 / 2.0)},0,0,0\n")

    # ---- annealing routines --------------------------------------------------------------
    def anneal(self, chi2_target, bins, spec, first_passage=False):         # edge:1646-1693
        """first_passage (no counterpart in the one-chain reference): the target rule of the second anneal is taken chain by
        chain -- every chain has its own target (sac_edge_set_targets_from_chi2_min) and the anneal stops at the stage in
        which half of all chains have had their stage average below it, i.e. at the MEDIAN of the chains' first passages.
        The reference's rule `chi2_res[2] < chi2_target` (edge:1680) is the first passage of its one chain's noisy curve;
        with one chain both rules are the same test."""
        theta = self.theta_0
        i_trans = self.N_anneal * 0.2
        nt = self.td.n_tau
        for i in range(1, self.N_anneal + 1):
            if i < i_trans:
                steps = int(math.ceil(self.anneal_steps * (1 - ((5 / 6) * i / i_trans))))
            else:
                steps = int(math.ceil(self.anneal_steps / 6))
            self.run_bins(steps, bins, theta)
            self.write_res(i, theta)
            if spec == 1:
                for rl in self.prob.rls:
                    self.write_spec(i, bins * steps, rl)
            if i == 5:                                                      # edge:1671-1675
                self.saved = [self.h.get_state(c) for c in range(self.n_chains)]
            chi2_avg = self.res["chi2_res"][1]
            if first_passage:
                reached = 2 * self.res["chains_below_target"] >= self.res["n_chains_total"]
            else:
                reached = chi2_avg < chi2_target
            if reached:
                return theta * self.f_anneal
            elif (chi2_avg - self.chi2_min) < (self.tol * nt):
                return theta
            else:
                theta /= self.f_anneal
                self.chi2_anneal[i - 1] = chi2_avg
        return theta

    def _restore_saved(self):                                               # edge:1724-1730
        for c, s in enumerate(self.saved):
            self.h.set_state(w=s["w"], dw=s["dw"], nm=s["nm"], chain=c)
        self.h.init_gbar()

    def anneal_and_sample(self):                                            # edge:1705-1774
        self.saved = None
        self.write_log("Beginning Main Anneal.")
        self.anneal(0.0, self.bins, 0)
        self.write_log("Main Anneal Finished.")
        chi2_target = self.chi2_min + self.a_criterion * math.sqrt(2 * self.chi2_min)
        # chain_mean mode: every chain gets its own target from its own chi2_min (before the restore resets chi2_min, edge:1730)
        first_passage = self.chi2_min_mode == "chain_mean"
        if first_passage:
            self.h.set_targets_from_chi2_min(self.a_criterion)
        if self.saved is not None:
            self._restore_saved()
        self.write_log("Beginning Final Anneal.")
        theta_opt = self.anneal(chi2_target, self.bins, 0, first_passage=first_passage)
        if first_passage:
            self.first_passages = self.h.first_passages()                   # this rank's chains: stage index of each first passage
            self.h.set_chain_targets(None)
        self.write_log("Final Anneal Finished.")
        self.write_log("Beginning Final Sampling.")
        bins = 10
        steps = self.sample_steps
        if self.spread_sampling:
            from .free import MIN_SPREAD_STEPS
            total = self.n_chains * (1 if self.comm is None else self.comm.world)
            steps = max(MIN_SPREAD_STEPS, -(-self.sample_steps // total))
        self.run_bins(steps, bins, theta_opt)
        for rl in self.prob.rls:
            self.write_spec(0, bins * steps, rl)
        self._sample_row(0, theta_opt, with_cont=True)
        self.write_log("Final Sampling Finished.")
        return theta_opt

    def _sample_row(self, j, theta, with_cont):
        ext = self._move_means()[2] if (with_cont and self.A_c > 0) else (0.0, 0.0)   # collective for host-side groups
        if not self.is_root:
            return
        r, NE, nt = self.res, self.N_e, self.td.n_tau
        vals = [theta, self.chi2_min / nt, r["chi2_res"][1] / nt, r["chi2_res"][2] / nt,
                r["edge_res"][1, 0], r["wn0_res"][1, 0], r["n0_res"][1, 0], r["eps0"][0],
                r["edge_res"][2, 0], r["wn0_res"][2, 0], r["n0_res"][2, 0],
                -r["edge_res"][1, 1], -r["wn0_res"][1, 1], r["n0_res"][1, 1], r["eps0"][1],
                r["edge_res"][2, 1], r["wn0_res"][2, 1], r["n0_res"][2, 1]]
        if with_cont:
            vals += [float(ext[0]), float(ext[1])] if self.A_c > 0 else [0.0, 0.0]
        with open(self._path("sample.csv"), "a") as f:
            f.write(f"{j}," + ",".join(jl_float(_jl_round(v, 8)) for v in vals) + "\n")

    def anneal_and_scan(self, theta_1, theta_2):                            # edge:1783-1843
        self.N_anneal = int(math.ceil(math.log(self.theta_0 / theta_1) / math.log(self.f_anneal)))
        self.chi2_anneal = np.zeros(self.N_anneal)
        self.write_log("Beginning First Anneal.")
        self.saved = None
        self.write_log("Beginning Main Anneal.")
        self.anneal(0.0, self.bins // 2, 0)
        self.write_log("Main Anneal Finished.")
        if self.chi2_min > 2 * self.td.n_tau:
            return self.anneal_and_scan(theta_1, theta_2)
        self.write_log("Beginning Final Sampling.")
        f_scan = 1.15
        N_scan = int(math.ceil(math.log(theta_1 / theta_2) / math.log(f_scan)))
        theta = theta_1
        for i in range(1, N_scan + 1):
            bins = int(math.ceil(self.bins * (1 + i / N_scan)))
            self.run_bins(self.sample_steps, bins, theta)
            self._sample_row(i, theta, with_cont=False)
            theta /= f_scan
            for rl in self.prob.rls:
                self.write_spec(i, bins * self.sample_steps, rl)
        self.write_log("Final Sampling Finished.")

    def run(self, in_file=None, theta_1=None, theta_2=None):                # edge:1882-2065
        os.makedirs(self.output_folder, exist_ok=True)
        if self.is_root and in_file and os.path.exists(in_file):
            shutil.copyfile(in_file, self._path("in_edge.in"))
        if self.is_root and self.timedata is None and os.path.exists(self.G_file):
            shutil.copyfile(self.G_file, self._path("t.in"))
        n_w = self.N_e + (self.N_c if self.A_c > 0 else 0)
        self.write_log(f"N_ω, N_e, A_r, A_c, p = {n_w}, {self.N_e}, {jl_float(self.A_r)}, {jl_float(self.A_c)},{jl_float(self.p)}", "w")
        self.write_log("Beginning Initialization.")
        if self.mode in ("single_edge", "double_edge_symm"):
            self.write_log("Only using rightward decaying spectrum.")
        else:
            self.write_log("Using rightward and leftward decaying spectrum (double edge).")
        self.write_log("Reading G_file.")
        self.init_outfiles()
        self.initialize()
        self.write_log("Running Equillibration Sweeps.")
        self.run_bins(5 * self.anneal_steps, self.bins, self.theta_0)       # edge:2052-2053
        self.write_res(0, self.theta_0)
        if theta_1 is None:
            return self.anneal_and_sample()
        return self.anneal_and_scan(theta_1, theta_2)
