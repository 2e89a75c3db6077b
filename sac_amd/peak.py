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
        self._tables = None

    @property
    def Kp(self):
        """Host-built tables, built on first use (device_tables=True never needs them)."""
        if self._tables is None:
            self._tables = self._kernel_tables()
        return self._tables[0]

    @property
    def Kn(self):
        if self._tables is None:
            self._tables = self._kernel_tables()
        return self._tables[1]

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

    def make_handle(self, n_chains=1, seed=0, device_tables=False, **kw):
        """SAC(...) + initialize! + calc_Gbar/calc_chi2 + default windows (peak:1482-1507).
        device_tables: build Kp / Kn on the GPU (sac_build_table) instead of uploading host-built tables."""
        h = SacHandle(self.td.n_tau, self.n_omega, self.n_grid, self.wi_0, self.wi_m, n_chains=n_chains,
                      symm=self.symm, seed=seed, kind=KIND_PEAK, n_peak=self.n_p, fix_edge=self.fix_edge,
                      bosonic=int(self.kernel_type == "bosonic"), **kw)
        if device_tables:
            td = self.td
            h.build_table(0, self.kernel_type, td.tau, td.beta, td.cov, 0, self.d_omega, offset=0.5, mirror=int(self.symm == 1))
            if self.sample_neg:
                h.build_table(1, self.kernel_type, td.tau, td.beta, td.cov, 0, self.d_omega, offset=0.5, negate=1)
            h.upload_data(td.G, td.sigma_inv)
        else:
            h.upload_problem_peak(self.Kp, self.Kn, self.td.G, self.td.sigma_inv)
        h.set_state(self.wi, self.A, np.full(11, self.window / 10))  # peak:1507
        h.init_gbar()
        return h


# --------------------------------------------------------------------------------------------
# Annealing driver: sac_peak.jl's run() and schedule routines (peak:1272-1569)
# --------------------------------------------------------------------------------------------

class PeakSAC:
    """The reference's SAC struct + schedule functions for the `peak` parameterisation; the
    fixed-Theta stages run on the GPU (SAC_KIND_PEAK).  Statistics are means over chains
    (chi2_min: the minimum)."""

    def __init__(self, N_omega, N_p, A_0, omega_m, d_omega, d_omega_h, theta_0, f_anneal, a_criterion,
                 N_anneal, anneal_steps, sample_steps, G_file, output_folder, fix_edge, symm, kernel_type,
                 n_chains=1, seed=0, tol=1e-3, device=0, timedata=None, comm=None, chi2_min_mode="chain_mean",
                 device_tables=False):
        if chi2_min_mode not in ("chain_mean", "min"):
            raise ValueError("chi2_min_mode must be 'chain_mean' or 'min'")
        self.device_tables = bool(device_tables)   # Kp / Kn built on the GPU (sac_build_table, row f1) instead of peak:248-276 on the host
        self.chi2_min_mode = chi2_min_mode      # see FreeSAC: the single-chain equivalent of the reference's chi2_min, or the minimum
        self.comm = comm
        self.is_root = comm is None or comm.rank == 0
        self.N_omega, self.N_p, self.A_0 = int(N_omega), int(N_p), float(A_0)
        self.omega_m, self.d_omega, self.d_omega_h = float(omega_m), float(d_omega), float(d_omega_h)
        self.theta_0, self.f_anneal, self.a_criterion = float(theta_0), float(f_anneal), float(a_criterion)
        self.N_anneal, self.anneal_steps, self.sample_steps = int(N_anneal), int(anneal_steps), int(sample_steps)
        self.G_file, self.output_folder = G_file, output_folder
        self.symm, self.kernel_type = int(symm), str(kernel_type)
        if fix_edge != 0:                                            # peak:1466-1471
            self.omega_0, self.fix_edge = float(fix_edge), 1
        else:
            self.omega_0, self.fix_edge = 0.0, 0
        self.tol, self.n_chains, self.seed, self.device = tol, n_chains, seed, device
        self.timedata = timedata
        self.chi2_anneal = np.zeros(self.N_anneal)
        self.indiv_update = False
        self.h = None

    @classmethod
    def from_infile(cls, path="in_peak.in", A_0_in=None, N_p_in=None, **kw):
        """peak:1441-1473."""
        with open(path) as f:
            rows = [ln.split() for ln in f]
        N_w, N_p = int(float(rows[0][0])), int(float(rows[0][1]))
        A_0, w_m, dw, dwh = (float(x) for x in rows[1][:4])
        th0, fa, acrit = (float(x) for x in rows[2][:3])
        Na, asteps, ssteps = (int(float(x)) for x in rows[3][:3])
        G_file, out = rows[4][0], rows[4][1]
        if len(rows[5]) >= 3:
            fix_edge, symm, ktype = float(rows[5][0]), int(float(rows[5][1])), rows[5][2]
        else:   # the reference's committed runs hold an older two-token row ("0 finiteT"): flag, kernel
            fix_edge, symm, ktype = 0.0, int(float(rows[5][0])), rows[5][1]
        if A_0_in is not None:
            A_0 = A_0_in
        if N_p_in is not None:
            N_p = int(N_p_in)
        if ktype == "bosonic":
            symm = 0
        elif symm == 1:
            out += "_symm"
        out = out + "/Np_%02i/A0_%.3f" % (N_p, A_0)
        return cls(N_w, N_p, A_0, w_m, dw, dwh, th0, fa, acrit, Na, asteps, ssteps, G_file, out, fix_edge,
                   symm, ktype, **kw)

    # ---- files ---------------------------------------------------------------------------
    @property
    def anneal_file(self):
        return self.output_folder + "/anneal.csv"

    @property
    def sample_file(self):
        return self.output_folder + "/sample.csv"

    @property
    def accept_rate_file(self):
        return self.output_folder + "/accept_rate.csv"

    def init_outfiles(self):                                        # peak:310-343
        import os
        if not self.is_root:
            return
        os.makedirs(self.output_folder, exist_ok=True)
        with open(self.anneal_file, "w") as f:
            f.write("i,theta,chi2_min,chi2_avg,edge_p,edge_n,Ap_p,Ap_n,Ac_p,Ac_n\n")
        with open(self.sample_file, "w") as f:
            f.write("i,a,theta,chi2_min,chi2_avg,edge_p,edge_n,sigma_edge_p,sigma_edge_n,Ap_p,Ap_n,Ac_p,Ac_n\n")
        with open(self.accept_rate_file, "w") as f:
            f.write("i,ar_pcont1,ar_pcont2,ar_pcont3,ar_ncont1,ar_ncont2,ar_ncont3,∆w_pcont1,∆w_pcont2,"
                    "∆w_ncont1,∆w_ncont2,ar_ppeak,ar_npeak,∆w_ppeak,∆w_npeak,ar_wAc,ar_wAp,ar_wApc,"
                    "∆w_Ac,∆w_Ap,∆w_Apc\n")
        open(self.output_folder + "/log.txt", "w").close()

    def write_log(self, message, mode="a"):                         # peak:1430-1434
        import datetime
        if not self.is_root:
            return
        now = datetime.datetime.now().isoformat(timespec="milliseconds")
        with open(self.output_folder + "/log.txt", mode) as f:
            f.write(f"{now} - {message}\n")

    # ---- initialisation (peak:1500-1507) ---------------------------------------------------
    def initialize(self):
        from .tin import read_tin
        if self.h is not None:
            self.h.close()
        td = self.timedata if self.timedata is not None else read_tin(self.G_file)
        self.td = td
        self.prob = PeakProblem(td, self.N_omega, self.N_p, self.A_0, self.omega_m, self.d_omega,
                                omega_0=self.omega_0, fix_edge=self.fix_edge, symm=self.symm,
                                kernel_type=self.kernel_type)
        offset = 0 if self.comm is None else self.comm.rank * self.n_chains
        self.h = self.prob.make_handle(n_chains=self.n_chains, seed=self.seed, device=self.device, chain_offset=offset,
                                       device_tables=self.device_tables)
        self.h.set_indiv_update(self.indiv_update)
        if self.comm is not None:
            self.comm.attach(self.h)                                # RCCL groups: the merge goes behind the C ABI
        self._merge()

    def _merge(self):
        st = self.h.peak_stats()                                    # over all ranks already when the handle merges
        self.sampled_chi2, self.chi2_min = st.sampled_chi2, st.chi2_min
        cmm = st.chi2_min_mean
        self.accept_rates = np.array(st.accept_rates[:])
        self.update_windows = np.array(st.windows[:])
        self.edge_res = np.array([list(r) for r in st.edge_res])
        self.bounds = np.array(st.bounds[:])
        self.weights = np.array(st.weights[:])
        if self.comm is not None and not self.h.merged_over_ranks:  # host-side groups: one all-reduce per stage
            m = self.comm.merge_named(self.n_chains, dict(sampled_chi2=self.sampled_chi2, accept_rates=self.accept_rates,
                                                          update_windows=self.update_windows, edge_res=self.edge_res,
                                                          bounds=self.bounds, weights=self.weights, chi2_min_mean=cmm),
                                      mins=dict(chi2_min=self.chi2_min))
            for k in ("sampled_chi2", "chi2_min", "accept_rates", "update_windows", "edge_res", "bounds", "weights"):
                setattr(self, k, m[k])
            cmm = m["chi2_min_mean"]
        # the schedule criteria (peak:1343, 1386, 1545) see ONE chain's minimum in the reference: chain mean by default
        self.chi2_min_all = self.chi2_min
        if self.chi2_min_mode == "chain_mean":
            self.chi2_min = cmm
        return st

    def repair_chains(self):
        """n_chains > 1 only (no counterpart in the one-chain reference).  The reference's transfer
        moves test lazily refreshed bounds (peak:918-934), so with the initial, very wide windows a
        continuum delta and the peak can cross and lock each other for good; a user of the reference
        sees chi2 ~ 1e7 N_tau and starts over (its own test for a failed start is chi2_min > 1000 N_tau,
        peak:1522).  With many chains, a chain that fails that same test takes over the configuration
        of a healthy chain and continues on its own random stream.  Returns the number replaced."""
        if self.n_chains == 1:
            return 0
        _, c2min = self.h.chain_chi2()
        bad = c2min > 1000 * self.td.n_tau
        good = np.flatnonzero(~bad)
        if len(good) == 0 or not bad.any():
            return 0
        good = good[np.argsort(c2min[good], kind="stable")]
        for j, c in enumerate(np.flatnonzero(bad)):
            self.h.copy_chain(int(good[j % len(good)]), int(c))
        return int(bad.sum())

    def adjust_windows(self, steps, theta):                         # peak:1063
        self.h.adjust_windows(steps, theta)

    def sample(self, steps, theta, bin=False):                      # peak:1100
        if bin:
            self.h.sample_bin(steps, theta)
        else:
            self.h.sample(steps, theta)
        self._merge()

    def _edges_and_weights(self):
        """edge_p, edge_n, A0_p, A0_n, Ac_p, Ac_n as printed by peak:1301-1312."""
        edge_p = self.bounds[0] * self.d_omega
        A0_p, Ac_p = self.weights[0], self.weights[1]
        if not self.prob.sample_neg:
            return edge_p, -edge_p, A0_p, A0_p, Ac_p, Ac_p
        return edge_p, -self.bounds[2] * self.d_omega, A0_p, self.weights[2], Ac_p, self.weights[3]

    def _accept_row(self, i, final):
        """peak:1322-1333 / 1410-1421 (the final row rounds the windows to 4 digits and does not
        scale window 11 by d_omega, as the reference does)."""
        from .free import _jl_round, jl_float
        if not self.is_root:
            return
        ar, w, dw = self.accept_rates, self.update_windows, self.d_omega
        wd = 4 if final else 8
        vals = ([_jl_round(x, 4) for x in ar[[0, 1, 2, 7, 8, 9]]] + [_jl_round(x * dw, wd) for x in w[[0, 1, 7, 8]]] +
                [_jl_round(x, 4) for x in ar[[3, 4]]] + [_jl_round(x * dw, wd) for x in w[[3, 4]]] +
                [_jl_round(x, 4) for x in ar[[5, 6, 10]]] + [_jl_round(x * dw, 8) for x in w[[5, 6]]] +
                [_jl_round(w[10] if final else w[10] * dw, 8)])
        with open(self.accept_rate_file, "a") as f:
            f.write(",".join([jl_float(float(i))] + [jl_float(v) for v in vals]) + "\n")

    # ---- schedule ------------------------------------------------------------------------
    def intial_sampling(self, theta_0):                             # peak:1272-1276
        self.adjust_windows(self.anneal_steps, 10 * theta_0)
        self.adjust_windows(self.anneal_steps, 5 * theta_0)
        self.adjust_windows(self.anneal_steps, 2 * theta_0)
        self._merge()

    def run_anneal(self, theta_0, wr=False, on_stage=None):         # peak:1289-1350
        from .free import _jl_round, jl_float
        theta, n_tau = theta_0, self.td.n_tau
        for i in range(1, self.N_anneal + 1):
            steps = self.anneal_steps
            self.adjust_windows(steps, theta)
            self.sample(steps, theta)
            if on_stage is not None:
                on_stage(i, theta)
            edge_p, edge_n, A0_p, A0_n, Ac_p, Ac_n = self._edges_and_weights()
            vals = [theta, self.chi2_min / n_tau, self.sampled_chi2 / n_tau, edge_p, edge_n, A0_p, A0_n, Ac_p, Ac_n]
            if self.is_root:
                with open(self.anneal_file, "a") as f:              # peak:1315-1319
                    f.write(",".join([jl_float(float(i))] + [jl_float(_jl_round(v, 8)) for v in vals]) + "\n")
            self._accept_row(i, final=False)
            if wr:
                self.write_spec(i)
            self.chi2_anneal[i - 1] = self.sampled_chi2             # peak:1341
            if (self.sampled_chi2 - self.chi2_min) < (self.tol * n_tau):   # peak:1343
                return
            theta /= self.f_anneal

    def final_anneal(self, theta_opt):                              # peak:1354-1427
        from .free import _jl_round, jl_float
        n_tau = self.td.n_tau
        self.write_log("Beginning Fast Anneal to θ_opt.")
        for i in range(1, 11):
            theta = theta_opt * (11 - i)
            steps = self.anneal_steps * i
            self.adjust_windows(steps, theta)
            self.sample(steps, theta)
        self.write_log("Fast Anneal to θ_opt Finished.")
        self.write_log("Beginning Final Sampling at θ_opt.")
        N_bins = 10
        self.h.zero_accumulators()                                  # peak:1370-1372
        theta = theta_opt
        for _ in range(N_bins):
            self.sample(self.sample_steps, theta, bin=True)
        self.spec_divisor = N_bins * self.sample_steps              # peak:1380-1381
        e = self.edge_res / N_bins                                  # peak:1382-1384
        e[2] = np.sqrt(np.abs(e[2] - e[1] ** 2)) / np.sqrt(N_bins - 1)
        e = e * self.d_omega
        self.edge_final = e
        a = (self.sampled_chi2 - self.chi2_min) / np.sqrt(2 * self.chi2_min)
        _, _, A0_p, A0_n, Ac_p, Ac_n = self._edges_and_weights()
        vals = [0.0, _jl_round(a, 4), _jl_round(theta, 8), _jl_round(self.chi2_min / n_tau, 4),
                _jl_round(self.sampled_chi2 / n_tau, 4), _jl_round(e[1, 0], 4), _jl_round(-e[1, 1], 4),
                _jl_round(e[2, 0], 4), _jl_round(e[2, 1], 4), _jl_round(A0_p, 4), _jl_round(A0_n, 4),
                _jl_round(Ac_p, 4), _jl_round(Ac_n, 4)]
        if self.is_root:
            with open(self.sample_file, "a") as f:                  # peak:1400-1408
                f.write(",".join(jl_float(v) for v in vals) + "\n")
        self._accept_row(0, final=True)
        self.write_spec(0)
        self.write_log("Final Sampling at θ_opt Finished.")

    def write_spec(self, n):
        """peak:1170-1265."""
        from .free import jl_float
        p = self.prob
        div = getattr(self, "spec_divisor", 1.0) if self.h is not None else 1.0
        hists = [self.h.histogram_peak(w) / (div if self._binned() else 1.0) for w in range(4)]
        if self.comm is not None and not self.h.merged_over_ranks:
            m = self.comm.merge_named(self.n_chains, {str(w): hists[w] for w in range(4)})
            hists = [m[str(w)] for w in range(4)]
        if not self.is_root:
            return None
        pspec = np.stack([hists[0], hists[2]], axis=1)
        cspec = np.stack([hists[1], hists[3]], axis=1)
        i = np.arange(0, p.wi_m + 1)
        f = 1.0 + np.exp(-self.td.beta * (self.d_omega * i)) if self.kernel_type == "bosonic" else np.ones(len(i))
        scale = ((self.td.norm * np.pi) / f)[:, None]               # peak:1176-1183
        pspec, cspec = pspec * scale, cspec * scale
        conv = int(round(self.d_omega_h / self.d_omega))            # peak:1185-1186
        N_h = int(p.wi_m / conv)
        P = np.zeros((N_h, 2))
        Cc = np.zeros((N_h, 2))
        for k in range(N_h):                                        # peak:1189-1197 (no overlap here)
            P[k] = pspec[k * conv:(k + 1) * conv].sum(axis=0) / self.d_omega_h
            Cc[k] = cspec[k * conv:(k + 1) * conv].sum(axis=0) / self.d_omega_h

        def last(col):                                              # peak:1202-1216
            j = N_h
            for k in range(N_h - 1, -1, -1):
                j = k
                if Cc[k, col] > 1e-10:
                    break
            return j
        jf, kf = last(0), last(1)
        with open(f"{self.output_folder}/sw{n:03d}.csv", "w") as fh:
            fh.write("omega,S,S_p,S_c\n")
            if self.kernel_type == "bosonic":                       # peak:1226-1233
                for k in range(jf, 0, -1):
                    w = self.d_omega_h * (k - 1)
                    e = np.exp(-self.td.beta * w)
                    Sp, Sc = P[k - 1, 0], Cc[k - 1, 0]
                    fh.write(",".join(jl_float(v) for v in (-w, (Sp + Sc) * e, Sp * e, Sc * e)) + "\n")
            elif self.symm == 1:                                    # peak:1235-1242
                for k in range(jf, 0, -1):
                    w = self.d_omega_h * (k - 1)
                    Sp, Sc = P[k - 1, 0], Cc[k - 1, 0]
                    fh.write(",".join(jl_float(v) for v in (-w, Sp + Sc, Sp, Sc)) + "\n")
            else:                                                   # peak:1245-1251
                for k in range(kf, 0, -1):
                    w = self.d_omega_h * (k - 1)
                    Sp, Sc = P[k - 1, 1], Cc[k - 1, 1]
                    fh.write(",".join(jl_float(v) for v in (-w, Sp + Sc, Sp, Sc)) + "\n")
            for k in range(1, jf + 1):                              # peak:1255-1261
                w = self.d_omega_h * (k - 1)
                Sp, Sc = P[k - 1, 0], Cc[k - 1, 0]
                fh.write(",".join(jl_float(v) for v in (w, Sp + Sc, Sp, Sc)) + "\n")
        return P, Cc

    def _binned(self):
        return hasattr(self, "spec_divisor")

    def optimal_theta(self):                                        # peak:1545-1548
        a_vals = (self.chi2_anneal - self.chi2_min) / np.sqrt(2 * self.chi2_min)
        theta_vals = self.theta_0 / (self.f_anneal ** np.arange(self.N_anneal))
        return a_vals, theta_vals, float(theta_vals[np.argmin(np.abs(a_vals - self.a_criterion))])

    def run(self, in_file=None):                                    # peak:1439-1569
        import os
        import shutil
        from .free import jl_float
        if self.is_root:
            os.makedirs(self.output_folder, exist_ok=True)
            if in_file and os.path.exists(in_file):
                shutil.copyfile(in_file, self.output_folder + "/in_peak.in")
            if self.timedata is None and os.path.exists(self.G_file):
                shutil.copyfile(self.G_file, self.output_folder + "/t.in")
        self.init_outfiles()
        self.write_log("Beginning Initialization.", "w")
        self.initialize()
        theta = self.theta_0
        self.write_log("Initialization Finished.")
        self.write_log("Beginning Initial Sampling.")
        self.intial_sampling(theta)
        if self.chi2_min_all > 1000 * self.td.n_tau:                # peak:1522-1533 (no chain at all found a start)
            self.indiv_update = True
            self.initialize()
            self.intial_sampling(theta)
        n_fixed = self.repair_chains()
        self.write_log(f"Initial Sampling Finished ({n_fixed} of {self.n_chains} chains restarted from a healthy chain).")
        self.write_log("Beginning Anneal.")
        self.run_anneal(theta, False)
        self.write_log("Anneal Finished.")
        a_vals, theta_vals, theta_opt = self.optimal_theta()
        if self.is_root:
            with open(self.output_folder + "/a_vals.csv", "w") as f:    # peak:1551-1554
                f.write("a,theta\n")
                for a, t in zip(a_vals, theta_vals):
                    f.write(f"{jl_float(a)},{jl_float(t)}\n")
        self.h.set_state(windows=np.full(11, self.prob.window / 10))   # peak:1557
        self.write_log("Beginning Final Anneal.")
        self.final_anneal(theta_opt)
        self.write_log("Final Anneal Finished.")
        return theta_opt
