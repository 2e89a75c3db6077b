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
        self._K = None

    @property
    def K(self):
        """The host-built table (built on first use: device_tables=True never needs it)."""
        if self._K is None:
            self._K = self._kernel_table()
        return self._K

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

    def make_handle(self, n_chains=1, seed=0, device_tables=False, **kw):
        """SAC(...) + read_G! + initialize! + calc_Gbar/calc_chi2 + default windows (free:1191-1218).
        device_tables: build the K table on the GPU (sac_build_table) instead of uploading the host-built one."""
        h = SacHandle(self.td.n_tau, self.n_omega, self.n_grid, self.wi_0, self.wi_m,
                      n_chains=n_chains, par=self.par, symm=self.symm, seed=seed, **kw)
        if device_tables:
            h.build_table(0, self.kernel_type, self.td.tau, self.td.beta, self.td.cov, self.wi_0, self.d_omega,
                          mirror=int(self.symm == 1))
            h.upload_data(self.td.G, self.td.sigma_inv)
        else:
            h.upload_problem(self.K, self.td.G, self.td.sigma_inv)
        h.set_state(self.wi, self.A, np.full(4, self.window))
        h.init_gbar()
        return h


# --------------------------------------------------------------------------------------------
# Annealing driver: the reference's run() and its schedule routines (free:931-1261)
# --------------------------------------------------------------------------------------------

def jl_float(x):
    """Float64 the way Julia's print/writedlm shows it (shortest round-trip digits; fixed notation
    when the decimal exponent is in [-4, 6), else d.ddde<exp>)."""
    import math
    x = float(x)
    if x != x:
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    r = repr(abs(x))
    if "e" in r:
        mant, exp = r.split("e")
        exp = int(exp)
    else:
        mant, exp = r, 0
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    if ip.strip("0"):
        e10 = exp + len(ip.lstrip("0")) - 1
    else:
        e10 = exp - (len(fp) - len(fp.lstrip("0"))) - 1
    digits = digits.rstrip("0") or "0"
    sign = "-" if x < 0 else ""
    if -4 <= e10 < 6:
        if e10 >= 0:
            whole = digits[:e10 + 1].ljust(e10 + 1, "0")
            frac = digits[e10 + 1:] or "0"
        else:
            whole, frac = "0", "0" * (-e10 - 1) + digits
        return f"{sign}{whole}.{frac}"
    return f"{sign}{digits[0]}.{digits[1:] or '0'}e{e10}"


def _jl_round(x, digits):
    """Julia round(x, digits=d): round-half-even on x*10^d, then divide (Base.round for Float64)."""
    p = 10.0 ** digits
    return np.round(x * p) / p


MIN_SPREAD_STEPS = 20       # spread_sampling: a chain's share of the final sampling is never shorter than this


class FreeSAC:
    """The reference's SAC struct + schedule functions for the `free` parameterisation, with the
    fixed-Theta stages (adjust_windows, sample) running on the GPU behind the C ABI.

    With n_chains > 1 every stage statistic is the mean over chains.  chi2_min, which the reference's schedule
    criteria take from its ONE chain (stop rule free:1087, a-values free:1128, Theta_opt free:1249-1252), is by default
    the single-chain equivalent -- the mean over chains of each chain's own minimum (chi2_min_mode="chain_mean"), so
    that Theta_opt does not drift with the number of chains; chi2_min_mode="min" takes the minimum over all chains (the
    lowest chi2 any chain has seen; it lowers the target as chains are added).  self.chi2_min_all always holds that
    minimum.  With a process group the per-stage merge is one fused all-reduce(sum) + one all-reduce(min) -- behind the
    C ABI for RCCL groups (dist.Comm.attach) -- so all ranks take the same schedule decisions."""

    def __init__(self, par, N_omega, omega_0, omega_m, d_omega, d_omega_h, theta_0, f_anneal, f_final,
                 a1, a2, N_anneal, anneal_steps, sample_steps, G_file, output_folder, symm, kernel_type,
                 n_chains=1, seed=0, tol=1e-3, device=0, comm=None, timedata=None, spread_sampling=False, device_tables=False,
                 chi2_min_mode="chain_mean"):
        if chi2_min_mode not in ("chain_mean", "min"):
            raise ValueError("chi2_min_mode must be 'chain_mean' or 'min'")
        self.chi2_min_mode = chi2_min_mode
        self.par, self.N_omega = int(par), int(N_omega)
        self.omega_0, self.omega_m, self.d_omega, self.d_omega_h = float(omega_0), float(omega_m), float(d_omega), float(d_omega_h)
        self.theta_0, self.f_anneal, self.f_final = float(theta_0), float(f_anneal), float(f_final)
        self.a1, self.a2 = float(a1), float(a2)
        self.N_anneal, self.anneal_steps, self.sample_steps = int(N_anneal), int(anneal_steps), int(sample_steps)
        self.G_file, self.output_folder = G_file, output_folder
        self.symm, self.kernel_type = int(symm), str(kernel_type)
        self.tol = tol                                              # free:27
        self.n_chains, self.seed, self.device, self.comm = n_chains, seed, device, comm
        self.timedata = timedata
        # no counterpart in the one-chain reference: the final sampling's sample_steps sweeps are shared out over all chains
        # (each runs ceil(sample_steps / chains) sweeps, at least MIN_SPREAD_STEPS); equilibration stages are not shortened
        self.spread_sampling = bool(spread_sampling)
        self.device_tables = bool(device_tables)                    # K table built on the GPU (sac_build_table, row f1)
        self.chi2_anneal = np.zeros(self.N_anneal)                  # free:160
        self.h = None
        self.is_root = comm is None or comm.rank == 0

    # ---- input ---------------------------------------------------------------------------
    @classmethod
    def from_infile(cls, path="in_free.in", **kw):
        """free:1163-1179 (readdlm: whitespace separated, trailing comment block ignored)."""
        with open(path) as f:
            rows = [ln.split() for ln in f]
        par = int(float(rows[0][0]))
        N_w, w0, wm, dw, dwh = (float(x) for x in rows[1][:5])
        th0, fa, ff, a1, a2 = (float(x) for x in rows[2][:5])
        Na, asteps, ssteps = (int(float(x)) for x in rows[3][:3])
        G_file, out = rows[4][0], rows[4][1]
        symm, ktype = int(float(rows[5][0])), rows[5][1]
        if ktype == "bosonic":
            w0, symm = 0.0, 0
        elif symm == 1:
            out += "_symm"
            w0 = 0.0
        return cls(par, N_w, w0, wm, dw, dwh, th0, fa, ff, a1, a2, Na, asteps, ssteps, G_file, out,
                   symm, ktype, **kw)

    # ---- STEP 1 (free:1206-1220) ---------------------------------------------------------
    def initialize(self):
        from .tin import read_tin
        td = self.timedata if self.timedata is not None else read_tin(self.G_file)   # read_G!
        self.td = td
        self.prob = FreeProblem(td, self.N_omega, self.omega_0, self.omega_m, self.d_omega,
                                par=self.par, symm=self.symm, kernel_type=self.kernel_type)
        offset = 0 if self.comm is None else self.comm.rank * self.n_chains
        self.h = self.prob.make_handle(n_chains=self.n_chains, seed=self.seed, chain_offset=offset,
                                       device=self.device, device_tables=self.device_tables)
        if self.comm is not None:
            self.comm.attach(self.h)                                # RCCL groups: the merge goes behind the C ABI
        self._merge()
        return self

    def _merge(self):
        st = self.h.stats()                                         # over all ranks already when the handle merges
        if self.comm is not None and not self.h.merged_over_ranks:
            st = self.comm.merge_stats(st, n_chains=self.n_chains)
        self.sampled_chi2, self.chi2_min_all = st.sampled_chi2, st.chi2_min
        self.chi2_min = st.chi2_min_mean if self.chi2_min_mode == "chain_mean" else st.chi2_min
        self.accept_rates = np.array(st.accept_rates[:])
        self.update_windows = np.array(st.windows[:])
        return st

    # ---- L2 stages (GPU) -----------------------------------------------------------------
    def adjust_windows(self, steps, theta):                         # free:866
        self.h.adjust_windows(steps, theta)

    def sample(self, steps, theta):                                 # free:903
        self.h.sample(steps, theta)
        self._merge()

    def sampled_spec(self):
        """self.sampled_spec (free:59) merged over chains (and ranks)."""
        if self.comm is None or self.h.merged_over_ranks:
            return self.h.histogram()
        return self.comm.merge_histogram(self.h)

    # ---- output files --------------------------------------------------------------------
    def init_outfiles(self):                                        # free:931-955
        if not self.is_root:
            return
        import os
        os.makedirs(self.output_folder, exist_ok=True)
        with open(self.anneal_file, "w") as f:
            f.write("i,theta,chi2_min,chi2_avg\n")
        with open(self.accept_rate_file, "w") as f:
            f.write("i,ar_1,ar_2,ar_3,ar_4,ar_5,ar_6,omega_range_1,omega_range_2,omega_range_4\n")
        with open(self.sample_file, "w") as f:
            f.write("i,a,theta,chi2_min,chi2_avg\n")
        open(os.path.join(self.output_folder, "log.txt"), "w").close()

    @property
    def anneal_file(self):
        return self.output_folder + "/anneal.csv"

    @property
    def accept_rate_file(self):
        return self.output_folder + "/accept_rate.csv"

    @property
    def sample_file(self):
        return self.output_folder + "/sample.csv"

    def write_log(self, message, mode="a"):                         # free:1150-1154
        if not self.is_root:
            return
        import datetime
        now = datetime.datetime.now().isoformat(timespec="milliseconds")
        with open(self.output_folder + "/log.txt", mode) as f:
            f.write(f"{now} - {message}\n")

    def write_spec(self, n):
        """free:959-1019: multiply the normalisation back, rebin the fine histogram to d_omega_h
        (inclusive window of conversion+1 fine bins, i.e. neighbouring coarse bins share one fine
        bin, as in the reference), trim the empty tail, write sw%03d.csv."""
        spec = self.sampled_spec().copy()
        if not self.is_root:
            return None
        p = self.prob
        conv = int(round(self.d_omega_h / self.d_omega))            # free:964
        N_h = int((p.wi_m - p.wi_0) / conv)                         # free:965
        i = np.arange(p.wi_0, p.wi_m + 1)
        f = 1.0 + np.exp(-self.td.beta * (self.d_omega * i)) if self.kernel_type == "bosonic" else 1.0
        spec[: len(i)] = spec[: len(i)] * (self.td.norm * np.pi) / f      # free:969-975
        for k in range(1, N_h + 1):                                 # free:979-982 (in place, 1-based)
            spec[k - 1] = spec[(k - 1) * conv: k * conv + 1].sum() / self.d_omega_h
        j0, jf = 1, N_h
        for k in range(N_h - 1, -1, -1):                            # free:987-992
            jf = k
            if k >= 1 and spec[k - 1] > 1e-10:
                break
        base = p.wi_0 * self.d_omega
        with open(f"{self.output_folder}/sw{n:03d}.csv", "w") as fh:
            fh.write("omega,S\n")
            if self.kernel_type == "bosonic":                       # free:1002-1007
                for k in range(jf, j0 - 1, -1):
                    w = self.d_omega_h * (k - 1) + base
                    fh.write(f"{jl_float(-w)},{jl_float(spec[k - 1] * np.exp(-self.td.beta * w))}\n")
            for k in range(j0, jf + 1):                             # free:1009-1013
                w = self.d_omega_h * (k - 1) + base
                fh.write(f"{jl_float(w)},{jl_float(spec[k - 1])}\n")
        return spec[:N_h]

    # ---- schedule (free:1036-1145) -------------------------------------------------------
    def fast_anneal(self):                                          # free:1036-1048
        for i in range(1, 11):
            theta = self.theta_0 * (11 - i) ** 2
            self.adjust_windows(self.anneal_steps // 2, theta)
            self.sample(self.anneal_steps // 2, theta)

    def mean_gbar(self):
        """Gbar (eigenbasis) averaged over this handle's chains (one chain: the reference's self.Gbar)."""
        g = self.h.gbar_mean()                                      # one call: summed on the device (over ranks too when merged)
        if self.comm is not None and not self.h.merged_over_ranks:
            tot = self.comm.sum_vector(np.concatenate([g * self.n_chains, [float(self.n_chains)]]))
            return tot[:-1] / tot[-1]
        return g

    def main_anneal(self, wr=False, on_stage=None):                 # free:1051-1094
        theta = self.theta_0
        n_tau = self.td.n_tau
        for i in range(1, self.N_anneal + 1):
            steps = self.anneal_steps
            self.adjust_windows(steps, theta)
            self.sample(steps, theta)
            if on_stage is not None:                                # cross_val.jl:105-108 hooks in here
                on_stage(i, theta)
            if self.is_root:
                with open(self.anneal_file, "a") as f:              # free:1064-1068
                    vals = [float(i), _jl_round(theta, 8), _jl_round(self.chi2_min / n_tau, 4),
                            _jl_round(self.sampled_chi2 / n_tau, 4)]
                    f.write(",".join(jl_float(v) for v in vals) + "\n")
                with open(self.accept_rate_file, "a") as f:         # free:1070-1076
                    vals = list(self.accept_rates) + list(self.update_windows[[0, 1, 3]] * self.d_omega)
                    f.write(",".join([jl_float(float(i))] + [jl_float(_jl_round(v, 8)) for v in vals]) + "\n")
            self.chi2_anneal[i - 1] = self.sampled_chi2             # free:1078
            if wr:
                self.write_spec(i)
            if (self.sampled_chi2 - self.chi2_min) < (self.tol * n_tau):   # free:1087
                return
            theta /= self.f_anneal

    def optimal_theta(self):                                        # free:1249-1252
        a_vals = (self.chi2_anneal - self.chi2_min) / np.sqrt(2 * self.chi2_min)
        theta_vals = self.theta_0 / (self.f_anneal ** np.arange(self.N_anneal))
        return float(theta_vals[np.argmin(np.abs(a_vals - self.a2))])

    def final_anneal(self, theta_opt):                              # free:1102-1145
        n_tau = self.td.n_tau
        for i in range(1, 11):
            theta = theta_opt * (11 - i)
            steps = self.anneal_steps * i
            self.adjust_windows(steps, theta)
            self.sample(steps, theta)
        n_final = 1 if self.a1 == self.a2 else 20
        theta = theta_opt
        steps = self.sample_steps
        if self.spread_sampling:
            total = self.n_chains * (1 if self.comm is None else self.comm.world)
            steps = max(MIN_SPREAD_STEPS, -(-self.sample_steps // total))
        for n in range(1, n_final + 1):
            self.sample(steps, theta)
            a = (self.sampled_chi2 - self.chi2_min) / np.sqrt(2 * self.chi2_min)   # free:1128
            if self.is_root:
                with open(self.sample_file, "a") as f:              # free:1131-1133
                    vals = [float(n - 1), a, theta, self.chi2_min / n_tau, self.sampled_chi2 / n_tau]
                    f.write(",".join(jl_float(v) for v in vals) + "\n")
            self.write_spec(n - 1)
            if a < self.a1:
                return
            theta /= self.f_final

    def run(self, in_file=None):                                    # free:1160-1261
        import os
        import shutil
        if self.is_root:
            os.makedirs(self.output_folder, exist_ok=True)
            if in_file and os.path.exists(in_file):
                shutil.copyfile(in_file, self.output_folder + "/in_free.in")
            if self.timedata is None and os.path.exists(self.G_file):
                shutil.copyfile(self.G_file, self.output_folder + "/t.in")
        self.init_outfiles()
        self.write_log("Beginning Initialization.", "a")
        self.initialize()
        self.h.set_state(windows=np.full(4, self.prob.window))      # free:1218
        self.write_log("Initialization Finished.")
        self.write_log("Beginning Equillibration.")
        self.fast_anneal()
        self.write_log("Equillibration Finished.")
        self.write_log("Beginning Main Anneal.")
        self.main_anneal(False)
        self.write_log("Main Anneal Finished.")
        self.write_log("Beginning Final Anneal.")
        theta_opt = self.optimal_theta()
        self.h.set_state(windows=np.full(4, self.prob.window))      # free:1254
        self.final_anneal(theta_opt)
        self.write_log("Final Anneal Finished.")
        return theta_opt
