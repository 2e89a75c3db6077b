"""ctypes binding of include/sac_hip.h (the same entry points a Julia `ccall` would bind)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAC_HIP_LIB") or os.path.join(_HERE, "libsac_hip.so")   # SAC_HIP_LIB: another build of the same library
_LIB = None

EXPORTS = [
    "sac_last_error", "sac_build_id", "sac_create", "sac_destroy", "sac_upload_problem", "sac_set_state",
    "sac_get_state", "sac_init_gbar", "sac_get_gbar", "sac_set_gbar", "sac_adjust_windows",
    "sac_sample", "sac_get_stats", "sac_get_chain_stats", "sac_get_histogram",
    "sac_get_histogram_sum_device", "sac_set_move_log", "sac_get_move_log",
    "sac_upload_problem_peak", "sac_sample_bin", "sac_set_indiv_update", "sac_zero_accumulators",
    "sac_get_peak_stats", "sac_get_histogram_peak", "sac_get_bounds", "sac_copy_chain", "sac_get_chain_chi2", "sac_replay",
    "sac_edge_create", "sac_edge_destroy", "sac_edge_upload_problem", "sac_edge_set_state", "sac_edge_get_state",
    "sac_edge_init_gbar", "sac_edge_run_bins", "sac_edge_get_results", "sac_edge_get_accept_rates",
    "sac_edge_get_histogram", "sac_edge_get_avg", "sac_edge_get_gbar", "sac_edge_set_move_log",
    "sac_edge_get_move_log",
    "sac_build_table", "sac_build_table_noalloc", "sac_get_table", "sac_upload_data",
    "sac_edge_build_tables", "sac_edge_upload_data", "sac_edge_get_tables",
    "sac_get_gbar_mean", "sac_edge_get_gbar_mean", "sac_edge_get_move_means",
    "sac_edge_set_targets_from_chi2_min", "sac_edge_set_chain_targets", "sac_edge_get_first_passages", "sac_comm_unique_id", "sac_comm_init", "sac_comm_info", "sac_edge_comm_init", "sac_edge_comm_info",
    "sac_set_chains_per_wave", "sac_get_chains_per_wave", "sac_set_proposals_per_step", "sac_get_proposals_per_step", "sac_replay_peak", "sac_edge_replay", "sac_validation_chi2", "sac_edge_validation_chi2",
]

KIND_FREE, KIND_PEAK = 0, 1

HIST_REFERENCE, HIST_ACCUMULATE = 0, 1


class SacError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("kind", C.c_int32), ("par", C.c_int32), ("symm", C.c_int32), ("n_tau", C.c_int32),
                ("n_omega", C.c_int32), ("n_chains", C.c_int32), ("n_grid", C.c_int64),
                ("wi_0", C.c_int64), ("wi_m", C.c_int64), ("seed", C.c_uint64),
                ("chain_offset", C.c_uint32), ("device", C.c_int32), ("hist_mode", C.c_int32),
                ("hist_copies", C.c_int32), ("sweeps_per_launch", C.c_int32), ("n_peak", C.c_int32),
                ("fix_edge", C.c_int32), ("bosonic", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("sampled_chi2", C.c_double), ("chi2_min", C.c_double),
                ("accept_rates", C.c_double * 6), ("windows", C.c_double * 4),
                ("proposals", C.c_uint64), ("evaluated", C.c_uint64), ("accepted", C.c_uint64),
                ("accepted_moves", C.c_uint64), ("sweeps", C.c_uint64),
                ("columns_gathered", C.c_uint64),
                ("last_stage_ms", C.c_double), ("last_kernel_ms", C.c_double),
                ("last_kernel_launches", C.c_int64),
                ("chi2_min_mean", C.c_double), ("n_chains_total", C.c_int64)]


class PeakStats(C.Structure):
    _fields_ = [("sampled_chi2", C.c_double), ("chi2_min", C.c_double),
                ("accept_rates", C.c_double * 11), ("windows", C.c_double * 11),
                ("edge_res", (C.c_double * 2) * 3), ("bounds", C.c_double * 4), ("weights", C.c_double * 4),
                ("proposals", C.c_uint64), ("evaluated", C.c_uint64), ("accepted", C.c_uint64),
                ("accepted_moves", C.c_uint64), ("sweeps", C.c_uint64), ("columns_gathered", C.c_uint64),
                ("last_stage_ms", C.c_double), ("last_kernel_ms", C.c_double),
                ("last_kernel_launches", C.c_int64),
                ("chi2_min_mean", C.c_double), ("n_chains_total", C.c_int64)]


class EdgeConfig(C.Structure):
    _fields_ = [("n_tau", C.c_int32), ("n_e", C.c_int32), ("n_omega", C.c_int32), ("n_rl", C.c_int32),
                ("double_in", C.c_int32), ("fix_edge", C.c_int32), ("n_chains", C.c_int32), ("device", C.c_int32),
                ("wi_0", C.c_int64), ("n_grid", C.c_int64), ("A_c", C.c_double), ("A_r", C.c_double),
                ("d0", C.c_double), ("w_floor", C.c_double * 2), ("w_m", C.c_double), ("d_omega", C.c_double),
                ("c", C.c_double * 2), ("seed", C.c_uint64), ("chain_offset", C.c_uint32),
                ("sweeps_per_launch", C.c_int32), ("hist_copies", C.c_int32), ("reserved", C.c_int32)]


class EdgeResults(C.Structure):
    _fields_ = [("chi2_res", C.c_double * 3), ("edge_res", (C.c_double * 2) * 3), ("n0_res", (C.c_double * 2) * 3),
                ("wn0_res", (C.c_double * 2) * 3), ("amp_accept_rates", (C.c_double * 2) * 3),
                ("n0", C.c_double * 2), ("eps0", C.c_double * 2), ("dn0", C.c_double * 2), ("deps0", C.c_double * 2),
                ("chi2", C.c_double), ("chi2_min", C.c_double),
                ("sweeps", C.c_uint64), ("proposals", C.c_uint64), ("evaluated", C.c_uint64),
                ("accepted", C.c_uint64), ("lookups", C.c_uint64), ("violations", C.c_uint64),
                ("last_stage_ms", C.c_double),
                ("chi2_min_mean", C.c_double), ("n_chains_total", C.c_int64), ("chains_below_target", C.c_int64)]


COMM_ID_BYTES = 128     # SAC_COMM_ID_BYTES

KERNEL_TYPES = {"finiteT": 0, "zeroT": 1, "bosonic": 2}


class TableSpec(C.Structure):
    """sac_table_spec_t"""
    _fields_ = [("kernel_type", C.c_int32), ("mirror", C.c_int32), ("negate", C.c_int32), ("reserved", C.c_int32),
                ("i_first", C.c_int64), ("offset", C.c_double), ("d_omega", C.c_double), ("beta", C.c_double)]


def _table_spec(kernel_type, i_first, d_omega, beta, offset=0.0, mirror=0, negate=0):
    if kernel_type not in KERNEL_TYPES:
        raise ValueError("Invalid Kernel type.")
    return TableSpec(KERNEL_TYPES[kernel_type], int(mirror), int(negate), 0, int(i_first), float(offset), float(d_omega), float(beta))


PROPOSAL = np.dtype([("move", "<i4"), ("k", "<i4", 3), ("w_new", "<i8", 3), ("u", "<f8")])

PEAK_PROPOSAL = np.dtype([("n", "<i4"), ("transfer", "<i4"), ("k", "<i4", 4), ("w_new", "<i8", 4), ("A_new", "<f8", 4),
                          ("u", "<f8")])

EDGE_PROPOSAL = np.dtype([("side", "<i4"), ("k", "<i4"), ("n", "<i4"), ("reserved", "<i4"), ("w_offset", "<i8"), ("u", "<f8")])

MOVE_REC = np.dtype([("chi2_new", "<f8"), ("flags", "<i4"), ("move", "<i4")])


def load_library():
    """Loads libsac_hip.so; there is no fallback -- a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SacError(f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950); "
                       "sac_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.sac_last_error.restype = C.c_char_p
    L.sac_build_id.restype = C.c_char_p
    L.sac_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.sac_destroy.argtypes = [vp]
    L.sac_upload_problem.argtypes = [vp, dp, dp, dp]
    L.sac_set_state.argtypes = [vp, C.c_int32, ip, dp, dp]
    L.sac_get_state.argtypes = [vp, C.c_int32, ip, dp, dp]
    L.sac_init_gbar.argtypes = [vp]
    L.sac_get_gbar.argtypes = [vp, C.c_int32, dp, dp, dp]
    L.sac_set_gbar.argtypes = [vp, C.c_int32, dp, C.c_double, C.c_double]
    L.sac_adjust_windows.argtypes = [vp, C.c_int64, C.c_double]
    L.sac_sample.argtypes = [vp, C.c_int64, C.c_double]
    L.sac_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sac_get_chain_stats.argtypes = [vp, C.c_int32, dp, dp]
    L.sac_get_histogram.argtypes = [vp, dp]
    L.sac_get_histogram_sum_device.argtypes = [vp, vp, dp]
    L.sac_set_move_log.argtypes = [vp, C.c_int32, C.c_int64]
    L.sac_get_move_log.argtypes = [vp, vp, C.c_int64, ip]
    L.sac_upload_problem_peak.argtypes = [vp, dp, dp, dp, dp]
    L.sac_sample_bin.argtypes = [vp, C.c_int64, C.c_double]
    L.sac_set_indiv_update.argtypes = [vp, C.c_int32]
    L.sac_zero_accumulators.argtypes = [vp]
    L.sac_get_peak_stats.argtypes = [vp, C.POINTER(PeakStats)]
    L.sac_get_histogram_peak.argtypes = [vp, C.c_int32, dp]
    L.sac_get_bounds.argtypes = [vp, C.c_int32, dp]
    L.sac_copy_chain.argtypes = [vp, C.c_int32, C.c_int32]
    L.sac_replay.argtypes = [vp, C.c_int32, C.c_double, vp, C.c_int64, dp, vp]
    L.sac_get_chain_chi2.argtypes = [vp, dp, dp]
    L.sac_validation_chi2.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp]
    L.sac_edge_validation_chi2.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp]
    L.sac_replay_peak.argtypes = [vp, C.c_int32, C.c_double, vp, C.c_int64, dp, vp]
    L.sac_build_table.argtypes = [vp, C.c_int32, C.POINTER(TableSpec), dp, dp]
    L.sac_build_table_noalloc.argtypes = [vp, C.c_int32, C.POINTER(TableSpec), dp, dp]
    L.sac_get_table.argtypes = [vp, C.c_int32, dp]
    L.sac_upload_data.argtypes = [vp, dp, dp]
    L.sac_set_chains_per_wave.argtypes = [vp, C.c_int32]
    L.sac_get_chains_per_wave.argtypes = [vp, C.POINTER(C.c_int32)]
    L.sac_set_proposals_per_step.argtypes = [vp, C.c_int32]
    L.sac_get_proposals_per_step.argtypes = [vp, C.POINTER(C.c_int32)]
    L.sac_get_gbar_mean.argtypes = [vp, dp]
    L.sac_edge_get_gbar_mean.argtypes = [vp, dp]
    L.sac_edge_get_move_means.argtypes = [vp, dp, dp, dp]
    L.sac_edge_set_targets_from_chi2_min.argtypes = [vp, C.c_double]
    L.sac_edge_set_chain_targets.argtypes = [vp, dp]
    L.sac_edge_get_first_passages.argtypes = [vp, C.POINTER(C.c_int32)]
    i32p = C.POINTER(C.c_int32)
    L.sac_comm_unique_id.argtypes = [vp]
    L.sac_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
    L.sac_comm_info.argtypes = [vp, i32p, i32p, ip, ip]
    L.sac_edge_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
    L.sac_edge_comm_info.argtypes = [vp, i32p, i32p, ip, ip]
    L.sac_edge_build_tables.argtypes = [vp, C.c_int32, C.POINTER(TableSpec), dp, dp]
    L.sac_edge_upload_data.argtypes = [vp, dp, dp]
    L.sac_edge_get_tables.argtypes = [vp, C.c_int32, dp, dp, dp]
    pdp = C.POINTER(dp)
    L.sac_edge_create.argtypes = [C.POINTER(EdgeConfig), C.POINTER(vp)]
    L.sac_edge_destroy.argtypes = [vp]
    L.sac_edge_upload_problem.argtypes = [vp, pdp, pdp, pdp, dp, dp]
    L.sac_edge_set_state.argtypes = [vp, C.c_int32, dp, dp, dp, ip, dp]
    L.sac_edge_get_state.argtypes = [vp, C.c_int32, dp, dp, dp, ip, dp]
    L.sac_edge_init_gbar.argtypes = [vp]
    L.sac_edge_run_bins.argtypes = [vp, C.c_int64, C.c_int64, C.c_double]
    L.sac_edge_get_results.argtypes = [vp, C.c_int32, C.POINTER(EdgeResults)]
    L.sac_edge_get_accept_rates.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, dp]
    L.sac_edge_get_histogram.argtypes = [vp, C.c_int32, C.c_int32, dp]
    L.sac_edge_get_avg.argtypes = [vp, C.c_int32, dp, dp]
    L.sac_edge_get_gbar.argtypes = [vp, C.c_int32, dp, dp, dp]
    L.sac_edge_set_move_log.argtypes = [vp, C.c_int32, C.c_int64]
    L.sac_edge_replay.argtypes = [vp, C.c_int32, C.c_double, vp, C.c_int64, dp, C.c_int64, dp, vp]
    L.sac_edge_get_move_log.argtypes = [vp, vp, C.c_int64, ip]
    for name in EXPORTS:
        if name not in ("sac_last_error", "sac_build_id"):
            getattr(L, name).restype = C.c_int
    _LIB = L
    return L


def build_id():
    """sac_build_id(): fingerprint of the sources the loaded library was built from."""
    return load_library().sac_build_id().decode()


def comm_unique_id():
    """sac_comm_unique_id (ncclGetUniqueId): 128 bytes rank 0 hands to the other ranks."""
    L = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = L.sac_comm_unique_id(buf)
    if rc != 0:
        raise SacError(f"libsac_hip error {rc}: {L.sac_last_error().decode()}")
    return buf.raw


def _comm_info(fn, h, check):
    r, w, m, a = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
    check(fn(h, C.byref(r), C.byref(w), C.byref(m), C.byref(a)))
    return dict(rank=r.value, world=w.value, merges=m.value, allreduce_calls=a.value)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int64))


class SacHandle:
    """One GPU's worth of chains of one problem (thin object wrapper over the C ABI)."""

    def __init__(self, n_tau, n_omega, n_grid, wi_0, wi_m, n_chains=1, par=1, symm=0, seed=0,
                 chain_offset=0, device=0, hist_mode=HIST_REFERENCE, hist_copies=0,
                 sweeps_per_launch=0, kind=KIND_FREE, n_peak=0, fix_edge=0, bosonic=0):
        self.L = load_library()
        self.cfg = Config(kind=kind, par=par, symm=symm, n_tau=n_tau, n_omega=n_omega,
                          n_chains=n_chains, n_grid=n_grid, wi_0=wi_0, wi_m=wi_m, seed=seed,
                          chain_offset=chain_offset, device=device, hist_mode=hist_mode,
                          hist_copies=hist_copies, sweeps_per_launch=sweeps_per_launch,
                          n_peak=n_peak, fix_edge=fix_edge, bosonic=bosonic)
        self.h = C.c_void_p()
        self._check(self.L.sac_create(C.byref(self.cfg), C.byref(self.h)))
        self.n_tau, self.n_omega, self.n_grid, self.n_chains = n_tau, n_omega, n_grid, n_chains
        self.kind = kind
        self.n_state = 2 * (n_omega + n_peak) if kind == KIND_PEAK else n_omega
        self.n_windows = 11 if kind == KIND_PEAK else 4
        self.n_rates = 11 if kind == KIND_PEAK else 6
        self.merged_over_ranks = False

    def _check(self, rc):
        if rc != 0:
            raise SacError(f"libsac_hip error {rc}: {self.L.sac_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.sac_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_problem(self, K, G, sigma_inv):
        K = np.ascontiguousarray(K, dtype=np.float64)
        if K.shape != (self.n_grid, self.n_tau):
            raise SacError(f"K has shape {K.shape}, expected {(self.n_grid, self.n_tau)}")
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        if G.shape != (self.n_tau,) or sigma_inv.shape != (self.n_tau,):
            raise SacError("G and sigma_inv must have n_tau entries")
        self._check(self.L.sac_upload_problem(self.h, _dp(K), _dp(G), _dp(sigma_inv)))

    def set_state(self, wi=None, A=None, windows=None, chain=-1):
        wi = None if wi is None else np.ascontiguousarray(wi, dtype=np.int64)
        A = None if A is None else np.ascontiguousarray(A, dtype=np.float64)
        windows = None if windows is None else np.ascontiguousarray(windows, dtype=np.float64)
        for a, n in ((wi, self.n_state), (A, self.n_state), (windows, self.n_windows)):
            if a is not None and a.shape != (n,):
                raise SacError(f"state array has shape {a.shape}, expected ({n},)")
        self._check(self.L.sac_set_state(self.h, chain, _ip(wi), _dp(A), _dp(windows)))

    def get_state(self, chain=0):
        wi = np.empty(self.n_state, dtype=np.int64)
        A = np.empty(self.n_state)
        w = np.empty(self.n_windows)
        self._check(self.L.sac_get_state(self.h, chain, _ip(wi), _dp(A), _dp(w)))
        return wi, A, w

    def init_gbar(self):
        self._check(self.L.sac_init_gbar(self.h))

    def get_gbar(self, chain=0):
        g = np.empty(self.n_tau)
        c, m = C.c_double(), C.c_double()
        self._check(self.L.sac_get_gbar(self.h, chain, _dp(g), C.byref(c), C.byref(m)))
        return g, c.value, m.value

    def gbar_mean(self):
        """Gbar averaged over the chains (all ranks' after a merged sample stage), eigenbasis, n_tau entries."""
        g = np.empty(self.n_tau)
        self._check(self.L.sac_get_gbar_mean(self.h, _dp(g)))
        return g

    def set_gbar(self, gbar, chi2, chi2_min, chain=-1):
        gbar = np.ascontiguousarray(gbar, dtype=np.float64)
        self._check(self.L.sac_set_gbar(self.h, chain, _dp(gbar), chi2, chi2_min))

    def set_chains_per_wave(self, chains_per_wave):
        """0 = the library chooses; 1, 2, 4 = Markov chains sharing one wavefront (same chains, other schedule)."""
        self._check(self.L.sac_set_chains_per_wave(self.h, int(chains_per_wave)))

    def chains_per_wave(self):
        """Chains per wavefront the last sweep launch ran with."""
        out = C.c_int32(0)
        self._check(self.L.sac_get_chains_per_wave(self.h, C.byref(out)))
        return out.value

    def set_proposals_per_step(self, proposals_per_step):
        """0 = the library chooses; 1, 2, 4 = consecutive proposals of one chain a wavefront evaluates per step (runs with
        few chains: same chain, shorter latency)."""
        self._check(self.L.sac_set_proposals_per_step(self.h, int(proposals_per_step)))

    def proposals_per_step(self):
        """Proposals per step the last sweep launch ran with."""
        out = C.c_int32(0)
        self._check(self.L.sac_get_proposals_per_step(self.h, C.byref(out)))
        return out.value

    def comm_init(self, unique_id, rank, world):
        """sac_comm_init: from here on every stage call ends with the merge over ranks (one fused all-reduce(sum) + one
        all-reduce(min) on the handle's stream) and the reduced getters answer for all ranks' chains."""
        if len(unique_id) != COMM_ID_BYTES:
            raise SacError(f"unique id has {len(unique_id)} bytes, expected {COMM_ID_BYTES}")
        self._check(self.L.sac_comm_init(self.h, C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES), int(rank), int(world)))
        self.merged_over_ranks = True

    def comm_info(self):
        return _comm_info(self.L.sac_comm_info, self.h, self._check)

    def adjust_windows(self, steps, theta):
        self._check(self.L.sac_adjust_windows(self.h, int(steps), float(theta)))

    def sample(self, steps, theta):
        self._check(self.L.sac_sample(self.h, int(steps), float(theta)))

    def stats(self):
        s = Stats()
        self._check(self.L.sac_get_stats(self.h, C.byref(s)))
        return s

    def chain_stats(self, chain=0):
        o = np.empty(6)
        ar = np.empty(self.n_rates)
        self._check(self.L.sac_get_chain_stats(self.h, chain, _dp(o), _dp(ar)))
        return dict(sampled_chi2=o[0], chi2=o[1], chi2_min=o[2], sweeps=int(o[3]), accept_rates=ar,
                    edge_mean=o[4:6].copy())

    # ---- SAC_KIND_PEAK ----------------------------------------------------------------------
    def build_table(self, which, kernel_type, tau, beta, U, i_first, d_omega, offset=0.0, mirror=0, negate=0, noalloc=False):
        """K table built on the device (sac_build_table): which = 0 K / Kp, 1 Kn; U = the reference's `cov`.
        noalloc: the entry point that stages tau / U in the library's file-scope device array (no hipMalloc)."""
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        U = np.ascontiguousarray(U, dtype=np.float64)
        if tau.shape != (self.n_tau,) or U.shape != (self.n_tau, self.n_tau):
            raise SacError("tau must have n_tau entries and U n_tau x n_tau")
        sp = _table_spec(kernel_type, i_first, d_omega, beta, offset, mirror, negate)
        fn = self.L.sac_build_table_noalloc if noalloc else self.L.sac_build_table
        self._check(fn(self.h, int(which), C.byref(sp), _dp(tau), _dp(U)))

    def get_table(self, which=0):
        out = np.empty((self.n_grid, self.n_tau))
        self._check(self.L.sac_get_table(self.h, int(which), _dp(out)))
        return out

    def upload_data(self, G, sigma_inv):
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        if G.shape != (self.n_tau,) or sigma_inv.shape != (self.n_tau,):
            raise SacError("G and sigma_inv must have n_tau entries")
        self._check(self.L.sac_upload_data(self.h, _dp(G), _dp(sigma_inv)))

    def upload_problem_peak(self, Kp, Kn, G, sigma_inv):
        Kp = np.ascontiguousarray(Kp, dtype=np.float64)
        Kn = None if Kn is None else np.ascontiguousarray(Kn, dtype=np.float64)
        for K in (Kp, Kn):
            if K is not None and K.shape != (self.n_grid, self.n_tau):
                raise SacError(f"K has shape {K.shape}, expected {(self.n_grid, self.n_tau)}")
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        self._check(self.L.sac_upload_problem_peak(self.h, _dp(Kp), _dp(Kn), _dp(G), _dp(sigma_inv)))

    def sample_bin(self, steps, theta):
        self._check(self.L.sac_sample_bin(self.h, int(steps), float(theta)))

    def set_indiv_update(self, on):
        self._check(self.L.sac_set_indiv_update(self.h, int(bool(on))))

    def zero_accumulators(self):
        self._check(self.L.sac_zero_accumulators(self.h))

    def peak_stats(self):
        s = PeakStats()
        self._check(self.L.sac_get_peak_stats(self.h, C.byref(s)))
        return s

    def histogram_peak(self, which):
        out = np.empty(self.n_grid)
        self._check(self.L.sac_get_histogram_peak(self.h, int(which), _dp(out)))
        return out

    def replay(self, props, theta, chain=0):
        """props: array of PROPOSAL records.  Returns (chi2_new[n], accepted[n])."""
        props = np.ascontiguousarray(props, dtype=PROPOSAL)
        n = len(props)
        c2, acc = np.empty(n), np.empty(n, dtype=np.int32)
        self._check(self.L.sac_replay(self.h, chain, float(theta), props.ctypes.data, n, _dp(c2), acc.ctypes.data))
        return c2, acc

    def validation_chi2(self, cov_this, td_other, gbar_mean=None, per_chain=False):
        """calc_cv.jl:29-58 on the device: chi2_v (not divided by N_tau) of the chain-mean Gbar (or of gbar_mean) against the
        other half's data td_other; with per_chain also every chain's own."""
        U_t = np.ascontiguousarray(cov_this, dtype=np.float64)
        U_o = np.ascontiguousarray(td_other.cov, dtype=np.float64)
        G_o = np.ascontiguousarray(td_other.G, dtype=np.float64)
        s_o = np.ascontiguousarray(td_other.sigma_inv, dtype=np.float64)
        gm = None if gbar_mean is None else np.ascontiguousarray(gbar_mean, dtype=np.float64)
        out = C.c_double(0.0)
        chains = np.empty(self.cfg.n_chains) if per_chain else None
        self._check(self.L.sac_validation_chi2(self.h, _dp(U_t), _dp(U_o), _dp(G_o), _dp(s_o), _dp(gm), C.byref(out), _dp(chains)))
        return (out.value, chains) if per_chain else out.value

    def replay_peak(self, props, theta, chain=0):
        """props: array of PEAK_PROPOSAL records.  Returns (chi2_new[n], accepted[n])."""
        props = np.ascontiguousarray(props, dtype=PEAK_PROPOSAL)
        n = len(props)
        c2, acc = np.empty(n), np.empty(n, dtype=np.int32)
        self._check(self.L.sac_replay_peak(self.h, chain, float(theta), props.ctypes.data, n, _dp(c2), acc.ctypes.data))
        return c2, acc

    def chain_chi2(self):
        a, b = np.empty(self.cfg.n_chains), np.empty(self.cfg.n_chains)
        self._check(self.L.sac_get_chain_chi2(self.h, _dp(a), _dp(b)))
        return a, b

    def copy_chain(self, src, dst):
        self._check(self.L.sac_copy_chain(self.h, int(src), int(dst)))

    def bounds(self, chain=0):
        b = np.empty(4)
        self._check(self.L.sac_get_bounds(self.h, chain, _dp(b)))
        return b

    def histogram(self):
        out = np.empty(self.n_grid)
        self._check(self.L.sac_get_histogram(self.h, _dp(out)))
        return out

    def histogram_sum_to_device(self, device_ptr):
        d = C.c_double()
        self._check(self.L.sac_get_histogram_sum_device(self.h, C.c_void_p(device_ptr), C.byref(d)))
        return d.value

    def set_move_log(self, chain, capacity):
        self._check(self.L.sac_set_move_log(self.h, chain, capacity))

    def move_log(self, capacity):
        out = np.zeros(capacity, dtype=MOVE_REC)
        n = C.c_int64()
        self._check(self.L.sac_get_move_log(self.h, out.ctypes.data, capacity, C.byref(n)))
        return out[:min(n.value, capacity)], n.value


class SacEdgeHandle:
    """ctypes view of one sac_edge_t (edge parameterisation).  State arrays are [N_w+1, 2] numpy arrays
    (the reference's shape); they cross the ABI column-major (side after side)."""

    def __init__(self, n_tau, n_e, n_omega, n_grid, wi_0, d_omega, w_m, w_floor, c, n_rl=1, double_in=0,
                 fix_edge=0, A_c=0.0, A_r=1.0, d0=0.0, n_chains=1, seed=0, chain_offset=0, device=0,
                 sweeps_per_launch=0, hist_copies=0):
        self.L = load_library()
        self.cfg = EdgeConfig(int(n_tau), int(n_e), int(n_omega), int(n_rl), int(double_in), int(fix_edge),
                              int(n_chains), int(device), int(wi_0), int(n_grid), float(A_c), float(A_r), float(d0),
                              (C.c_double * 2)(*[float(x) for x in w_floor]), float(w_m), float(d_omega),
                              (C.c_double * 2)(float(c), float(c)), int(seed), int(chain_offset),
                              int(sweeps_per_launch), int(hist_copies), 0)
        self.n_tau, self.n_e, self.n_omega, self.n_grid, self.n_rl = int(n_tau), int(n_e), int(n_omega), int(n_grid), int(n_rl)
        self.h = C.c_void_p()
        self._check(self.L.sac_edge_create(C.byref(self.cfg), C.byref(self.h)))
        self._keep = None
        self.merged_over_ranks = False

    def comm_init(self, unique_id, rank, world):
        """sac_edge_comm_init: see SacHandle.comm_init."""
        if len(unique_id) != COMM_ID_BYTES:
            raise SacError(f"unique id has {len(unique_id)} bytes, expected {COMM_ID_BYTES}")
        self._check(self.L.sac_edge_comm_init(self.h, C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES), int(rank), int(world)))
        self.merged_over_ranks = True

    def comm_info(self):
        return _comm_info(self.L.sac_edge_comm_info, self.h, self._check)

    def _check(self, rc):
        if rc != 0:
            raise SacError(f"libsac_hip error {rc}: {self.L.sac_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.sac_edge_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_tables(self, side, kernel_type, tau, beta, U, i_first, d_omega, mirror=0, negate=0):
        """K, dK, d2K of one side built on the device (sac_edge_build_tables)."""
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        U = np.ascontiguousarray(U, dtype=np.float64)
        if tau.shape != (self.n_tau,) or U.shape != (self.n_tau, self.n_tau):
            raise SacError("tau must have n_tau entries and U n_tau x n_tau")
        sp = _table_spec(kernel_type, i_first, d_omega, beta, 0.0, mirror, negate)
        self._check(self.L.sac_edge_build_tables(self.h, int(side), C.byref(sp), _dp(tau), _dp(U)))

    def upload_data(self, G, sigma_inv):
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        self._check(self.L.sac_edge_upload_data(self.h, _dp(G), _dp(sigma_inv)))

    def get_tables(self, side):
        out = [np.empty((self.n_grid, self.n_tau)) for _ in range(3)]
        self._check(self.L.sac_edge_get_tables(self.h, int(side), _dp(out[0]), _dp(out[1]), _dp(out[2])))
        return tuple(out)

    def upload_problem(self, tables, G, sigma_inv):
        """tables: per side (K, dK, d2K), each [n_grid, n_tau] C-order (= Julia's [N_tau, N_wi] slices)."""
        if len(tables) != self.n_rl:
            raise SacError(f"{len(tables)} table sets for n_rl = {self.n_rl}")
        keep = [[np.ascontiguousarray(t, dtype=np.float64) for t in side] for side in tables]
        for side in keep:
            for t in side:
                if t.shape != (self.n_grid, self.n_tau):
                    raise SacError(f"table has shape {t.shape}, expected {(self.n_grid, self.n_tau)}")
        dp = C.POINTER(C.c_double)
        arrs = [(dp * 2)(*([_dp(side[i]) for side in keep] + [None] * (2 - len(keep)))) for i in range(3)]
        G = np.ascontiguousarray(G, dtype=np.float64)
        sigma_inv = np.ascontiguousarray(sigma_inv, dtype=np.float64)
        self._check(self.L.sac_edge_upload_problem(self.h, arrs[0], arrs[1], arrs[2], _dp(G), _dp(sigma_inv)))

    @staticmethod
    def _cm(a, dtype=np.float64):
        return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=dtype).T).ravel()

    def set_state(self, w=None, A=None, dw=None, nm=None, scal=None, chain=-1):
        n1 = self.n_omega + 1
        for a in (w, A, dw, nm):
            if a is not None and np.asarray(a).shape != (n1, 2):
                raise SacError(f"state array has shape {np.asarray(a).shape}, expected {(n1, 2)}")
        w, A, dw = self._cm(w), self._cm(A), self._cm(dw)
        nm = self._cm(nm, np.int64)
        scal = None if scal is None else np.ascontiguousarray(scal, dtype=np.float64)
        self._check(self.L.sac_edge_set_state(self.h, chain, _dp(w), _dp(A), _dp(dw), _ip(nm), _dp(scal)))

    def get_state(self, chain=0):
        n1 = self.n_omega + 1
        w, A, dw, scal = np.empty(2 * n1), np.empty(2 * n1), np.empty(2 * n1), np.empty(8)
        nm = np.empty(2 * n1, dtype=np.int64)
        self._check(self.L.sac_edge_get_state(self.h, chain, _dp(w), _dp(A), _dp(dw), _ip(nm), _dp(scal)))
        r = lambda a: a.reshape(2, n1).T.copy()
        return dict(w=r(w), A=r(A), dw=r(dw), nm=r(nm), scal=scal)

    def init_gbar(self):
        self._check(self.L.sac_edge_init_gbar(self.h))

    def run_bins(self, steps, bins, theta):
        self._check(self.L.sac_edge_run_bins(self.h, int(steps), int(bins), float(theta)))

    def results(self, chain=-1):
        r = EdgeResults()
        self._check(self.L.sac_edge_get_results(self.h, chain, C.byref(r)))
        f = lambda a: np.array([list(x) for x in a])
        return dict(chi2_res=np.array(r.chi2_res[:]), edge_res=f(r.edge_res), n0_res=f(r.n0_res), wn0_res=f(r.wn0_res),
                    amp_ar=f(r.amp_accept_rates), n0=np.array(r.n0[:]), eps0=np.array(r.eps0[:]),
                    dn0=np.array(r.dn0[:]), deps0=np.array(r.deps0[:]), chi2=r.chi2, chi2_min=r.chi2_min,
                    chi2_min_mean=r.chi2_min_mean, n_chains_total=r.n_chains_total, chains_below_target=r.chains_below_target,
                    sweeps=r.sweeps, proposals=r.proposals, evaluated=r.evaluated, accepted=r.accepted,
                    lookups=r.lookups, violations=r.violations, last_stage_ms=r.last_stage_ms)

    def accept_rates(self, m, side, chain=-1):
        a = np.empty(self.n_omega + 1)
        self._check(self.L.sac_edge_get_accept_rates(self.h, chain, m, side, _dp(a)))
        return a

    def histogram(self, which, side):
        o = np.empty(self.n_grid)
        self._check(self.L.sac_edge_get_histogram(self.h, which, side, _dp(o)))
        return o

    def avg(self, chain=-1):
        n1 = self.n_omega + 1
        w, A = np.empty(2 * n1), np.empty(2 * n1)
        self._check(self.L.sac_edge_get_avg(self.h, chain, _dp(w), _dp(A)))
        return w.reshape(2, n1).T.copy(), A.reshape(2, n1).T.copy()

    def set_targets_from_chi2_min(self, a_criterion):
        """Every chain's own target chi2_min_c + a sqrt(2 chi2_min_c) (edge:1718) for the first-passage bookkeeping of the
        second anneal; clears the marks."""
        self._check(self.L.sac_edge_set_targets_from_chi2_min(self.h, float(a_criterion)))

    def set_chain_targets(self, targets=None):
        t = None if targets is None else np.ascontiguousarray(targets, dtype=np.float64)
        if t is not None and t.shape != (self.cfg.n_chains,):
            raise SacError(f"targets has shape {t.shape}, expected ({self.cfg.n_chains},)")
        self._check(self.L.sac_edge_set_chain_targets(self.h, _dp(t)))

    def first_passages(self):
        """Per chain the 1-based index of the stage (since the targets were set) in which its stage average first fell below
        its target; 0 = not yet."""
        out = np.zeros(self.cfg.n_chains, dtype=np.int32)
        self._check(self.L.sac_edge_get_first_passages(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def move_means(self):
        """Chain means (all ranks' after a merged stage) of the per-delta windows and cluster sizes, [N_w+1, 2] each, and
        (lowest, highest) continuum frequency of side 1: what write_res prints (edge:1454-1506)."""
        n1 = self.n_omega + 1
        dw, nm, ext = np.empty(2 * n1), np.empty(2 * n1), np.empty(2)
        self._check(self.L.sac_edge_get_move_means(self.h, _dp(dw), _dp(nm), _dp(ext)))
        return dw.reshape(2, n1).T.copy(), nm.reshape(2, n1).T.copy(), ext

    def get_gbar(self, chain=0):
        g = np.empty(self.n_tau)
        c, m = C.c_double(), C.c_double()
        self._check(self.L.sac_edge_get_gbar(self.h, chain, _dp(g), C.byref(c), C.byref(m)))
        return g, c.value, m.value

    def gbar_mean(self):
        """Gbar averaged over the chains (all ranks' after a merged stage), eigenbasis, n_tau entries."""
        g = np.empty(self.n_tau)
        self._check(self.L.sac_edge_get_gbar_mean(self.h, _dp(g)))
        return g

    def validation_chi2(self, cov_this, td_other, gbar_mean=None, per_chain=False):
        """calc_cv.jl:29-58 on the device: chi2_v (not divided by N_tau) of the chain-mean Gbar (or of gbar_mean) against the
        other half's data td_other; with per_chain also every chain's own."""
        U_t = np.ascontiguousarray(cov_this, dtype=np.float64)
        U_o = np.ascontiguousarray(td_other.cov, dtype=np.float64)
        G_o = np.ascontiguousarray(td_other.G, dtype=np.float64)
        s_o = np.ascontiguousarray(td_other.sigma_inv, dtype=np.float64)
        gm = None if gbar_mean is None else np.ascontiguousarray(gbar_mean, dtype=np.float64)
        out = C.c_double(0.0)
        chains = np.empty(self.cfg.n_chains) if per_chain else None
        self._check(self.L.sac_edge_validation_chi2(self.h, _dp(U_t), _dp(U_o), _dp(G_o), _dp(s_o), _dp(gm), C.byref(out), _dp(chains)))
        return (out.value, chains) if per_chain else out.value

    def replay(self, props, w_new, theta, chain=0):
        """props: array of EDGE_PROPOSAL records, w_new: the new frequencies they index.  Returns (chi2_new[n], accepted[n])."""
        props = np.ascontiguousarray(props, dtype=EDGE_PROPOSAL)
        w_new = np.ascontiguousarray(w_new, dtype=np.float64)
        n = len(props)
        c2, acc = np.empty(n), np.empty(n, dtype=np.int32)
        self._check(self.L.sac_edge_replay(self.h, chain, float(theta), props.ctypes.data, n, _dp(w_new), len(w_new), _dp(c2),
                                           acc.ctypes.data))
        return c2, acc

    def set_move_log(self, chain, capacity):
        self._check(self.L.sac_edge_set_move_log(self.h, chain, capacity))

    def move_log(self, capacity):
        out = np.zeros(capacity, dtype=MOVE_REC)
        n = C.c_int64()
        self._check(self.L.sac_edge_get_move_log(self.h, out.ctypes.data, capacity, C.byref(n)))
        return out[: min(n.value, capacity)], n.value
