"""Cross-validation wrapper: mirror of sac/cross_validation/cross_val.jl (free_cv cv:6-144,
peak_cv cv:147-312, edge_cv cv:315-509) above the C ABI.  For replica `rep`, half `ab` it runs the sampler's main anneal
with tol = 1e-6 and appends [Theta, cov * Gbar] to GSAC.csv after every Theta stage (cv:105-108);
calc_cv.jl then scores half a's Gbar against half b's data.  With n_chains > 1 the row holds the
chain-averaged Gbar."""
import os
import shutil

import numpy as np

from .edge import EdgeSAC
from .free import FreeSAC, _jl_round, jl_float
from .peak import PeakSAC


def _append_gsac(path, theta, cov, gbar):
    row = [_jl_round(theta, 8)] + list(cov @ gbar)                  # cv:106
    with open(path, "a") as f:
        f.write(",".join(jl_float(float(v)) for v in row) + "\n")


def _fused_validation(s, validate_tin):
    """Row f3 of SURVEY section 8: the validation chi^2 of calc_cv.jl:29-58 taken on the device from the chains' Gbar at
    the end of every Theta stage (sac_validation_chi2), no GSAC.csv round trip.  Returns the hook's scorer: it appends
    (Theta, chi2_v / N_tau) to s.validation and to <output_folder>/validation.csv."""
    if validate_tin is None:
        return None
    from .tin import read_tin
    td_o = read_tin(validate_tin) if isinstance(validate_tin, str) else validate_tin
    s.validation = []
    path = s.output_folder + "/validation.csv"
    if s.is_root:
        with open(path, "w") as f:
            f.write("theta,chi2_val\n")

    def score(theta, gbar_mean=None):
        v = s.h.validation_chi2(s.td.cov, td_o, gbar_mean=gbar_mean) / s.td.n_tau
        s.validation.append((_jl_round(theta, 8), v))
        if s.is_root:
            with open(path, "a") as f:
                f.write(f"{jl_float(float(_jl_round(theta, 8)))},{jl_float(float(v))}\n")
    return score


def cv_score_fused(s_a, s_b):
    """calc_cv.jl:82-107 from two fused runs: s_a validated against half b's data and s_b against half a's."""
    va, vb = np.array(s_a.validation), np.array(s_b.validation)
    n = min(len(va), len(vb))
    return va[:n, 0], 0.5 * (va[:n, 1] + vb[:n, 1])


def free_cv(rep, ab, in_file="in_free.in", validate_tin=None, **kw):
    """cv:6-144.  Returns the FreeSAC object (its output_folder holds GSAC.csv, anneal.csv, ...)."""
    s = FreeSAC.from_infile(in_file, **kw)
    G_folder = s.G_file                                             # row 5 holds the folder here (cv:14)
    s.G_file = "%s/t_%03i%s.in" % (G_folder, rep, ab)               # cv:17
    base = s.output_folder[:-5] if (s.symm == 1 and s.kernel_type != "bosonic") else s.output_folder
    s.output_folder = base + "/%03i%s/free" % (rep, ab)             # cv:18
    if s.symm == 1 and s.kernel_type != "bosonic":
        s.output_folder += "_symm"                                  # cv:24-26
    return _free_cv_run(s, in_file, validate_tin)


def _free_cv_run(s, in_file=None, validate_tin=None):
    if s.is_root:
        os.makedirs(s.output_folder, exist_ok=True)
        if in_file and os.path.exists(in_file):
            shutil.copyfile(in_file, s.output_folder + "/in_free.in")   # cv:32-33
        if s.timedata is None:
            shutil.copyfile(s.G_file, s.output_folder + "/t.in")
    s.tol = 1e-6                                                    # cv:50
    gsac = s.output_folder + "/GSAC.csv"
    if s.is_root:
        open(gsac, "w").close()                                     # cv:54-58
    s.init_outfiles()
    s.write_log("Beginning Initialization.", "w")
    s.initialize()                                                  # cv:67-75
    s.write_log("Initialization Finished.")
    s.write_log("Beginning Equillibration.")
    s.fast_anneal()                                                 # cv:85
    s.write_log("Equillibration Finished.")
    s.write_log("Beginning Main Anneal.")

    score = _fused_validation(s, validate_tin)

    def dump(i, theta):
        g = s.mean_gbar()
        if s.is_root:
            _append_gsac(gsac, theta, s.td.cov, g)
        if score is not None:                                       # the same quantity without the file (row f3)
            score(theta, gbar_mean=g if s.comm is not None else None)

    s.main_anneal(on_stage=dump)                                    # cv:98-140
    s.write_log("Main Anneal Finished.")
    return s


def peak_cv(rep, ab, in_file="in_peak.in", A_0=None, N_p=None, validate_tin=None, **kw):
    """cv:147-312: as free_cv for the peak parameterisation (initial sampling, then the anneal with the
    GSAC dump)."""
    s = PeakSAC.from_infile(in_file, A_0_in=A_0, N_p_in=N_p, **kw)
    G_folder = s.G_file
    s.G_file = "%s/t_%03i%s.in" % (G_folder, rep, ab)
    tail = "/Np_%02i/A0_%.3f" % (s.N_p, s.A_0)                     # cv:158-177
    root = s.output_folder[: -len(tail)]
    symm_tag = "_symm" if (s.symm == 1 and s.kernel_type != "bosonic") else ""
    if symm_tag:
        root = root[: -len(symm_tag)]
    s.output_folder = root + "/%03i%s/peak" % (rep, ab) + symm_tag + tail
    if s.is_root:
        os.makedirs(s.output_folder, exist_ok=True)
    s.tol = 1e-6
    gsac = s.output_folder + "/GSAC.csv"
    if s.is_root:
        open(gsac, "w").close()
    s.init_outfiles()
    s.write_log("Beginning Initialization.", "w")
    s.initialize()
    s.write_log("Initialization Finished.")
    s.intial_sampling(s.theta_0)
    if s.chi2_min_all > 1000 * s.td.n_tau:
        s.indiv_update = True
        s.initialize()
        s.intial_sampling(s.theta_0)
    s.repair_chains()

    def dump(i, theta):
        g = s.h.gbar_mean()                                         # one call; over all ranks when the handle merges
        if s.comm is not None and not s.h.merged_over_ranks:        # host-side groups
            tot = s.comm.sum_vector(np.concatenate([g * s.n_chains, [float(s.n_chains)]]))
            g = tot[:-1] / tot[-1]
        if s.is_root:
            _append_gsac(gsac, theta, s.td.cov, g)
        if score is not None:
            score(theta, gbar_mean=g if s.comm is not None else None)

    score = _fused_validation(s, validate_tin)
    s.run_anneal(s.theta_0, on_stage=dump)
    return s


def edge_cv(rep, ab, in_file="in_edge.in", A_c=None, A_r=None, p=None, validate_tin=None, **kw):
    """cv:315-509: equilibration, then the full Theta ladder (no stop rule) with one GSAC row per stage."""
    import math
    s = EdgeSAC.from_infile(in_file, A_c_in=A_c, A_r_in=A_r, p_in=p, **kw)
    with open(in_file) as f:
        rows = [ln.split() for ln in f]
    G_folder, out = rows[5][0], rows[5][1]
    s.G_file = "%s/t_%03i%s.in" % (G_folder, rep, ab)               # cv:329-330
    out += "/%03i%s/edge" % (rep, ab) + "/p_%.3f" % s.p             # cv:330, 362
    if s.mode in ("double_edge_in", "double_edge_out"):
        out += "/Ar_%.3f" % s.A_r
    s.output_folder = out
    if s.is_root:
        os.makedirs(out, exist_ok=True)
    gsac = out + "/GSAC.csv"
    if s.is_root:
        open(gsac, "w").close()
    s.write_log("Beginning Initialization.", "w")
    s.init_outfiles()
    s.initialize()
    score = _fused_validation(s, validate_tin)
    s.write_log("Running Equillibration Sweeps.")
    s.run_bins(5 * s.anneal_steps, s.bins, s.theta_0)               # cv:466-467
    s.write_res(0, s.theta_0)
    theta, i_trans = s.theta_0, s.N_anneal * 0.2
    for i in range(1, s.N_anneal + 1):                              # cv:470-500
        if i < i_trans:
            steps = int(math.ceil(s.anneal_steps * (1 - ((5 / 6) * i / i_trans))))
        else:
            steps = int(math.ceil(s.anneal_steps / 6))
        s.run_bins(steps, s.bins, theta)
        s.write_res(i, theta)
        g = s.h.gbar_mean()                                         # one call; over all ranks when the handle merges
        if s.comm is not None and not s.h.merged_over_ranks:        # host-side groups
            tot = s.comm.sum_vector(np.concatenate([g * s.n_chains, [float(s.n_chains)]]))
            g = tot[:-1] / tot[-1]
        if s.is_root:
            _append_gsac(gsac, theta, s.td.cov, g)
        if score is not None:
            score(theta, gbar_mean=g if s.comm is not None else None)
        theta /= s.f_anneal
        s.chi2_anneal[i - 1] = s.res["chi2_res"][1]
    s.write_log("Main Anneal Finished.")
    return s


def calc_cv(tin_path, gsac_path):
    """calc_cv.jl:29-58: chi2 / N_tau of every GSAC row (a half's fitted G(tau) per Theta) against the OTHER half's
    t.in, in that half's eigenbasis.  Returns (thetas, chi2_val / N_tau)."""
    from .tin import read_tin
    td = read_tin(tin_path)
    rows = np.loadtxt(gsac_path, delimiter=",", ndmin=2)
    thetas, gsac = rows[:, 0], rows[:, 1:]
    d = (gsac @ td.cov - td.G) * td.sigma_inv                      # cvcalc:37, 45-51
    return thetas, (d * d).sum(axis=1) / td.n_tau


def cv_score(folder_a, folder_b, tin_a, tin_b):
    """calc_cv.jl:82-107: validation curve of one replica = mean of (GSAC_b vs t_a) and (GSAC_a vs t_b)."""
    th, v_ab = calc_cv(tin_a, folder_b + "/GSAC.csv")
    _, v_ba = calc_cv(tin_b, folder_a + "/GSAC.csv")
    n = min(len(v_ab), len(v_ba))
    return th[:n], 0.5 * (v_ab[:n] + v_ba[:n])


# ---- cross-validation units over the GPUs of a node (run_cv.sh fans the same units out as
# separate Julia processes) --------------------------------------------------------------------------------

def cv_units(reps, params=("free", "peak", "edge")):
    """The independent units of a K-fold run: (parameterisation, replica, half)."""
    return [(prm, int(rep), ab) for rep in reps for ab in ("a", "b") for prm in params]


def partition_units(units, rank, world):
    """Round-robin: unit i runs on rank i % world (whole GPUs per unit, no communication while it runs)."""
    return list(units[rank::world])


# Units are whole runs (an edge run takes minutes) dealt round-robin, so the ranks reach the final gather far apart: it
# must not sit under a collective watchdog sized for per-stage all-reduces (NCCL's default is 10 minutes).
import datetime
GATHER_TIMEOUT = datetime.timedelta(hours=48)
_gather_group = None


def gather_objects(obj, world):
    """Every rank's `obj` on every rank, through a host-side (gloo) group of its own with GATHER_TIMEOUT -- whatever
    backend and timeout the default group has.  Collective; the group is created on first use."""
    global _gather_group
    import torch.distributed as dist
    if _gather_group is None:
        _gather_group = dist.new_group(backend="gloo", timeout=GATHER_TIMEOUT)
    out = [None] * world
    dist.all_gather_object(out, obj, group=_gather_group)
    return out


def run_partition(units, rank, world, runner, gather=None, side_by_side=1, **kw):
    """Deals `units` round-robin to the ranks, runs this rank's share with runner(unit, **kw) and returns, on every
    rank, {unit: result} for all units.  gather(obj) -> list of every rank's obj (default: gather_objects, a gloo
    all_gather_object with a long timeout, when world > 1).  Shared by the cross-validation units and the scans (scan.py).
    side_by_side > 1: that many of the rank's units at a time, one host thread each.  A unit is a complete run with few
    chains -- a latency-bound sequence of small launches that leaves the GPU empty -- and handles are independent of each
    other (own stream, copies ordered on it), so a rank's units overlap on its GPU (three to four per
    process is what the hardware queues of one process carry).  The results do not depend on it."""
    share = partition_units(units, rank, world)
    if side_by_side > 1 and len(share) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(int(side_by_side)) as ex:
            futures = [(unit, ex.submit(runner, unit, **kw)) for unit in share]
            mine = {unit: f.result() for unit, f in futures}
    else:
        mine = {unit: runner(unit, **kw) for unit in share}
    if world == 1:
        return mine
    if gather is None:
        def gather(obj):
            return gather_objects(obj, world)
    merged = {}
    for part in gather(mine):
        merged.update(part)
    missing = [u for u in units if u not in merged]
    if missing:
        raise RuntimeError(f"units without a result: {missing}")
    return merged


def run_cv_partition(units, rank=0, world=1, runner=None, gather=None, side_by_side=1, **kw):
    """Runs this rank's share of `units` and returns, on every rank, {unit: output folder} for all units.
    runner(unit, **kw) -> output folder; default = free_cv / peak_cv / edge_cv on this rank's GPU."""
    if runner is None:
        def runner(unit, **k):
            prm, rep, ab = unit
            return {"free": free_cv, "peak": peak_cv, "edge": edge_cv}[prm](rep, ab, **k).output_folder
    return run_partition(units, rank, world, runner, gather, side_by_side=side_by_side, **kw)


def cv_curves(folders, tin_folder, params=("free", "peak", "edge")):
    """Validation curves per (parameterisation, replica) from the folders run_cv_partition returns
    (calc_cv.jl:82-107): {(prm, rep): (thetas, chi2_val / N_tau)}."""
    out = {}
    for prm in params:
        for rep in sorted({u[1] for u in folders if u[0] == prm}):
            fa, fb = folders.get((prm, rep, "a")), folders.get((prm, rep, "b"))
            if fa is None or fb is None:
                continue
            out[(prm, rep)] = cv_score(fa, fb, "%s/t_%03ia.in" % (tin_folder, rep), "%s/t_%03ib.in" % (tin_folder, rep))
    return out
