"""Parameter scans and Theta scans over the GPUs of a node.

The reference fans a scan out as an SGE job array, one Julia process per parameter point:
  sac/qsub_files/edge/submit.sh:25-51 with qsub_p_scan.sh:13-19 (p), qsub_ac_scan.sh (A_c), qsub_ar_scan.sh (A_r):
      value = v_0 + delta * (task - 1);  julia sac_edge.jl A_c A_r p
  sac/qsub_files/peak/qsub_a0_scan.sh:12-18 (A_0):  julia sac_peak.jl A_0 N_p
and walks the Theta points of anneal_and_scan (edge:1783-1846) one after the other in one process.

Here a scan point is a whole-GPU unit: points are dealt round-robin to the ranks (one process per GPU), every rank runs its
points with all its chains, nothing is exchanged while a point runs, and the {point: result} map is gathered at the end
(cross_val.run_partition, the same routine that deals out cross-validation units).  Theta points are sharded the same way:
every rank repeats the (deterministic: same seed, same chain ids) anneal down to theta_1 and then walks the scan ladder,
sampling in full only at ITS points and passing the others with a short equilibration -- an algorithmic change against the
sequential reference, valid because each point's sample is taken in equilibrium at its own Theta.
"""
import os

import numpy as np

from .cross_val import run_partition

SCAN_AXES = {"edge": ("A_c", "A_r", "p"), "peak": ("A_0",)}


def scan_values(v_0, delta, n):
    """qsub_*_scan.sh: v_0 + delta * (task - 1), task = 1..n, as `bc` with scale=2 evaluates it: + and * keep the operands'
    own scale (p_0 = 0.5, delta = 0.025 gives .525, not 0.52), so the values are formed in exact decimal arithmetic from
    the argument STRINGS and carry max(scale of v_0, scale of delta) decimals.  Floats are taken by their shortest repr."""
    from decimal import Decimal
    a, d = Decimal(str(v_0)), Decimal(str(delta))
    return [float(a + d * i) for i in range(int(n))]


def scan_units(program, axis, values, **fixed):
    """One unit per scan value: (program, ((name, value), ...)) with the other command-line parameters fixed.
    edge: A_c, A_r, p (edge:2069-2071); peak: A_0, N_p (peak:1573-1575)."""
    if program not in SCAN_AXES or axis not in SCAN_AXES[program]:
        raise ValueError(f"no scan axis {axis!r} for {program!r}: {SCAN_AXES}")
    units = []
    for v in values:
        par = dict(fixed)
        par[axis] = float(v)
        units.append((program, tuple(sorted(par.items()))))
    return units


def _run_point(unit, in_file=None, **kw):
    program, par = unit
    par = dict(par)
    if program == "edge":
        from .edge import EdgeSAC
        s = EdgeSAC.from_infile(in_file or "in_edge.in", A_c_in=par.get("A_c"), A_r_in=par.get("A_r"), p_in=par.get("p"), **kw)
        theta_opt = s.run(in_file or "in_edge.in")
        r = s.res
        out = dict(folder=s.output_folder, theta_opt=theta_opt, chi2_min=s.chi2_min / s.td.n_tau,
                   chi2_avg=float(r["chi2_res"][1]) / s.td.n_tau, edge=float(r["edge_res"][1, 0]))
    else:
        from .peak import PeakSAC
        n_p = par.get("N_p")
        s = PeakSAC.from_infile(in_file or "in_peak.in", A_0_in=par.get("A_0"), N_p_in=None if n_p is None else int(n_p), **kw)
        theta_opt = s.run(in_file or "in_peak.in")
        out = dict(folder=s.output_folder, theta_opt=theta_opt, chi2_min=s.chi2_min / s.td.n_tau,
                   chi2_avg=s.sampled_chi2 / s.td.n_tau, edge=float(s.bounds[0] * s.d_omega))
    s.h.close()
    return out


def run_scan_partition(units, rank=0, world=1, runner=None, gather=None, side_by_side=1, **kw):
    """Runs this rank's scan points and returns, on every rank, {unit: result} for all points.
    runner(unit, **kw) -> result (default: the program's complete run at that parameter point on this rank's GPU).
    side_by_side: points a rank runs at a time on its GPU (cross_val.run_partition)."""
    return run_partition(units, rank, world, runner or _run_point, gather, side_by_side=side_by_side, **kw)


def best_point(results, key="chi2_avg"):
    """The scan's answer as the reference's users read it off the collected sample.csv files: the point with the
    smallest <chi^2> (README: 'the optimal parameter minimises chi^2')."""
    return min(results.items(), key=lambda kv: kv[1][key])


# ---- Theta points of anneal_and_scan over ranks ---------------------------------------------------------------------

def theta_scan_points(theta_1, theta_2, f_scan=1.15):
    """edge:1805-1809: N_scan = ceil(log(theta_1 / theta_2) / log(f_scan)) points theta_1 / f_scan^(i-1), i = 1..N_scan."""
    import math
    n = int(math.ceil(math.log(theta_1 / theta_2) / math.log(f_scan)))
    return [theta_1 / f_scan ** i for i in range(n)]


def sharded_theta_scan(s, theta_1, theta_2, rank=0, world=1, pass_fraction=0.1, gather=None):
    """anneal_and_scan (edge:1783-1846) with the Theta points dealt round-robin to the ranks.  `s`: an initialised
    EdgeSAC (initialize() + equilibration done) that is NOT chain-sharded (comm=None): every rank holds the same chains.
    Returns {i: row} for all points on every rank; row = the values of sample.csv's row i (edge:1757-1771)."""
    import math
    s.N_anneal = int(math.ceil(math.log(s.theta_0 / theta_1) / math.log(s.f_anneal)))
    s.chi2_anneal = np.zeros(s.N_anneal)
    s.saved = None
    root_was = s.is_root
    s.is_root = root_was and rank == 0                       # the shared anneal's files: once
    s.anneal(0.0, s.bins // 2, 0)
    f_scan = 1.15
    thetas = theta_scan_points(theta_1, theta_2, f_scan)
    n_scan = len(thetas)
    mine = {}
    for i, theta in enumerate(thetas, start=1):
        bins = int(math.ceil(s.bins * (1 + i / n_scan)))     # edge:1813
        if (i - 1) % world == rank:
            s.is_root = root_was                             # this rank owns the point: it writes its rows and spectra
            s.run_bins(s.sample_steps, bins, theta)
            s._sample_row(i, theta, with_cont=False)
            for rl in s.prob.rls:
                s.write_spec(i, bins * s.sample_steps, rl)
            r = s.res
            mine[i] = dict(theta=theta, chi2_min=s.chi2_min / s.td.n_tau, chi2_avg=float(r["chi2_res"][1]) / s.td.n_tau,
                           chi2_sigma=float(r["chi2_res"][2]) / s.td.n_tau, edge=float(r["edge_res"][1, 0]),
                           n0=float(r["n0_res"][1, 0]))
        else:                                                # someone else's point: pass through it
            s.is_root = False
            s.run_bins(max(1, int(s.sample_steps * pass_fraction)), max(2, bins // 4), theta)
    s.is_root = root_was
    if world == 1:
        return mine
    if gather is None:
        from .cross_val import gather_objects

        def gather(obj):
            return gather_objects(obj, world)     # gloo group with a long timeout: the ranks' points differ in length
    merged = {}
    for part in gather(mine):
        merged.update(part)
    return dict(sorted(merged.items()))
