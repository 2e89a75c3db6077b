"""Theta_opt must not depend on how many chains the GPU run uses.  The reference forms its criteria from ONE chain:
chi2_min is that chain's own minimum (free:1301-1310, peak:1480-1500) and, in the edge program, the second anneal stops the
first time that chain's stage average drops below the target (edge:1680).  A minimum over many chains keeps falling as chains are
added, and so would Theta_opt.  The drivers' default (chi2_min_mode="chain_mean") uses the single-chain equivalents: the mean over
the chains of each chain's own chi2_min, and for the edge rule the median of the chains' own first passages.  Each test runs the
reference's ladder and stop rules on few and on many chains (different seeds, fewer sweeps per stage than the reference to keep
the suite short) and requires the same ladder point or a close neighbour, and the reference run's own Theta_opt nearby."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from sac_amd.edge import EdgeSAC
from sac_amd.free import FreeSAC
from sac_amd.peak import PeakSAC

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _together(*jobs):
    """Runs side by side, one host thread each: they are independent handles on their own streams, every one leaves most of the
    GPU idle (at most 64 wavefronts) and its wall-clock is the latency of its chains, so together they take about the time of
    the longest (handles are per driver task; the library's error state is thread-local and its copies are ordered on the
    handle's own stream, so one run never waits for another's kernels)."""
    with ThreadPoolExecutor(len(jobs)) as ex:
        futures = [ex.submit(*job) for job in jobs]
        return [f.result() for f in futures]


def _free_theta(tmp_path, n_chains, seed, mode):
    s = FreeSAC.from_infile(f"{G}/free_bdp/in_free.in", n_chains=n_chains, seed=seed, chi2_min_mode=mode,
                            device_tables=True)          # 1.5 million columns: 14 s on the host, 1 ms on the device (row f1)
    s.G_file = f"{G}/free_bdp/t.in"
    s.output_folder = str(tmp_path / f"free_{n_chains}_{mode}")
    s.anneal_steps //= 2                                   # 250 + 250 sweeps per Theta (reference: 500 + 500)
    s.init_outfiles()
    s.initialize()
    s.h.set_state(windows=np.full(4, s.prob.window))
    s.fast_anneal()
    s.main_anneal(False)
    return s.optimal_theta(), s.chi2_min / s.td.n_tau, s.chi2_min_all / s.td.n_tau


def test_free_theta_opt_is_independent_of_the_chain_count(tmp_path):
    ref_theta = np.loadtxt(f"{G}/free_bdp/sample.csv", delimiter=",", skiprows=1, ndmin=2)[0][2]
    step = np.log(1.1)                                     # f_anneal of the run
    (few, c_few, _), (many, c_many, c_all), (low, _, _) = _together(
        (_free_theta, tmp_path, 4, 21, "chain_mean"), (_free_theta, tmp_path, 64, 7, "chain_mean"), (_free_theta, tmp_path, 64, 7, "min"))
    print("free theta_opt: 4 chains", few, "64 chains", many, "64 chains, minimum over chains", low, "reference", ref_theta,
          "chi2_min/N", c_few, c_many, c_all)
    assert abs(np.log(few / many)) <= 2.01 * step
    assert abs(np.log(many / ref_theta)) <= 2.01 * step and abs(np.log(few / ref_theta)) <= 3.01 * step
    assert abs(c_few / c_many - 1) < 0.03                  # the single-chain-equivalent chi2_min does not move with the count
    assert c_all <= c_many and low <= many                 # the minimum over chains does: lower chi2_min, lower (or equal) Theta_opt


def _peak_theta(tmp_path, n_chains, seed):
    s = PeakSAC.from_infile(f"{G}/peak_cv001a/in_peak.in", n_chains=n_chains, seed=seed, device_tables=True)
    s.G_file = f"{G}/cv/t_001a.in"
    s.output_folder = str(tmp_path / f"peak_{n_chains}")
    s.anneal_steps //= 2
    s.tol = 1e-6                                           # cross_val.jl:200
    s.init_outfiles()
    s.initialize()
    s.intial_sampling(s.theta_0)
    s.repair_chains()
    s.run_anneal(s.theta_0)
    return s.optimal_theta()[2], s.chi2_min / s.td.n_tau, s.a_criterion, s.td.n_tau


def test_peak_theta_opt_is_independent_of_the_chain_count(tmp_path):
    step = np.log(1.1)
    (few, c_few, _, _), (many, c_many, a_crit, n_tau) = _together((_peak_theta, tmp_path, 4, 5), (_peak_theta, tmp_path, 64, 11))
    # the reference run (cross-validation leg: no sample.csv) fixes Theta_opt through its anneal.csv: peak:1545-1548 applied to
    # the columns it printed (chi2_min / N_tau of the last row, <chi2> / N_tau per Theta, both rounded to 4 digits)
    ref = np.loadtxt(f"{G}/peak_cv001a/anneal.csv", delimiter=",", skiprows=1)
    cmin = ref[-1, 2] * n_tau
    a_vals = (ref[:, 3] * n_tau - cmin) / np.sqrt(2 * cmin)
    ref_theta = ref[np.argmin(np.abs(a_vals - a_crit)), 1]
    print("peak theta_opt: 4 chains", few, "64 chains", many, "reference", ref_theta, "chi2_min/N", c_few, c_many)
    assert abs(np.log(few / many)) <= 2.01 * step
    assert abs(np.log(many / ref_theta)) <= 2.01 * step and abs(np.log(few / ref_theta)) <= 3.01 * step
    assert abs(c_few / c_many - 1) < 0.03


def _edge_theta(tmp_path, n_chains, seed):
    s = EdgeSAC.from_infile(f"{G}/edge/hchain_single/in_edge.in", n_chains=n_chains, seed=seed)
    s.G_file = f"{G}/edge/hchain_single/t.in"
    s.output_folder = str(tmp_path / f"edge_{n_chains}")
    stride = 5                                             # every 5th Theta of the reference's ladder, a 10 x coarser kernel grid
    s.d_omega = 1e-4
    s.f_anneal = 1.05 ** stride
    s.N_anneal, s.anneal_steps, s.sample_steps = 30, 300, 100
    s.chi2_anneal = np.zeros(s.N_anneal)
    theta = s.run()
    fp = np.array(s.first_passages)
    s.h.close()
    return theta, fp, s.f_anneal


def test_edge_theta_opt_is_the_median_first_passage_at_any_chain_count(tmp_path):
    """edge:1680 stops at ONE chain's first passage, a broad distribution (profiles/r03_time_to_solution_edge.jsonl: the 10 % - 90 %
    range of the 64 chains of the full run spans a factor > 3 in Theta; the reference's own run sits at its 12th percentile).  The
    many-chain rule stops when half the chains have passed: its Theta_opt is the median of that distribution, whatever the count."""
    (few, fp_few, f), (many, fp_many, _), (one, fp_one, _) = _together(
        (_edge_theta, tmp_path, 16, 9), (_edge_theta, tmp_path, 64, 3), (_edge_theta, tmp_path, 1, 5))
    print("edge theta_opt: 16 chains", few, "64 chains", many, "passages", np.sort(fp_few[fp_few > 0]), np.sort(fp_many[fp_many > 0]))
    # The first-passage distribution is broad and flat around its middle (stages 6 ... 21 on this ladder), so the comparison is made
    # in quantiles, not in ladder stages: by the stage at which the 16-chain run stopped, between a quarter and three quarters of
    # the 64 chains must have passed (the median of 16 draws has a standard error of 0.125 in quantile), and vice versa
    k_few, k_many = fp_few.max(), fp_many.max()                      # the stage each run stopped at
    q_many_at_few = ((fp_many > 0) & (fp_many <= k_few)).mean()
    q_few_at_many = ((fp_few > 0) & (fp_few <= k_many)).mean() if k_many <= k_few else 1.0
    print("quantile of the 64-chain distribution at the 16-chain stop:", q_many_at_few, "and the reverse:", q_few_at_many)
    assert 0.25 <= q_many_at_few <= 0.75
    assert abs(np.log(few / many)) <= 4.01 * np.log(f)
    for fp in (fp_few, fp_many):
        crossed = (fp > 0).sum()
        assert 2 * crossed >= len(fp) and 2 * (crossed - (fp == fp.max()).sum()) < len(fp)     # stopped at the stage the median chain passed
    # a single chain follows the reference's rule to the letter: the stage of ITS first passage
    assert (fp_one > 0).all() and one == pytest.approx(10.0 / f ** (int(fp_one[0]) - 2), rel=1e-12)
