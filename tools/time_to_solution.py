#!/usr/bin/env python3
"""Time to solution of complete reference runs on the GPU, next to the wall-clock the reference's own log.txt records
(unknown CPU, one Julia process, one chain), with the statistical agreement of the whole anneal.

  python tools/time_to_solution.py free [chains] [spread]   sac/free/out_files/bosonic_delta_peak   (tests/golden/free_bdp)
  python tools/time_to_solution.py edge [chains] [spread]   sac/edge/out_files/hchain_beta2048_single/...  (tests/golden/edge/hchain_single)

The reference's FULL schedule is run (same Theta ladder, same sweeps per stage, same stop rules).  One JSON line.
SAC_CHI2_MIN_MODE=chain_mean|min selects how the many-chain run forms the chi2_min of the Theta criteria (default chain_mean)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
what = sys.argv[1]
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spread = len(sys.argv) > 3 and sys.argv[3] == "spread"
mode = os.environ.get("SAC_CHI2_MIN_MODE", "chain_mean")
t_all = time.time()
phases = {}


def timed(name, fn):
    t0 = time.time()
    r = fn()
    phases[name] = round(time.time() - t0, 2)
    return r


if what == "free":
    from sac_amd.free import FreeSAC
    REF = f"{G}/free_bdp"
    s = FreeSAC.from_infile(f"{REF}/in_free.in", n_chains=chains, seed=7, spread_sampling=spread, device_tables=True, chi2_min_mode=mode)
    s.G_file, s.output_folder = f"{REF}/t.in", "/tmp/tts_free"
    per = []
    s.init_outfiles()
    timed("initialization", s.initialize)
    s.h.set_state(windows=np.full(4, s.prob.window))
    timed("equilibration", s.fast_anneal)
    timed("main_anneal", lambda: s.main_anneal(False, on_stage=lambda i, th: per.append(
        np.array([s.h.chain_stats(c)["sampled_chi2"] for c in range(min(chains, 256))]) / s.td.n_tau)))
    theta_opt = s.optimal_theta()
    s.h.set_state(windows=np.full(4, s.prob.window))
    timed("final_anneal", lambda: s.final_anneal(theta_opt))
    ref = np.loadtxt(f"{REF}/anneal.csv", delimiter=",", skiprows=1)
    got = np.loadtxt(s.anneal_file, delimiter=",", skiprows=1)
    n = min(len(ref), len(got))
    v = np.array(per)[:n]
    z = (v.mean(axis=1) - ref[:n, 3]) / (v.std(axis=1, ddof=1) * np.sqrt(1 + 1 / v.shape[1]))
    ref_row = np.loadtxt(f"{REF}/sample.csv", delimiter=",", skiprows=1, ndmin=2)[0]
    got_row = np.loadtxt(s.sample_file, delimiter=",", skiprows=1, ndmin=2)[0]
    out = {"run": "sac/free bosonic_delta_peak, full reference schedule", "chains": chains, "spread_sampling": spread, "chi2_min_mode": mode,
           "gpu_seconds": phases, "gpu_total_s": round(time.time() - t_all, 1),
           "reference_log_seconds": {"equilibration": 1.68, "main_anneal": 24.27, "final_anneal": 14.16, "note": "one chain, unknown CPU"},
           "anneal_points": [len(got), len(ref)], "z_max": float(np.abs(z).max()), "z_rms": float(np.sqrt((z ** 2).mean())),
           "theta_opt": [theta_opt, float(ref_row[2])], "chi2_avg_final": [float(got_row[4]), float(ref_row[4])],
           "sampling_sweeps_total": [int(chains * (max(20, -(-s.sample_steps // chains)) if spread else s.sample_steps)), s.sample_steps]}
elif what == "edge":
    from sac_amd.edge import EdgeSAC
    REF = f"{G}/edge/hchain_single"
    s = EdgeSAC.from_infile(f"{REF}/in_edge.in", n_chains=chains, seed=3, spread_sampling=spread, chi2_min_mode=mode)
    s.G_file, s.output_folder = f"{REF}/t.in", "/tmp/tts_edge"
    if len(sys.argv) > 4:
        s.d_omega = float(sys.argv[4])
    stamps = []
    orig = s.write_log

    def log(msg, mode="a"):
        stamps.append((time.time(), msg))
        orig(msg, mode)
    s.write_log = log
    orig_res = s.write_res

    def res(j, theta):                       # a progress line per stage (a silent run is taken to be hung)
        print(f"stage {j} theta {theta:.5g} chi2 {s.res['chi2_res'][1] / s.td.n_tau:.4f} t={time.time() - t_all:.0f}s", file=sys.stderr, flush=True)
        orig_res(j, theta)
    s.write_res = res
    theta_opt = s.run()
    marks = {m: t for t, m in stamps}
    def span(a, b):
        return round(marks[b] - marks[a], 2) if a in marks and b in marks else None
    phases = {"initialization": span("Beginning Initialization.", "Initialization Finished."),
              "equilibration": span("Running Equillibration Sweeps.", "Beginning Main Anneal."),
              "main_anneal": span("Beginning Main Anneal.", "Main Anneal Finished."),
              "final_anneal": span("Beginning Final Anneal.", "Final Anneal Finished."),
              "final_sampling": span("Beginning Final Sampling.", "Final Sampling Finished.")}
    ref = np.genfromtxt(f"{REF}/anneal.csv", delimiter=",", skip_header=1)
    got = np.genfromtxt(os.path.join(s.output_folder, "anneal.csv"), delimiter=",", skip_header=1)
    # first pass of the ladder (rows 1..N until the row index restarts for the second anneal)
    first = lambda a: a[1:1 + np.argmax(np.diff(a[1:, 0]) < 0)] if (np.diff(a[1:, 0]) < 0).any() else a[1:]   # noqa: E731
    r1, g1 = first(ref), first(got)
    n = min(len(r1), len(g1))
    # (the chain-by-chain comparison of <chi^2>(Theta) with the reference's single chain is tools/edge_full_check.py: the error bar
    # the reference prints is the scatter of its bins, far below the run-to-run scatter of the slow edge mode)
    rs = np.genfromtxt(f"{REF}/sample.csv", delimiter=",", skip_header=1)
    gs = np.genfromtxt(os.path.join(s.output_folder, "sample.csv"), delimiter=",", skip_header=1)
    out = {"run": "sac/edge hchain_beta2048_single, full reference schedule", "chains": chains, "spread_sampling": spread, "chi2_min_mode": mode,
           "d_omega": s.d_omega, "gpu_seconds": phases, "gpu_total_s": round(time.time() - t_all, 1),
           "reference_log_seconds": {"initialization": 3.7, "equilibration": 5.32, "main_anneal": 32.88, "final_anneal": 24.95,
                                     "final_sampling": 5.37, "note": "one chain, unknown CPU"},
           "anneal_points_first_pass": [len(g1), len(r1)],
           "chi2_ratio_last": float(g1[n - 1, 3] / r1[n - 1, 3]),
           "theta_opt": [float(theta_opt), float(rs[1])], "chi2_avg_final": [float(gs[3]), float(rs[3])], "edge_final": [float(gs[5]), float(rs[5])]}
    fp = getattr(s, "first_passages", None)
    if fp is not None:
        # theta of every chain's own first passage below its own target (0 = none before the anneal stopped): the reference's
        # theta_opt is ONE draw from this distribution
        th = np.array([s.theta_0 / s.f_anneal ** (k - 2) for k in fp if k > 0])      # anneal returns theta * f_anneal (edge:1681)
        out["first_passage_theta"] = {"chains_crossed": int((fp > 0).sum()), "min": float(th.min()), "p10": float(np.percentile(th, 10)),
                                      "median": float(np.median(th)), "p90": float(np.percentile(th, 90)), "max": float(th.max()),
                                      "reference_theta_opt_percentile": float((th <= float(rs[1])).mean())}
print(json.dumps(out))
