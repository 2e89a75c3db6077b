#!/usr/bin/env python3
"""Exploratory calibration of the statistical parity tests: z-scores of <chi^2>(Theta) of the GPU chains against the
reference's committed anneal.csv files.  Usage: python tools/stat_scan.py free|peak|edge [args]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
what = sys.argv[1]
args = [float(a) for a in sys.argv[2:]]


def zrow(vals, ref, ratio, sig_ref=None):
    """vals: per-chain stage averages; ref: the reference's single-chain value; ratio = sweeps here / sweeps there."""
    m, s, n = vals.mean(), vals.std(ddof=1), len(vals)
    var = s * s / n + (s * s * ratio if sig_ref is None else sig_ref ** 2)
    return m, s, (m - ref) / np.sqrt(var)


if what == "free":
    from sac_amd.free import FreeSAC
    n_pts, chains = int(args[0]) if args else 40, int(args[1]) if len(args) > 1 else 64
    s = FreeSAC.from_infile(f"{G}/cv/run001a_free_in_free.in", n_chains=chains, seed=2025)
    s.G_file = f"{G}/cv/t_001a.in"
    s.output_folder = "/tmp/stat_free"
    s.N_anneal = n_pts
    s.chi2_anneal = np.zeros(n_pts)
    s.tol = 1e-6
    ref = np.loadtxt(f"{G}/cv/run001a_free_anneal.csv", delimiter=",", skiprows=1)
    nt = 22
    t0 = time.time()
    zs = []

    def hook(i, theta):
        v = np.array([s.h.chain_stats(c)["sampled_chi2"] for c in range(chains)]) / nt
        m, sd, z = zrow(v, ref[i - 1, 3], 1.0)
        zs.append(z)
        print(f"{i:3d} theta {theta:9.5f} gpu {m:9.4f} +- {sd:7.4f} ref {ref[i-1,3]:9.4f} z {z:6.2f} t={time.time()-t0:.0f}s", flush=True)
    s.init_outfiles(); s.initialize()
    s.h.set_state(windows=np.full(4, s.prob.window))
    s.fast_anneal()
    s.main_anneal(False, on_stage=hook)
    zs = np.array(zs)
    print("max |z|", np.abs(zs).max(), "mean z", zs.mean(), "mean z^2", (zs ** 2).mean())

elif what == "peak":
    from sac_amd.peak import PeakSAC
    stride, steps, chains, stages = (int(a) for a in (args + [18, 400, 32, 12][len(args):]))
    REF = f"{G}/peak_f2"
    s = PeakSAC.from_infile(f"{REF}/in_peak.in", n_chains=chains, seed=11)
    s.G_file = f"{REF}/t.in"
    s.output_folder = "/tmp/stat_peak"
    s.f_anneal = 1.03 ** stride
    s.N_anneal, s.anneal_steps = stages, steps
    s.chi2_anneal = np.zeros(stages)
    ref = np.loadtxt(f"{REF}/anneal.csv", delimiter=",", skiprows=1)[::stride]
    nt = None
    t0 = time.time()
    s.init_outfiles(); s.initialize()
    nt = s.td.n_tau
    s.intial_sampling(s.theta_0)
    print("repaired", s.repair_chains(), flush=True)
    zs = []

    def hook(i, theta):
        v = np.array([s.h.chain_stats(c)["sampled_chi2"] for c in range(chains)]) / nt
        m, sd, z = zrow(v, ref[i - 1, 3], steps / 5000.0)
        zs.append(z)
        print(f"{i:3d} theta {theta:9.5f} ref {ref[i-1,1]:9.5f} gpu {m:9.4f} +- {sd:7.4f} ref {ref[i-1,3]:9.4f} z {z:6.2f} t={time.time()-t0:.0f}s", flush=True)
    s.run_anneal(s.theta_0, on_stage=hook)
    zs = np.array(zs)
    print("max |z|", np.abs(zs).max(), "mean z", zs.mean(), "mean z^2", (zs ** 2).mean())

elif what == "peakcv":
    from sac_amd.peak import PeakSAC
    n_pts, chains = int(args[0]) if args else 50, int(args[1]) if len(args) > 1 else 64
    REF = f"{G}/peak_cv001a"
    s = PeakSAC.from_infile(f"{REF}/in_peak.in", n_chains=chains, seed=11)
    s.G_file = f"{G}/cv/t_001a.in"
    s.output_folder = "/tmp/stat_peakcv"
    s.N_anneal = n_pts
    s.chi2_anneal = np.zeros(n_pts)
    s.tol = 1e-6
    print("fix_edge", s.fix_edge, "steps", s.anneal_steps, s.kernel_type, s.N_omega, s.N_p)
    ref = np.loadtxt(f"{REF}/anneal.csv", delimiter=",", skiprows=1)
    t0 = time.time()
    s.init_outfiles(); s.initialize()
    nt = s.td.n_tau
    s.intial_sampling(s.theta_0)
    print("repaired", s.repair_chains(), "chi2_min", s.chi2_min / nt, flush=True)
    zs = []

    def hook(i, theta):
        v = np.array([s.h.chain_stats(c)["sampled_chi2"] for c in range(chains)]) / nt
        m, sd, z = zrow(v, ref[i - 1, 3], 1.0)
        zs.append(z)
        print(f"{i:3d} theta {theta:9.5f} ref {ref[i-1,1]:9.5f} gpu {m:9.4f} +- {sd:7.4f} ref {ref[i-1,3]:9.4f} z {z:6.2f} min {s.chi2_min/nt:.4f} ref {ref[i-1,2]:.4f} t={time.time()-t0:.0f}s", flush=True)
    s.run_anneal(s.theta_0, on_stage=hook)
    zs = np.array(zs)
    print("max |z|", np.abs(zs).max(), "mean z", zs.mean(), "mean z^2", (zs ** 2).mean())

elif what == "edge":
    from sac_amd.edge import EdgeSAC
    stride, steps, chains, stages = (int(a) for a in (args[:4] + [10, 300, 16, 12][len(args[:4]):]))
    dw = args[4] if len(args) > 4 else 1e-4
    REF = f"{G}/edge/hchain_single"
    s = EdgeSAC.from_infile(f"{REF}/in_edge.in", n_chains=chains, seed=3)
    s.G_file = f"{REF}/t.in"
    s.output_folder = "/tmp/stat_edge"
    s.d_omega = dw
    s.f_anneal = 1.05 ** stride
    s.N_anneal, s.anneal_steps = stages, steps
    s.chi2_anneal = np.zeros(stages)
    os.makedirs(s.output_folder, exist_ok=True)
    s.write_log("start", "w")
    s.init_outfiles()
    t0 = time.time()
    s.initialize()
    ref = np.genfromtxt(f"{REF}/anneal.csv", delimiter=",", skip_header=1)
    nt = s.td.n_tau
    s.run_bins(5 * steps, s.bins, s.theta_0)
    theta = s.theta_0
    i_trans = (stages * stride) * 0.2 if stride > 1 else 200 * 0.2
    for i in range(1, stages + 1):
        iref = 1 + (i - 1) * stride
        st = int(np.ceil(steps * (1 - (5 / 6) * iref / i_trans))) if iref < i_trans else int(np.ceil(steps / 6))
        if stride > 1:
            s.run_bins(st, s.bins, theta)      # equilibration at this Theta
        s.run_bins(st, s.bins, theta)
        per = [s.h.results(c) for c in range(chains)]
        rr = ref[iref]
        out = []
        for name, key, col, scol, div in (("chi2", "chi2_res", 3, 4, nt), ("edge", "edge_res", 5, 9, 1), ("n0", "n0_res", 7, 11, 1)):
            v = np.array([(p[key][1] if key == "chi2_res" else p[key][1, 0]) for p in per]) / div
            m, sd, z = zrow(v, rr[col], 0.0, sig_ref=rr[scol])
            out.append(f"{name} {m:8.4f}+-{sd:6.4f} ref {rr[col]:8.4f}+-{rr[scol]:6.4f} z {z:6.2f}")
        print(f"{i:3d} theta {theta:8.4f} ref {rr[1]:8.4f} steps {st} " + " | ".join(out) + f" t={time.time()-t0:.0f}s", flush=True)
        theta /= s.f_anneal
