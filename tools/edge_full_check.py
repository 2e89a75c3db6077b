#!/usr/bin/env python3
"""The reference's complete first anneal of edge/out_files/hchain_beta2048_single (same Theta ladder, same sweeps per stage)
on many chains, with the reference's single-chain value placed in the distribution of the GPU chains' stage means:
z = (ref - m) / (s sqrt(1 + 1/n)), m, s = mean and scatter over the chains.  Usage: edge_full_check.py [chains] [stages] [d_omega]"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from sac_amd.edge import EdgeSAC  # noqa: E402

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 64
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dw = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
REF = os.path.join(ROOT, "tests", "golden", "edge", "hchain_single")
s = EdgeSAC.from_infile(f"{REF}/in_edge.in", n_chains=chains, seed=3)
s.G_file, s.output_folder, s.d_omega = f"{REF}/t.in", "/tmp/edge_full", dw
os.makedirs(s.output_folder, exist_ok=True)
s.write_log("start", "w")
s.init_outfiles()
s.initialize()
ref = np.genfromtxt(f"{REF}/anneal.csv", delimiter=",", skip_header=1)
nt = s.td.n_tau
t0 = time.time()
s.run_bins(5 * s.anneal_steps, s.bins, s.theta_0)
theta, i_trans = s.theta_0, s.N_anneal * 0.2
zs = []
for i in range(1, stages + 1):
    steps = int(math.ceil(s.anneal_steps * (1 - (5 / 6) * i / i_trans))) if i < i_trans else int(math.ceil(s.anneal_steps / 6))
    s.run_bins(steps, s.bins, theta)
    per = [s.h.results(c) for c in range(chains)]
    rr = ref[i]
    line = []
    for name, col, vals in (("chi2", 3, [p["chi2_res"][1] / nt for p in per]), ("edge", 5, [p["edge_res"][1, 0] for p in per]),
                            ("n0", 7, [p["n0_res"][1, 0] for p in per]), ("eps", 8, [p["eps0"][0] for p in per])):
        v = np.array(vals)
        z = (rr[col] - v.mean()) / (v.std(ddof=1) * math.sqrt(1 + 1 / chains))
        line.append(f"{name} {v.mean():.4f}+-{v.std(ddof=1):.4f} [{v.min():.4f},{v.max():.4f}] ref {rr[col]:.4f} z {z:5.2f}")
        if name == "chi2":
            zs.append(z)
    print(f"{i:3d} theta {theta:.5g} " + " | ".join(line) + f" min {s.chi2_min / nt:.4f} ref {rr[2]:.4f} t={time.time() - t0:.0f}s", flush=True)
    theta /= s.f_anneal
zs = np.array(zs)
print("chi2: max |z|", np.abs(zs).max(), "rms", np.sqrt((zs ** 2).mean()), "mean", zs.mean())
