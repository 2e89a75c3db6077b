"""Exploratory: walk the committed reference peak run's Theta ladder (tests/golden/peak_f2) with long
stages and print <chi2>/N_tau against the reference's anneal.csv.  Usage: python tools/peak_equilibrium_check.py
[stride] [anneal_steps] [n_chains] [n_stages]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sac_amd.peak import PeakSAC  # noqa: E402

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "peak_f2")
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chains = int(sys.argv[3]) if len(sys.argv) > 3 else 32
stages = int(sys.argv[4]) if len(sys.argv) > 4 else 48
s = PeakSAC.from_infile(f"{REF}/in_peak.in", n_chains=chains, seed=5)
s.G_file = f"{REF}/t.in"
s.output_folder = "/tmp/peak_eq_out"
s.f_anneal = 1.03 ** stride
s.N_anneal, s.anneal_steps = stages, steps
s.chi2_anneal = np.zeros(stages)
s.init_outfiles()
s.initialize()
t0 = time.time()
s.intial_sampling(s.theta_0)
print("initial sampling", time.time() - t0, "s; repaired", s.repair_chains(), flush=True)
ref = np.loadtxt(f"{REF}/anneal.csv", delimiter=",", skiprows=1)[::stride]
theta = s.theta_0
for i in range(stages):
    s.adjust_windows(steps, theta)
    s.sample(steps, theta)
    c2, c2min = s.h.chain_chi2()
    e = s._edges_and_weights()
    print(f"{i:3d} theta {theta:9.5f} ref {ref[i,1]:9.5f} chi2avg {s.sampled_chi2/s.td.n_tau:9.4f} ref {ref[i,3]:9.4f} "
          f"ratio {s.sampled_chi2/s.td.n_tau/ref[i,3]:6.3f} min {s.chi2_min/s.td.n_tau:7.4f} ref {ref[i,2]:7.4f} "
          f"edge {e[0]:.5f} {e[1]:.5f} A {e[2]:.4f} {e[3]:.4f} {e[4]:.4f} {e[5]:.4f}  t={time.time()-t0:.0f}s", flush=True)
    theta /= s.f_anneal
