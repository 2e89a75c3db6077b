"""Exploratory: walk the committed reference edge run's Theta ladder (tests/golden/edge/<run>) with the reference's
own sweeps per stage and print <chi2>/N_tau, edge and n0 against its anneal.csv.
Usage: python tools/edge_equilibrium_check.py [run] [stride] [anneal_steps] [n_chains] [n_stages] [d_omega]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sac_amd.edge import EdgeSAC  # noqa: E402

run = sys.argv[1] if len(sys.argv) > 1 else "hchain_single"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
chains = int(sys.argv[4]) if len(sys.argv) > 4 else 16
stages = int(sys.argv[5]) if len(sys.argv) > 5 else 24
dw = float(sys.argv[6]) if len(sys.argv) > 6 else 1e-4
REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "edge", run)
s = EdgeSAC.from_infile(f"{REF}/in_edge.in", n_chains=chains, seed=7)
s.G_file = f"{REF}/t.in"
s.output_folder = "/tmp/edge_eq_out"
s.d_omega = dw
f_ref = s.f_anneal
s.f_anneal = f_ref ** stride
s.N_anneal, s.anneal_steps = stages, steps
s.chi2_anneal = np.zeros(stages)
os.makedirs(s.output_folder, exist_ok=True)
s.write_log("start", "w")
s.init_outfiles()
t0 = time.time()
s.initialize()
print("init", time.time() - t0, "s", (s.prob.j_opt, s.prob.k_opt), flush=True)
ref = np.genfromtxt(f"{REF}/anneal.csv", delimiter=",", skip_header=1)
nt = s.td.n_tau
s.run_bins(5 * steps, s.bins, s.theta_0)
r = s.res
print(f"equil chi2avg {r['chi2_res'][1]/nt:.3f} ref {ref[0,3]:.3f} min {s.chi2_min/nt:.4f} ref {ref[0,2]:.4f} "
      f"edge {r['edge_res'][1,0]:.4f} ref {ref[0,5]:.4f} ms {r['last_stage_ms']:.0f} sweeps/s/chain "
      f"{5*steps*s.bins/(r['last_stage_ms']/1e3):.0f}", flush=True)
theta = s.theta_0
i_trans = (stages * stride) * 0.2
for i in range(1, stages + 1):
    iref = 1 + (i - 1) * stride
    st = int(np.ceil(steps * (1 - (5 / 6) * iref / i_trans))) if iref < i_trans else int(np.ceil(steps / 6))
    s.run_bins(st, s.bins, theta)
    r = s.res
    rr = ref[iref]
    print(f"{i:3d} theta {theta:8.4f} ref {rr[1]:8.4f} chi2avg {r['chi2_res'][1]/nt:8.4f} ref {rr[3]:8.4f} ratio "
          f"{r['chi2_res'][1]/nt/rr[3]:6.3f} min {s.chi2_min/nt:7.4f} ref {rr[2]:7.4f} edge {r['edge_res'][1,0]:.4f} ref {rr[5]:.4f} "
          f"n0 {r['n0_res'][1,0]:.3f} ref {rr[7]:.3f} eps {r['eps0'][0]:.3f} t={time.time()-t0:.0f}s", flush=True)
    theta /= s.f_anneal
