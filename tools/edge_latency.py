#!/usr/bin/env python3
"""Latency of one edge chain (sweeps/s per chain) on the reference's hchain_beta2048_single input with few chains:
   python tools/edge_latency.py [chains] [d_omega]   (an -DEDGE_PROF build prints chain 0's section profile)"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from sac_amd.edge import EdgeSAC
REF = os.path.join(ROOT, "tests", "golden", "edge", "hchain_single")
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = EdgeSAC.from_infile(f"{REF}/in_edge.in", n_chains=chains, seed=3)
s.G_file, s.output_folder = f"{REF}/t.in", "/tmp/edge_latency"
s.d_omega = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
os.makedirs(s.output_folder, exist_ok=True)
s.write_log("start", "w")
s.init_outfiles()
s.initialize()
for theta in (10.0, 1.0):
    s.run_bins(600, 5, theta)                 # tune windows and cluster sizes
    s.run_bins(600, 5, theta)
    ms = s.h.results(-1)["last_stage_ms"]
    print(json.dumps({"chains": chains, "d_omega": s.d_omega, "theta": theta, "sweeps_per_s_per_chain": round(3000 / (ms / 1e3), 1),
                      "chi2_per_ntau": s.res["chi2_res"][0] / s.N_tau if hasattr(s, "N_tau") else None}), flush=True)
