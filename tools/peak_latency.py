#!/usr/bin/env python3
"""Latency of one peak chain (sweeps/s per chain) on the reference's fermionic2 input (N_omega = 2000, N_p = 1, both axes,
two 1.5-million-column tables) with few chains:   python tools/peak_latency.py [chains]
The reference's log of this run (sac/peak/out_files/fermionic2/Np_01/A0_0.700/log.txt) is the context number."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from sac_amd.peak import PeakSAC
REF = os.path.join(ROOT, "tests", "golden", "peak_f2")
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = PeakSAC.from_infile(f"{REF}/in_peak.in", n_chains=chains, seed=11)
s.G_file, s.output_folder = f"{REF}/t.in", "/tmp/peak_latency"
s.init_outfiles()
s.initialize()
for theta in (15.0, 1.0):
    s.adjust_windows(300, theta)
    t0 = time.time()
    s.sample(300, theta)
    dt = time.time() - t0
    print(json.dumps({"chains": chains, "theta": theta, "sweeps_per_s_per_chain": round(300 / dt, 1), "n_tau": int(s.td.n_tau) if hasattr(s, "td") else None}), flush=True)
