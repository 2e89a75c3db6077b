#!/usr/bin/env python3
"""Cluster sizes (n_multi, edge:1408-1432) of the chains at the edge bench shape (tools/bench_param.py edge: N_tau = 64, N_e = 80,
Theta = 1) after the bench's stages and after 2000 more sweeps: how many cluster moves go through K_new (more than four
deltas) instead of committing from registers.  Measured on the final build of round 3: 1 % after the bench's stages (mean size
1.7), 25 % after 2000 more sweeps (mean 3.2, 3 % above eight)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sac_amd.edge import EdgeProblem
from sac_amd.synth import synthetic_timedata
td = synthetic_timedata(n_tau=64, seed=777)
prob = EdgeProblem(td, "single_edge", "bosonic", 80, 0, 0.5, 0.0, 1.0, 0.0, 10.0, 1e-4)
h = prob.make_handle(n_chains=256, seed=777)
h.run_bins(20, 2, 1.0)
h.run_bins(40, 4, 1.0)
nm = np.concatenate([h.get_state(c)["nm"][2:81, 0] for c in range(0, 256, 4)])
vals, cnt = np.unique(nm, return_counts=True)
print("cluster sizes n_multi (delta 2..80) after the bench's stages:", dict(zip(vals.tolist(), (cnt / cnt.sum()).round(3).tolist())))
print("share of cluster proposals with n > 4:", float((nm > 4).mean()), " n > 8:", float((nm > 8).mean()), " mean n:", float(nm.mean()))
h.run_bins(200, 10, 1.0)
nm = np.concatenate([h.get_state(c)["nm"][2:81, 0] for c in range(0, 256, 4)])
print("after 2000 more sweeps: share n > 4:", float((nm > 4).mean()), " n > 8:", float((nm > 8).mean()), " mean n:", float(nm.mean()))
