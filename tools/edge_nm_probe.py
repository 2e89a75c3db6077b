#!/usr/bin/env python3
"""Development aid: cluster sizes (nm) and per-delta accept rates of the edge bench workload after tuning."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sac_amd.synth import synthetic_timedata
from sac_amd.edge import EdgeProblem
td = synthetic_timedata(n_tau=64, seed=777)
prob = EdgeProblem(td, "single_edge", "bosonic", 80, 0, 0.5, 0.0, 1.0, 0.0, 10.0, 1e-5)
h = prob.make_handle(n_chains=4096, seed=777)
h.run_bins(20, 2, 1.0)
h.run_bins(40, 4, 1.0)
for c in (0, 1, 100):
    st = h.get_state(c)
    nm = np.asarray(st["nm"] if isinstance(st, dict) else st[3]).reshape(-1)
    print("chain", c, "nm[0:90] =", nm[:90].tolist())
r = h.results(-1)
print({k: r[k] for k in ("proposals", "evaluated", "accepted", "lookups") if k in r})
