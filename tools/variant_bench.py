#!/usr/bin/env python3
"""Headline-shape sample stage timed on alternative builds of the library (SAC_LIB=path): development aid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.free import FreeProblem
from sac_amd.synth import synthetic_timedata
td = synthetic_timedata(n_tau=64, seed=12345)
prob = FreeProblem(td, 2000, 0.0, 10.0, 1e-4, par=1, kernel_type="bosonic")
cpw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h = prob.make_handle(n_chains=4096, seed=12345, device_tables=True)
h.set_chains_per_wave(cpw)
h.adjust_windows(500, 1.0)
h.sample(100, 1.0)
s0 = h.stats(); kms = 0.0
for _ in range(3):
    h.sample(200, 1.0); s = h.stats(); kms += s.last_kernel_ms
print(json.dumps({"lib": os.environ.get("SAC_HIP_LIB", "default"), "cpw": h.chains_per_wave(), "chain_sweeps_per_s": (s.sweeps - s0.sweeps) / (kms / 1e3),
                  "accepted_moves_per_s": (s.accepted_moves - s0.accepted_moves) / (kms / 1e3), "wi_sum": int(h.get_state(7)[0].sum()), "chi2": h.get_gbar(7)[1]}))
