#!/usr/bin/env python3
"""The workload of the few-chains instruction budget (rocprofv3 --pmc around it): 64 chains, reference-shaped input
(N_tau = 22, 1000 deltas) or BASELINE configs[1]'s shape, 1 or 2 proposals per step, one 200-sweep sample stage = 2 launches of
100 sweeps.   python tools/few_chains_pmc.py PPS [ntau] [nomega]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.free import FreeProblem
from sac_amd.synth import synthetic_timedata
pps = int(sys.argv[1])
n_tau = int(sys.argv[2]) if len(sys.argv) > 2 else 22
n_omega = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
td = synthetic_timedata(n_tau=n_tau, seed=12345)
prob = FreeProblem(td, n_omega, 0.0, 10.0, 1e-4, par=1, kernel_type="bosonic")
h = prob.make_handle(n_chains=64, seed=12345, device_tables=True)
h.set_chains_per_wave(1)
h.set_proposals_per_step(pps)
h.adjust_windows(300, 1.0)
s0 = h.stats()
h.sample(200, 1.0)
s1 = h.stats()
print(json.dumps({"pps": h.proposals_per_step(), "n_tau": n_tau, "n_omega": n_omega, "chains": 64, "sweeps": 200,
                  "proposals": s1.proposals - s0.proposals, "evaluated": s1.evaluated - s0.evaluated, "accepted": s1.accepted - s0.accepted,
                  "kernel_ms": s1.last_kernel_ms, "launches": s1.last_kernel_launches}))
