#!/usr/bin/env python3
"""Latency of one chain (sweeps/s per chain) with 1, 2 and 4 proposals of the chain per step (free_sweep_spec.h), for runs
with few chains: the reference's own input shape (N_tau = 22, 1000 deltas), an N_tau = 32 one, and BASELINE configs[1]'s
(N_tau = 64, 2000 deltas), 100001-column table.  One JSON line per (shape, chains, proposals per step)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.free import FreeProblem
from sac_amd.synth import synthetic_timedata

for n_tau, n_omega in ((22, 1000), (32, 1000), (64, 2000)):
    td = synthetic_timedata(n_tau=n_tau, seed=12345)
    prob = FreeProblem(td, n_omega, 0.0, 10.0, 1e-4, par=1, kernel_type="bosonic")
    for chains in (1, 64, 1024):
        base = None
        for pps in (1, 2, 4):
            h = prob.make_handle(n_chains=chains, seed=12345, device_tables=True)
            h.set_chains_per_wave(1)
            h.set_proposals_per_step(pps)
            h.adjust_windows(300, 1.0)
            h.sample(100, 1.0)
            kms = 0.0
            for _ in range(2):
                h.sample(200, 1.0)
                kms += h.stats().last_kernel_ms
            rate = 400 / (kms / 1e3)
            base = base or rate
            print(json.dumps({"n_tau": n_tau, "n_omega": n_omega, "chains": chains, "proposals_per_step": h.proposals_per_step(),
                              "sweeps_per_s_per_chain": round(rate, 1), "vs_one": round(rate / base, 3),
                              "chain_sweeps_per_s": round(rate * chains, 1)}), flush=True)
            h.close()
