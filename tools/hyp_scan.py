#!/usr/bin/env python3
"""Latency of one chain (sweeps/s per chain) for the three ways a run with few chains can be stepped: one proposal per step
(free_sweep.h), two proposals per step in lane groups (free_sweep_spec.h), hypothesis pairs (free_sweep_hyp.h).  Shapes: the
reference's own input shape (N_tau = 22, 1000 deltas), an N_tau = 32 one, N_tau = 48 and BASELINE configs[1]'s (N_tau = 64,
2000 deltas), 100001-column table.  One JSON line per (shape, chains, mode)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.free import FreeProblem
from sac_amd.synth import synthetic_timedata

shapes = ((22, 1000), (32, 1000), (48, 2000), (64, 2000))
chain_counts = tuple(int(a) for a in sys.argv[1:]) or (1, 64, 1024)
for n_tau, n_omega in shapes:
    td = synthetic_timedata(n_tau=n_tau, seed=12345)
    prob = FreeProblem(td, n_omega, 0.0, 10.0, 1e-4, par=1, kernel_type="bosonic")
    for chains in chain_counts:
        base = None
        for mode in ("one", "groups2", "pairs"):
            h = prob.make_handle(n_chains=chains, seed=12345, device_tables=True)
            h.set_chains_per_wave(1)
            h.set_hypothesis_pairs(1 if mode == "pairs" else -1)
            h.set_proposals_per_step(2 if mode == "groups2" else 1)
            h.adjust_windows(300, 1.0)
            h.sample(100, 1.0)
            kms = 0.0
            for _ in range(2):
                h.sample(200, 1.0)
                kms += h.stats().last_kernel_ms
            rate = 400 / (kms / 1e3)
            base = base or rate
            print(json.dumps({"n_tau": n_tau, "n_omega": n_omega, "chains": chains, "mode": mode,
                              "hypothesis_pairs": h.hypothesis_pairs(), "proposals_per_step": h.proposals_per_step(),
                              "sweeps_per_s_per_chain": round(rate, 1), "vs_one": round(rate / base, 3),
                              "chain_sweeps_per_s": round(rate * chains, 1)}), flush=True)
            h.close()
