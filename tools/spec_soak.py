#!/usr/bin/env python3
"""Soak test of the few-chains kernel (free_sweep_spec.h): 2 and 4 proposals per step against one proposal per step over long runs
on shapes that hit the rare paths hard (few deltas: proposals of one step share deltas; low and high acceptance)."""
import itertools, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.free import FreeProblem
from sac_amd.synth import synthetic_timedata

bad = 0
for n_tau, n_omega, theta in itertools.product((13, 22, 40, 64), (7, 20, 100, 1000), (0.05, 1.0, 20.0)):
    td = synthetic_timedata(n_tau=n_tau, seed=99)
    prob = FreeProblem(td, n_omega, 0.0, 8.0, 1e-3, par=1, kernel_type="bosonic")
    sweeps = 3000 if n_omega <= 100 else 400
    ref = None
    for pps in (1, 2, 4):
        h = prob.make_handle(n_chains=16, seed=2025, hist_copies=16)
        h.set_chains_per_wave(1)
        h.set_proposals_per_step(pps)
        h.adjust_windows(sweeps, theta)
        h.sample(sweeps // 2, theta)
        st = h.stats()
        out = ([h.get_state(c)[0] for c in range(16)], [h.get_gbar(c) for c in range(16)], h.histogram(),
               (st.proposals, st.evaluated, st.accepted, st.accepted_moves, st.columns_gathered, st.chi2_min, st.sampled_chi2))
        h.close()
        if ref is None:
            ref = out
            continue
        ok = all(np.array_equal(a, b) for a, b in zip(out[0], ref[0])) and np.array_equal(out[2], ref[2]) and out[3] == ref[3] \
            and all(np.array_equal(g, g0) and c == c0 and m == m0 for (g, c, m), (g0, c0, m0) in zip(out[1], ref[1]))
        if not ok:
            bad += 1
        print(json.dumps({"n_tau": n_tau, "n_omega": n_omega, "theta": theta, "pps": pps, "sweeps": sweeps + sweeps // 2, "equal": bool(ok),
                          "accepted": int(out[3][2])}), flush=True)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
