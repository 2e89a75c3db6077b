#!/usr/bin/env python3
"""Chain-sweeps/s of the free sample stage for 1 / 2 / 4 chains per wavefront (free_sweep_packed.h) at the headline
shape and at the small-N_tau shapes the reference's own inputs have.  One JSON line per (shape, chains per wave)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.free import FreeProblem          # noqa: E402
from sac_amd.synth import synthetic_timedata  # noqa: E402

SHAPES = [
    # name, n_tau, n_omega, omega_m, d_omega, chains
    ("ntau64_n2000_grid1e5 (BASELINE configs[1])", 64, 2000, 10.0, 1e-4, 4096),
    ("ntau22_n1000_grid1e5 (bosonic_delta_peak shape)", 22, 1000, 10.0, 1e-4, 4096),
    ("ntau32_n1000_grid1e5 (hchain_beta2048 shape)", 32, 1000, 10.0, 1e-4, 4096),
    ("ntau64_n2000_grid1e6", 64, 2000, 10.0, 1e-5, 4096),
]
only = sys.argv[1:] and [int(a) for a in sys.argv[1:]]
for si, (name, n_tau, n_omega, omega_m, d_omega, chains) in enumerate(SHAPES):
    if only and si not in only:
        continue
    td = synthetic_timedata(n_tau=n_tau, seed=12345)
    prob = FreeProblem(td, n_omega, 0.0, omega_m, d_omega, par=1, kernel_type="bosonic")
    for cpw in (1, 2, 4):
        for nch in ((chains,) if cpw == 1 else (chains, chains * 2)):
            h = prob.make_handle(n_chains=nch, seed=12345, device_tables=True)
            h.set_chains_per_wave(cpw)
            h.adjust_windows(500, 1.0)
            h.sample(100, 1.0)
            s0 = h.stats()
            kms = 0.0
            for _ in range(2):
                h.sample(200, 1.0)
                s = h.stats()
                kms += s.last_kernel_ms
            print(json.dumps({"shape": name, "chains": nch, "chains_per_wave": h.chains_per_wave(),
                              "chain_sweeps_per_s": (s.sweeps - s0.sweeps) / (kms / 1e3),
                              "accepted_moves_per_s": (s.accepted_moves - s0.accepted_moves) / (kms / 1e3),
                              "ms_per_100_sweeps": kms / 4}), flush=True)
            h.close()
