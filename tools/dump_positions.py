#!/usr/bin/env python3
"""Writes the delta positions and tuned single-move windows of every chain of a tuned run at the bench shape (BASELINE
configs[1]) for `tools/gather_probe --positions`: int32 header (chains, n_omega, n_grid, 0), uint32 positions
[chains][n_omega] (relative grid index = table row), float64 window[chains].
    dump_positions.py [out] [chains] [d_omega]      d_omega 1e-4 (48.8 MiB table, default), 1e-5 (488 MiB), 2.5e-6 (1953 MiB)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.free import FreeProblem          # noqa: E402
from sac_amd.synth import synthetic_timedata  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/positions.bin"
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d_omega = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
td = synthetic_timedata(n_tau=64, seed=12345)
prob = FreeProblem(td, 2000, 0.0, 10.0, d_omega, par=1, kernel_type="bosonic")
h = prob.make_handle(n_chains=chains, seed=12345, device_tables=True)
h.adjust_windows(1000, 1.0)
pos = np.empty((chains, 2000), dtype=np.uint32)
win = np.empty(chains)
for c in range(chains):
    wi, _, w = h.get_state(c)
    pos[c] = wi - prob.wi_0
    win[c] = w[0]
with open(out, "wb") as f:
    np.array([chains, 2000, prob.n_grid, 0], dtype=np.int32).tofile(f)
    pos.tofile(f)
    win.tofile(f)
print("wrote", out, "mean window", win.mean(), "distinct columns per chain", np.mean([len(np.unique(p)) for p in pos[:64]]))
