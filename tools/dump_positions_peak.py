#!/usr/bin/env python3
"""The peak counterpart of dump_positions.py: continuum delta positions and tuned single-move windows of every chain of a tuned
sac/peak run at tools/bench_param.py's shapes, in the format `tools/gather_probe --positions` reads.
    dump_positions_peak.py [out] [chains] [one|two] [d_omega]
one  = bosonic, one axis, one 10^6+1-column table (488 MiB; bench.py hbm_regime peak entry, BASELINE configs[3])
two  = finite-T, both axes: the two 488 MiB tables are laid out as ONE table of 2 (10^6+1) rows (negative-axis rows behind the
       positive-axis ones), 2000 continuum deltas per chain -- the footprint and the column distribution of the two-axis kernel"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sac_amd.peak import PeakProblem          # noqa: E402
from sac_amd.synth import synthetic_timedata  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/positions_peak.bin"
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
two = len(sys.argv) > 3 and sys.argv[3] == "two"
d_omega = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-5
SEED = 777
if two:
    td = synthetic_timedata(n_tau=64, seed=SEED, kernel_type="finiteT")
    prob = PeakProblem(td, 1000, 1, 0.5, 10.0, d_omega, kernel_type="finiteT", approx_ratio=2.5)
else:
    td = synthetic_timedata(n_tau=64, seed=SEED)
    prob = PeakProblem(td, 1000, 1, 0.5, 10.0, d_omega, kernel_type="bosonic")
h = prob.make_handle(n_chains=chains, seed=SEED, device_tables=True)
h.adjust_windows(100, 1.0)
h.sample(20, 1.0)
n_p, n = 1, 1000
per = 2 * n if two else n
pos = np.empty((chains, per), dtype=np.uint32)
win = np.empty(chains)
for c in range(chains):
    wi, _, w = h.get_state(c)
    wi = np.asarray(wi).reshape(-1)
    # state layout (peak:166-169): peak+, cont+, peak-, cont-
    cp = wi[n_p:n_p + n] - prob.wi_0
    pos[c, :n] = cp
    if two:
        cn = wi[2 * n_p + n:2 * n_p + 2 * n] - prob.wi_0
        pos[c, n:] = cn + prob.n_grid
    win[c] = w[0]
rows = prob.n_grid * (2 if two else 1)
with open(out, "wb") as f:
    np.array([chains, per, rows, 0], dtype=np.int32).tofile(f)
    pos.tofile(f)
    win.tofile(f)
print("wrote", out, "rows", rows, "mean window", win.mean(), "distinct columns per chain", np.mean([len(np.unique(p)) for p in pos[:64]]))
