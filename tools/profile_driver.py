#!/usr/bin/env python3
"""Where the wall-clock of a driver run goes on the host: cProfile around the main anneal of the reference's free run
(tests/golden/free_bdp) on 64 chains, next to the kernel time the library reports.   python tools/profile_driver.py [chains]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from sac_amd.free import FreeSAC
G = os.path.join(ROOT, "tests", "golden")
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = FreeSAC.from_infile(f"{G}/free_bdp/in_free.in", n_chains=chains, seed=7)
s.G_file = f"{G}/free_bdp/t.in"
s.output_folder = "/tmp/profile_driver"
s.anneal_steps //= 2
s.N_anneal = 40
s.chi2_anneal = np.zeros(s.N_anneal)
s.init_outfiles()
t0 = time.time()
s.initialize()
print("initialize", round(time.time() - t0, 2), "s", flush=True)
s.h.set_state(windows=np.full(4, s.prob.window))
kms = [0.0]
orig_adj, orig_smp = s.h.adjust_windows, s.h.sample
def adj(*a, **k):
    r = orig_adj(*a, **k); kms[0] += s.h.stats().last_kernel_ms; return r
def smp(*a, **k):
    r = orig_smp(*a, **k); kms[0] += s.h.stats().last_kernel_ms; return r
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
s.fast_anneal()
s.main_anneal(False)
pr.disable()
wall = time.time() - t0
print("fast + main anneal (40 points):", round(wall, 2), "s wall", flush=True)
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
print(out.getvalue()[:6000])
# the same again, kernel time summed from the library's HIP events
s2 = FreeSAC.from_infile(f"{G}/free_bdp/in_free.in", n_chains=chains, seed=7)
s2.G_file, s2.output_folder = f"{G}/free_bdp/t.in", "/tmp/profile_driver2"
s2.anneal_steps //= 2
s2.N_anneal = 40
s2.chi2_anneal = np.zeros(40)
s2.init_outfiles(); s2.initialize(); s2.h.set_state(windows=np.full(4, s2.prob.window))
tk = 0.0
t0 = time.time()
theta = s2.theta_0
for i in range(40):
    s2.h.adjust_windows(s2.anneal_steps, theta); tk += s2.h.stats().last_kernel_ms
    s2.h.sample(s2.anneal_steps, theta); tk += s2.h.stats().last_kernel_ms
    theta /= s2.f_anneal
print("bare stage calls, 40 points:", round(time.time() - t0, 2), "s wall, kernel time", round(tk / 1e3, 2), "s")
