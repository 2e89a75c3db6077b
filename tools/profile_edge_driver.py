import cProfile, io, os, pstats, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sac_amd.edge import EdgeSAC
G = "/root/repo/tests/golden"
def run(n_chains, seed):
    s = EdgeSAC.from_infile(f"{G}/edge/hchain_single/in_edge.in", n_chains=n_chains, seed=seed)
    s.G_file = f"{G}/edge/hchain_single/t.in"
    s.output_folder = f"/tmp/edge_prof_{n_chains}"
    s.d_omega = 1e-4
    s.f_anneal = 1.05 ** 5
    s.N_anneal, s.anneal_steps, s.sample_steps = 30, 300, 100
    s.chi2_anneal = np.zeros(s.N_anneal)
    kms = [0.0]
    orig = s.run_bins
    def rb(*a, **k):
        r = orig(*a, **k)
        kms[0] += s.h.results(-1)["last_stage_ms"]
        return r
    s.run_bins = rb
    t0 = time.time()
    th = s.run()
    print("chains", n_chains, "wall", round(time.time() - t0, 2), "stage ms total", round(kms[0] / 1e3, 2), flush=True)
pr = cProfile.Profile()
pr.enable()
run(64, 3)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(30)
print(out.getvalue()[:7000])
