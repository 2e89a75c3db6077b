#!/usr/bin/env python3
"""Complete reference runs per GPU: R independent runs (own handle, own seed, own output folder, the reference's FULL schedule)
side by side on one MI355X, P processes x T host threads, next to the wall-clock the reference's log.txt records for ONE such
run on one CPU core.  This is the reference's own way of scaling -- one Julia process per parameter point / replica
(qsub_files/**, cross_validation/run_cv.sh) -- with the GPU as the node: a run is a latency-bound chain of small launches that
leaves the chip empty, so runs are stacked until the wave slots are used.

  python tools/runs_per_gpu.py free|edge PROCS THREADS [CHAINS]      one JSON line"""
import json
import multiprocessing as mp
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
REFERENCE_S = {"free": 40.1, "edge": 72.2}      # log.txt of the reference runs, first to last stamp (one chain, one core)


def one_run(kind, idx, chains, folder):
    import numpy as np
    t0 = time.time()
    if kind == "free":
        from sac_amd.free import FreeSAC
        s = FreeSAC.from_infile(f"{G}/free_bdp/in_free.in", n_chains=chains, seed=1000 + idx, device_tables=True)
        s.G_file, s.output_folder = f"{G}/free_bdp/t.in", folder
        theta = s.run()
        row = np.loadtxt(s.sample_file, delimiter=",", skiprows=1, ndmin=2)[0]
        res = {"theta_opt": float(theta), "chi2_avg": float(row[4])}
    else:
        from sac_amd.edge import EdgeSAC
        s = EdgeSAC.from_infile(f"{G}/edge/hchain_single/in_edge.in", n_chains=chains, seed=1000 + idx)
        s.G_file, s.output_folder, s.d_omega = f"{G}/edge/hchain_single/t.in", folder, 1e-4
        theta = s.run()
        row = np.genfromtxt(os.path.join(folder, "sample.csv"), delimiter=",", skip_header=1)
        res = {"theta_opt": float(theta), "chi2_avg": float(row[3]), "edge": float(row[5])}
    s.h.close()
    res["seconds"] = round(time.time() - t0, 1)
    return res


def proc_main(kind, p, threads, chains, base, q):
    with ThreadPoolExecutor(threads) as ex:
        futs = [ex.submit(one_run, kind, p * threads + t, chains, f"{base}/run_{p:02d}_{t:02d}") for t in range(threads)]
        q.put([f.result() for f in futs])


if __name__ == "__main__":
    kind, procs, threads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    chains = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    base = f"/tmp/runs_per_gpu_{kind}"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    ps = [ctx.Process(target=proc_main, args=(kind, p, threads, chains, base, q)) for p in range(procs)]
    for p in ps:
        p.start()
    rows = []
    for _ in ps:
        while True:
            try:
                rows += q.get(timeout=60)
                break
            except Exception:
                print(f"... {time.time() - t0:.0f} s", file=sys.stderr, flush=True)     # a silent run is taken to be hung
                if not any(p.is_alive() for p in ps) and q.empty():
                    raise SystemExit("a worker died")
    for p in ps:
        p.join()
    wall = time.time() - t0
    import numpy as np
    th = np.array([r["theta_opt"] for r in rows])
    out = {"what": f"complete sac/{kind} reference runs side by side on one MI355X", "runs": len(rows), "processes": procs, "threads_per_process": threads,
           "chains_per_run": chains, "wall_s": round(wall, 1), "seconds_per_run": round(wall / len(rows), 2),
           "slowest_run_s": max(r["seconds"] for r in rows), "fastest_run_s": min(r["seconds"] for r in rows),
           "reference_seconds_per_run_one_core": REFERENCE_S[kind], "runs_per_reference_run_time": round(REFERENCE_S[kind] / (wall / len(rows)), 1),
           "theta_opt_quartiles": [float(x) for x in np.percentile(th, [25, 50, 75])],
           "chi2_avg_mean": float(np.mean([r.get("chi2_avg", float("nan")) for r in rows]))}
    print(json.dumps(out))
