#!/usr/bin/env python3
"""Cross-validation units of one rank run one after the other and side by side (run_cv_partition(side_by_side=K)): same GSAC.csv
files byte for byte, shorter wall-clock.  Four units (free, replicas 1-2, halves a/b) of the reference's bosonic_delta_peak
cross-validation input, 60 Theta points of 200 + 200 sweeps, 8 chains each.   python tools/side_by_side_check.py [K]"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from sac_amd.cross_val import cv_units, run_cv_partition
CV = os.path.join(ROOT, "tests", "golden", "cv")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = open(f"{CV}/run001a_free_in_free.in").read().split("\n")
rows[3] = "60 200 200"
out = {}
for tag, k in (("one_by_one", 1), ("side_by_side", K)):
    base = f"/tmp/side_by_side_{tag}"
    os.makedirs(base, exist_ok=True)
    rows[4] = f"{CV} {base}/out"
    open(f"{base}/in_free.in", "w").write("\n".join(rows))
    units = cv_units((1, 2), ("free",))
    t0 = time.time()
    folders = run_cv_partition(units, side_by_side=k, in_file=f"{base}/in_free.in", n_chains=8, seed=3, device_tables=True)
    out[tag] = {"seconds": round(time.time() - t0, 1), "gsac": {str(u): open(f + "/GSAC.csv").read() for u, f in folders.items()}}
same = out["one_by_one"]["gsac"] == out["side_by_side"]["gsac"]
print(json.dumps({"units": 4, "side_by_side": K, "seconds_one_by_one": out["one_by_one"]["seconds"],
                  "seconds_side_by_side": out["side_by_side"]["seconds"], "gsac_files_identical": same,
                  "rows_per_file": out["one_by_one"]["gsac"][next(iter(out["one_by_one"]["gsac"]))].count("\n")}))
sys.exit(0 if same else 1)
