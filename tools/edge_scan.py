import sys, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import bench_param as b
for dw in (1e-5, 1e-3):
    for ch in (512, 1024, 2048, 4096, 8192):
        r = b.bench_edge(ch, d_omega=dw)
        print(dw, ch, round(r["chain_sweeps_per_s"]), round(r["chain_sweeps_per_s"]/ch), round(r["algorithmic_GBps"]), flush=True)
