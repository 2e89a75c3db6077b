"""Command line twin of the reference's program entries (free:1264-1268, peak:1571-1580, edge:2068-2075,
cross_val.jl:514-531): run from a directory holding in_free.in / in_peak.in / in_edge.in.

    python -m sac_amd free  [--chains N] [--seed S]
    python -m sac_amd peak  [A_0 N_p] [--chains N]
    python -m sac_amd edge  [A_c A_r p] [--chains N]
    python -m sac_amd cv free|peak|edge REP a|b [--chains N]
    python -m torch.distributed.run --nproc-per-node G -m sac_amd free|peak|edge ...
        (the run's chains sharded over G GPUs, --chains per GPU: one RCCL all-reduce per stage merges the statistics
         and every rank takes the same schedule decisions; only rank 0 writes files.  The reference's fan-out is one
         Julia process per parameter point, run_cv.sh:5-9 / qsub_files/edge/submit.sh:25-51)
    python -m torch.distributed.run --nproc-per-node G -m sac_amd cv-all REP_FIRST REP_LAST [free,peak,edge]
        (replicas x halves x parameterisations round-robin over the G GPUs; in-files in the working directory)
    python -m torch.distributed.run --nproc-per-node G -m sac_amd scan edge p|A_c|A_r V_0 DELTA N [A_c A_r p]
    python -m torch.distributed.run --nproc-per-node G -m sac_amd scan peak A_0 V_0 DELTA N [N_p]
        (the parameter scans of qsub_files/edge/submit.sh:25-51 and qsub_files/peak/qsub_a0_scan.sh: value = V_0 + DELTA (i - 1),
         one whole GPU per scan point, round-robin; the trailing values fix the other command-line parameters)
"""
import argparse
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m sac_amd")
    ap.add_argument("program", choices=["free", "peak", "edge", "cv", "cv-all", "scan"])
    ap.add_argument("args", nargs="*")
    ap.add_argument("--chains", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--side-by-side", type=int, default=1, help="scan / cv-all: units a rank runs at a time on its GPU, one host "
                    "thread each (complete runs with few chains leave the GPU empty; 3 or 4 per process is what its queues carry)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend when launched with WORLD_SIZE > 1 "
                                                       "(nccl = RCCL; gloo for rehearsals on shared GPUs)")
    a = ap.parse_args(argv)
    kw = dict(n_chains=a.chains, seed=a.seed, device=a.device)
    if a.program in ("free", "peak", "edge"):
        import os
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:                                   # one process per GPU, chains sharded over the ranks
            import torch
            import torch.distributed as dist
            from .dist import Comm
            kw["device"] = int(os.environ.get("LOCAL_RANK", str(a.device))) % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(kw["device"])
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", kw["device"]))
                kw["comm"] = Comm(device=torch.device("cuda", kw["device"]))
            else:
                dist.init_process_group(a.backend)
                kw["comm"] = Comm()
        try:
            if a.program == "free":
                from .free import FreeSAC
                return FreeSAC.from_infile("in_free.in", **kw).run("in_free.in")
            if a.program == "peak":
                from .peak import PeakSAC
                A_0 = float(a.args[0]) if len(a.args) > 0 else None
                N_p = int(a.args[1]) if len(a.args) > 1 else None
                return PeakSAC.from_infile("in_peak.in", A_0_in=A_0, N_p_in=N_p, **kw).run("in_peak.in")
            from .edge import EdgeSAC
            v = [float(x) for x in a.args[:3]] + [None] * (3 - len(a.args[:3]))
            return EdgeSAC.from_infile("in_edge.in", A_c_in=v[0], A_r_in=v[1], p_in=v[2], **kw).run("in_edge.in")
        finally:
            if world > 1:
                dist.destroy_process_group()
    if a.program == "scan":
        import os
        from .scan import best_point, run_scan_partition, scan_units, scan_values
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        kw["device"] = int(os.environ.get("LOCAL_RANK", str(a.device)))
        if len(a.args) < 5:
            ap.error("scan needs: edge|peak AXIS V_0 DELTA N [fixed values]")
        prog, axis = a.args[0], a.args[1]
        values = scan_values(a.args[2], a.args[3], int(a.args[4]))        # the argument strings: exact decimals, as bc has them
        rest = [float(x) for x in a.args[5:]]
        names = ("A_c", "A_r", "p") if prog == "edge" else ("N_p",)
        fixed = {n: v for n, v in zip(names, rest) if n != axis}
        units = scan_units(prog, axis, values, **fixed)
        if world > 1:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(kw["device"])
            from .cross_val import GATHER_TIMEOUT
            # scan points are whole runs (minutes each) dealt round-robin: the ranks reach the final gather far apart
            dist.init_process_group(a.backend, timeout=GATHER_TIMEOUT,
                                    **({"device_id": torch.device("cuda", kw["device"])} if a.backend == "nccl" else {}))
        results = run_scan_partition(units, rank, world, side_by_side=a.side_by_side, **kw)
        if rank == 0:
            for u in units:
                print(dict(u[1]), results[u])
            print("best:", dict(best_point(results)[0][1]))
        if world > 1:
            dist.destroy_process_group()
        return results
    if a.program == "cv-all":
        import os
        from .cross_val import cv_units, run_cv_partition
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        kw["device"] = int(os.environ.get("LOCAL_RANK", str(a.device)))
        if world > 1:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(kw["device"])
            from .cross_val import GATHER_TIMEOUT
            dist.init_process_group("nccl", timeout=GATHER_TIMEOUT, device_id=torch.device("cuda", kw["device"]))
        if len(a.args) < 2:
            ap.error("cv-all needs: REP_FIRST REP_LAST [free,peak,edge]")
        params = tuple(a.args[2].split(",")) if len(a.args) > 2 else ("free", "peak", "edge")
        units = cv_units(range(int(a.args[0]), int(a.args[1]) + 1), params)
        folders = run_cv_partition(units, rank, world, side_by_side=a.side_by_side, **kw)
        if rank == 0:
            for u in units:
                print(u, folders[u])
        if world > 1:
            dist.destroy_process_group()
        return folders
    from .cross_val import edge_cv, free_cv, peak_cv
    if len(a.args) < 3:
        ap.error("cv needs: free|peak|edge REP a|b")
    fn = {"free": free_cv, "peak": peak_cv, "edge": edge_cv}[a.args[0]]
    return fn(int(a.args[1]), a.args[2], **kw)


if __name__ == "__main__":
    main()
    sys.exit(0)
