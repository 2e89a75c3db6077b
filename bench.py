#!/usr/bin/env python3
"""Headline benchmark: accepted delta-moves/s of the sac/free fixed-Theta sampling stage.

Workload (BASELINE.json configs[1]): sac/free bosonic, N_tau=64,
2000 delta functions, 10^5+1-point omega grid, 4096 independent chains per MI355X, Theta=1,
windows pre-tuned by adjust_windows.  A "step" = one `sample` stage of SWEEPS_PER_STEP sweeps of
every chain (proposals, K-column gathers, chi^2, Metropolis, S(omega) histogram, <chi^2>), followed
for N>1 by the per-stage RCCL all-reduce that merges histogram and chi^2 statistics.

    python bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TAU, N_OMEGA, OMEGA_M, D_OMEGA = 64, 2000, 10.0, 1e-4
CHAINS_PER_GPU = 4096
SWEEPS_PER_STEP = 400
THETA = 1.0
SEED = 12345
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def cpu_baseline(prob, windows, wi, theta=THETA, seconds=12.0):
    """The oracle (CPU port of the reference loops, -O2 -march=native, 1 thread) on the same
    problem, same Theta, same tuned windows: sample sweeps for ~`seconds` s."""
    from oracle import oracle as ora
    o = ora.OracleFree(prob.K, prob.td.G, prob.td.sigma_inv, prob.n_omega, prob.wi_0, prob.wi_m)
    o.set_state(wi, prob.A, windows)
    o.init_gbar()
    o.use_philox(SEED, 1 << 30)
    o.sample(5, theta)
    c0 = o.counters()
    t0 = time.perf_counter()
    sweeps = 0
    while time.perf_counter() - t0 < seconds:
        o.sample(50, theta)
        sweeps += 50
    dt = time.perf_counter() - t0
    c1 = o.counters()
    return {"value": (c1[3] - c0[3]) / dt, "unit": "accepted delta-moves/s", "cores": 1, "kind": "port",
            "sample": f"{sweeps} sample sweeps of one chain ({dt:.1f} s) on the bench problem at the tuned windows",
            "sweeps_per_s": sweeps / dt, "proposals_per_s": (c1[0] - c0[0]) / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU)
    ap.add_argument("--sweeps-per-step", type=int, default=SWEEPS_PER_STEP)
    ap.add_argument("--d-omega", type=float, default=D_OMEGA)
    ap.add_argument("--sweeps-per-launch", type=int, default=0, help="0 = the library's default")
    ap.add_argument("--theta", type=float, default=THETA, help="sampling temperature (headline: 1.0)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # SAC_BENCH_REHEARSE=1 (development only, never a reported number): ranks share the GPUs that exist and merge
    # over gloo, to walk the N>1 code path on a one-GPU box; the driver's runs use RCCL, one GPU per rank.
    rehearse = os.environ.get("SAC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if rehearse:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    from sac_amd.free import FreeProblem
    from sac_amd.synth import synthetic_timedata
    td = synthetic_timedata(n_tau=N_TAU, seed=SEED)
    prob = FreeProblem(td, N_OMEGA, 0.0, OMEGA_M, args.d_omega, par=1, kernel_type="bosonic")
    # chains are block-partitioned over ranks; the Philox key holds the global chain id
    h = prob.make_handle(n_chains=args.chains, seed=SEED, chain_offset=rank * args.chains,
                         device=local_rank, sweeps_per_launch=args.sweeps_per_launch)
    theta = args.theta
    h.adjust_windows(1000, theta)            # equilibrate + tune the windows (untimed; 10 rounds of 100 sweeps, so that
                                             # every launch of the run has the size of the timed ones)

    hist_dev = torch.zeros(prob.n_grid, dtype=torch.float64, device="cuda")
    stat_dev = torch.zeros(8, dtype=torch.float64, device="cuda")
    min_dev = torch.zeros(1, dtype=torch.float64, device="cuda")

    def step():
        h.sample(args.sweeps_per_step, theta)
        if world > 1:
            # per-stage merge (the only collective on the path): histogram + chi^2 statistics
            h.histogram_sum_to_device(hist_dev.data_ptr())
            st = h.stats()
            stat_dev.copy_(torch.tensor([st.sampled_chi2, *st.accept_rates, 1.0], dtype=torch.float64))
            min_dev.fill_(st.chi2_min)
            all_reduce(hist_dev)
            all_reduce(stat_dev)
            all_reduce(min_dev, op=dist.ReduceOp.MIN)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    s0 = h.stats()
    kernel_ms, launches = 0.0, 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = h.stats()
        kernel_ms += st.last_kernel_ms
        launches += st.last_kernel_launches
    fence()
    dt = time.perf_counter() - t0
    s1 = h.stats()

    local = torch.tensor([dt, s1.accepted_moves - s0.accepted_moves, s1.proposals - s0.proposals,
                          s1.sweeps - s0.sweeps, s1.columns_gathered - s0.columns_gathered,
                          s1.evaluated - s0.evaluated], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = local[:1].clone()
        all_reduce(tmax, op=dist.ReduceOp.MAX)
        all_reduce(local)
        local[0] = tmax[0]
    dt_max, moves, props, sweeps, cols, evald = (float(x) for x in local.cpu())

    if rank == 0:
        # roofline of the dominant kernel (free_stage_kernel), this rank, measured with HIP events
        # on the library's stream around every launch of the timed region.
        cols_r0 = s1.columns_gathered - s0.columns_gathered
        sweeps_r0 = s1.sweeps - s0.sweeps
        alg_bytes = cols_r0 * N_TAU * 8 + sweeps_r0 * N_OMEGA * 16   # K columns + histogram RMW
        per_launch_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = alg_bytes / max(launches, 1) / per_launch_s / 1e9
        traffic = None
        try:    # PMC-derived traffic of the same launch shape, collected in separate rocprofv3 passes
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            w = t["workload"]
            if (w["n_tau"], w["n_omega"], w["n_grid"], w["chains"]) == (N_TAU, N_OMEGA, prob.n_grid, args.chains):
                # per launch of THIS run: the profiled launch scaled by the sweeps a launch ran here
                sweeps_per_launch_here = sweeps_r0 / args.chains / max(launches, 1)
                traffic = ((2.0 * t["fetch_size_kb_median"] + t["write_size_kb_median"]) * 1024.0
                           * sweeps_per_launch_here / w["sweeps_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
        wi0, _, win0 = h.get_state(0)
        out = {
            "metric": "accepted delta-moves/sec (whole node) at N_tau=64, N_delta=2000",
            "value": moves / dt_max, "unit": "accepted delta-moves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: ranks share a GPU, gloo merge -- not a result)",
            "config": {"workload": "sac/free bosonic, N_tau=64, 2000 delta-functions, "
                                   f"{prob.n_grid}-point omega grid, {args.chains} chains per MI355X, Theta={theta:g} "
                                   "(BASELINE.json configs[1])",
                       "chains_per_gpu": args.chains, "sweeps_per_step": args.sweeps_per_step,
                       "theta": theta, "parallelism": f"chains sharded over {world} GPU(s), "
                                                      "one RCCL all-reduce per stage"},
            "sweeps_per_s": sweeps / dt_max, "proposals_per_s": props / dt_max,
            "acceptance": {"evaluated_frac": evald / props, "accepted_moves_per_proposal": moves / props},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "bytes per launch beyond L2, (2*FETCH_SIZE+WRITE_SIZE)*1024 from profiles/traffic.json",
                         "kernel": "free_stage_kernel<1,false,true,false>", "launches": launches,
                         "avg_launch_ms": per_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes / max(launches, 1)},
        }
        if world == 1 and not args.no_cpu:
            from oracle import oracle as ora
            ora.lib()
            cb = cpu_baseline(prob, win0, wi0, theta)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_core"] = out["value"] / cb["value"]
        print(json.dumps(out))
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
