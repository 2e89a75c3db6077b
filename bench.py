#!/usr/bin/env python3
"""Headline benchmark: accepted delta-moves/s of the sac/free fixed-Theta sampling stage.

Workload (BASELINE.json configs[1]): sac/free bosonic, N_tau=64,
2000 delta functions, 10^5+1-point omega grid, 4096 independent chains per MI355X, Theta=1,
windows pre-tuned by adjust_windows.  A "step" = one `sample` stage of SWEEPS_PER_STEP sweeps of
every chain (proposals, K-column gathers, chi^2, Metropolis, S(omega) histogram, <chi^2>), followed
for N>1 by the per-stage RCCL all-reduce that merges histogram and chi^2 statistics.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment the parent starts the N ranks itself (a child
`python -m torch.distributed.run ... bench.py`, before anything here touches the GPU) and relays rank 0's line;
under `torch.distributed.run` (RANK / WORLD_SIZE set) it is one rank.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
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
# measured gather-only ceilings (GB/s) on the SAMPLER'S OWN column distribution (tools/gather_probe --positions on the delta
# positions and tuned windows of a run at this shape, 8 B per lane as the kernel loads, 4096 waves, 16 rows in flight; round 3:
# profiles/r03_gather_probe.txt, which also holds the deliberately flat-support case and the 16 B/lane form): the roof of the
# kernel's access pattern on that table, by table size.  The 48.8 MiB table is L2 / Infinity-Cache resident; of the 488 MiB
# table at most 256 MB (52 %) can sit in the Infinity Cache, of the 1953 MiB table at most 13 %.
GATHER_CEILING_GBS = 16467.5
GATHER_CEILINGS_BY_GRID = {1000001: 9728.6, 4000001: 8133.9}
GATHER_CEILING_SOURCE = "profiles/r03_gather_probe.txt"
# the same probe on the sac/peak samplers' column distributions (tools/dump_positions_peak.py; the continuum deltas of a peak run
# crowd behind the peak: fewer distinct columns per chain, hence higher ceilings than sac/free's on the same table size);
# "two" = both axes, the two 488 MiB tables as one 976.6 MiB footprint
PEAK_GATHER_CEILINGS = {1000001: 11607.4, 4000001: 8077.6, "two_axes_1000001": 12543.0}
PEAK_GATHER_CEILING_SOURCE = "profiles/r03_gather_probe_peak.txt"
N_SIMD = 256 * 4          # MI355X: 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4           # peak engine clock
# a wave64 VALU instruction occupies its SIMD for one "quad-cycle" (4 clocks: 16 lanes per clock); SQ_ACTIVE_INST_VALU
# counts those.  Ceiling of VALU issue on the whole chip, in G quad-cycles per second:
VALU_ISSUE_PEAK = N_SIMD * CLOCK_GHZ / 4.0


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (CPU port of the reference loops), P concurrent single-thread processes
# ------------------------------------------------------------------------------------------------

def cpu_worker(state_dir, idx, seconds):
    """One single-thread process of the CPU baseline: the oracle's `sample` sweeps on the bench problem for
    ~`seconds` s (own Philox stream).  Prints one JSON line."""
    from oracle import oracle as ora
    K = np.load(os.path.join(state_dir, "K.npy"), mmap_mode="r")
    st = np.load(os.path.join(state_dir, "state.npz"))
    o = ora.OracleFree(np.ascontiguousarray(K), st["G"], st["sigma_inv"], int(st["n_omega"]), int(st["wi_0"]), int(st["wi_m"]))
    o.set_state(st["wi"], st["A"], st["windows"])
    o.init_gbar()
    o.use_philox(SEED, (1 << 30) + idx)
    theta = float(st["theta"])
    o.sample(5, theta)
    c0 = o.counters()
    t0 = time.perf_counter()
    sweeps = 0
    while time.perf_counter() - t0 < seconds:
        o.sample(50, theta)
        sweeps += 50
    dt = time.perf_counter() - t0
    c1 = o.counters()
    print(json.dumps({"moves_per_s": (c1[3] - c0[3]) / dt, "sweeps_per_s": sweeps / dt,
                      "proposals_per_s": (c1[0] - c0[0]) / dt, "sweeps": sweeps, "seconds": dt}))


def cpu_baseline(prob, windows, wi, theta=THETA, seconds=10.0, procs=None):
    """P independent single-thread oracle processes (-O2 -march=native) running concurrently on the box's host
    cores, each one chain of the same problem at the same Theta and tuned windows; aggregate rate."""
    if procs is None:
        procs = min(16, len(os.sched_getaffinity(0)))      # the GPU box gives one GPU's job 16 host cores
    with tempfile.TemporaryDirectory(prefix="sac_cpu_") as d:
        np.save(os.path.join(d, "K.npy"), prob.K)
        np.savez(os.path.join(d, "state.npz"), G=prob.td.G, sigma_inv=prob.td.sigma_inv, n_omega=prob.n_omega,
                 wi_0=prob.wi_0, wi_m=prob.wi_m, wi=wi, A=prob.A, windows=windows, theta=theta)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", d, str(i), str(seconds)],
                               stdout=subprocess.PIPE, env=env, text=True) for i in range(procs)]
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
    tot = {k: sum(o[k] for o in outs) for k in ("moves_per_s", "sweeps_per_s", "proposals_per_s", "sweeps")}
    return {"value": tot["moves_per_s"], "unit": "accepted delta-moves/s", "cores": procs, "kind": "port",
            "sample": f"{procs} concurrent single-thread processes of the CPU port (oracle/), one chain each, "
                      f"{tot['sweeps']} sample sweeps in total ({max(o['seconds'] for o in outs):.1f} s) on the bench problem "
                      "at the tuned windows",
            "sweeps_per_s": tot["sweeps_per_s"], "proposals_per_s": tot["proposals_per_s"],
            "per_core_value": tot["moves_per_s"] / procs}


def load_traffic_json():
    """profiles/traffic.json (rocprofv3 --pmc passes) if it was measured on the library that is running: (dict, None), else
    (None, reason).  The file records sac_build_id() of the library under the profiler; a kernel edit without a new PMC
    pass must not keep the old numerators."""
    from sac_amd.lib import build_id
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError) as ex:
        return None, f"profiles/traffic.json not readable ({ex})"
    have, want = pmc.get("build_id"), build_id()
    if have != want:
        return None, f"profiles/traffic.json was measured on build {have}, this run uses build {want}: counter-based numbers withheld"
    return pmc, None


# ------------------------------------------------------------------------------------------------
# HBM-resident regime, driver-timed in the same run (extra keys; the headline table is cache resident)
# ------------------------------------------------------------------------------------------------

def hbm_regime(device, chains, theta=THETA):
    """Two short `sample` stages each of sac/free and of BASELINE configs[3]'s sac/peak shape on a 10^6+1-column and on a
    4*10^6+1-column table (488 MiB: half of it fits the 256 MB Infinity Cache; 1.9 GiB: HBM), tables built on the device.
    algorithmic bytes = gathered K columns x N_tau x 8 B; `achieved` = the bytes that crossed the fabric per launch by the
    rocprofv3 counters (profiles/traffic.json, same launch shape) over the HIP-event launch time of this run; `frac` is
    stated only where HBM is the roof (the 1.9 GiB tables)."""
    from sac_amd.free import FreeProblem
    from sac_amd.peak import PeakProblem
    from sac_amd.synth import synthetic_timedata
    td = synthetic_timedata(n_tau=N_TAU, seed=SEED)
    out = []

    pmc_all, pmc_why = load_traffic_json()
    pmc = {} if pmc_all is None else pmc_all.get("hbm_regime", {})

    def run(name, key, h, stats, sweeps, table_bytes, rebuild_cols=0, ceil=None, ceil_source=None, pmc_entry=None):
        h.adjust_windows(200, theta)
        h.sample(sweeps, theta)
        s0 = stats()
        kms, nl = 0.0, 0
        for _ in range(2):
            h.sample(sweeps, theta)
            s = stats()
            kms += s.last_kernel_ms
            nl += s.last_kernel_launches
        # the library counts the columns the proposals gather; sac/peak also rebuilds Gbar from all its columns before every
        # sweep (peak:1072), which is table traffic too
        cols = s.columns_gathered - s0.columns_gathered + rebuild_cols * (s.sweeps - s0.sweeps)
        alg = cols * N_TAU * 8 / (kms / 1e3) / 1e9
        e = {"workload": name, "bound": "hbm", "launches": nl, "avg_launch_ms": kms / nl, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "algorithmic_GBps": alg, "algorithmic_over_hbm_peak": alg / HBM_PEAK_GBS,
             "algorithmic_bytes_per_launch": cols * N_TAU * 8 / nl}
        # `achieved` = the ALGORITHMIC gather rate (columns the proposals read x N_tau x 8 B over the HIP-event launch time of
        # this run); `frac` = against the MEASURED gather-only ceiling of this access pattern on this table size
        # (GATHER_CEILINGS_BY_GRID); the nominal 8 TB/s HBM peak is the second key.  The counter traffic beyond L2 rides
        # along only when profiles/traffic.json was measured on the build that is running.
        e.update({"achieved": alg, "measured_gather_ceiling_GBps": ceil, "measured_gather_ceiling_source": ceil_source,
                  "frac": None if ceil is None else alg / ceil, "frac_of_nominal_hbm_peak": alg / HBM_PEAK_GBS,
                  "infinity_cache_share_upper_bound": min(1.0, 256e6 / table_bytes),
                  "frac_note": "algorithmic rate (for sac/peak including the per-sweep Gbar rebuild) over the gather-only ceiling measured on "
                               "this sampler's own column distribution on a table of this size (8 B per lane, 4096 waves, 16 rows in flight)"})
        t = pmc_entry if pmc_entry is not None else pmc.get(key)
        if t and sweeps == 100 and h.n_chains == 4096:
            cnt = t["bytes_per_launch"] / (kms / nl / 1e3) / 1e9
            e.update({"traffic": t["bytes_per_launch"], "counter_GBps_beyond_L2": cnt,
                      "traffic_note": "(2*FETCH_SIZE+WRITE_SIZE)*1024 per launch from profiles/traffic.json (same build), over the launch "
                                      "time of this run; FETCH_SIZE includes Infinity-Cache hits"})
        else:
            e.update({"traffic": None, "traffic_note": pmc_why or "no counter pass for this launch shape"})
        e.update({"chain_sweeps_per_s": (s.sweeps - s0.sweeps) / (kms / 1e3),
                  "accepted_moves_per_s": (s.accepted_moves - s0.accepted_moves) / (kms / 1e3)})
        out.append(e)
        h.close()

    for d_omega in (1e-5, 2.5e-6):
        prob = FreeProblem(td, N_OMEGA, 0.0, OMEGA_M, d_omega, par=1, kernel_type="bosonic")
        h = prob.make_handle(n_chains=chains, seed=SEED, device=device, device_tables=True)
        run(f"sac/free bosonic, N_tau=64, 2000 deltas, {prob.n_grid}-column table ({prob.n_grid * 512 / 2**20:.0f} MiB), "
            f"{chains} chains", f"free_{prob.n_grid}", h, h.stats, 100, prob.n_grid * 512,
            ceil=GATHER_CEILINGS_BY_GRID.get(prob.n_grid), ceil_source=GATHER_CEILING_SOURCE)
    for d_omega in (1e-5, 2.5e-6):
        prob = PeakProblem(td, 1000, 1, 0.5, OMEGA_M, d_omega, kernel_type="bosonic")
        h = prob.make_handle(n_chains=chains, seed=SEED, device=device, device_tables=True)
        run(f"sac/peak bosonic, N_tau=64, 1000+1 deltas, {prob.n_grid}-column table ({prob.n_grid * 512 / 2**20:.0f} MiB), "
            f"{chains} chains" + (" (BASELINE.json configs[3])" if d_omega == 1e-5 else ""), f"peak_{prob.n_grid}", h, h.peak_stats, 100,
            prob.n_grid * 512, rebuild_cols=1001, ceil=PEAK_GATHER_CEILINGS.get(prob.n_grid), ceil_source=PEAK_GATHER_CEILING_SOURCE)
    # sac/peak on both frequency axes with the weight-transfer moves (fermionic finite-T; the workload of tools/bench_param.py peak2
    # and of its PMC passes, same seeds): two 488 MiB tables
    t2 = synthetic_timedata(n_tau=N_TAU, seed=777, kernel_type="finiteT")
    prob = PeakProblem(t2, 1000, 1, 0.5, OMEGA_M, 1e-5, kernel_type="finiteT", approx_ratio=2.5)
    h = prob.make_handle(n_chains=chains, seed=777, device=device)
    run(f"sac/peak finite-T, both axes with weight transfers, N_tau=64, 2 x (1000+1) deltas, two {prob.n_grid}-column tables (977 MiB), "
        f"{chains} chains", "peak_two_axes", h, h.peak_stats, 100, 2 * prob.n_grid * 512, rebuild_cols=2002,
        ceil=PEAK_GATHER_CEILINGS["two_axes_1000001"], ceil_source=PEAK_GATHER_CEILING_SOURCE,
        pmc_entry=None if pmc_all is None else pmc_all.get("other", {}).get("peak_two_axes"))
    # BASELINE configs[2]'s kernel on one GPU: sac/edge single_edge, N_e=80, K / dK / d2K tables of 10^6+1 columns (1465 MiB),
    # one 40-sweep launch per bin (the workload of tools/bench_param.py edge and of the PMC passes, same seeds)
    from sac_amd.edge import EdgeProblem
    te = synthetic_timedata(n_tau=N_TAU, seed=777)
    prob = EdgeProblem(te, "single_edge", "bosonic", 80, 0, 0.5, 0.0, 1.0, 0.0, 10.0, 1e-5)
    h = prob.make_handle(n_chains=chains, seed=777, device=device)
    h.run_bins(20, 2, theta)
    r0 = h.results(-1)
    h.run_bins(40, 4, theta)
    r1 = h.results(-1)
    ms = r1["last_stage_ms"]
    look = r1["lookups"] - r0["lookups"]
    tab = look * 3 * N_TAU * 8 / (ms / 1e3) / 1e9
    e = {"workload": f"sac/edge single_edge bosonic, N_tau=64, N_e=80, 3 x {prob.kernel.n_grid}-column tables (1465 MiB), {chains} chains "
                     "(BASELINE.json configs[2] on one GPU)", "bound": "hbm (random 1536-B row gathers)", "launches": 4, "avg_launch_ms": ms / 4,
         "peak": HBM_PEAK_GBS, "unit": "GB/s", "table_GBps": tab, "chain_sweeps_per_s": chains * 160 / (ms / 1e3),
         "lookups_per_s": look / (ms / 1e3), "gather_ceiling_GBps": 7200.0,
         "gather_ceiling_source": "profiles/r02_gather_probe_edge.txt (dependent random 1536-B rows of a 1465 MiB table, 4096 waves)"}
    t = None if pmc_all is None else pmc_all.get("other", {}).get("edge")
    if t and chains == 4096:
        ach = t["bytes_per_launch"] / (ms / 4 / 1e3) / 1e9
        e.update({"achieved": ach, "traffic": t["bytes_per_launch"], "frac": ach / HBM_PEAK_GBS, "frac_of_gather_ceiling": ach / 7200.0,
                  "achieved_note": "counter traffic beyond L2 per 40-sweep launch (table rows + the chains' K_curr / K_new columns), "
                                   "profiles/traffic.json, over the launch time of this run"})
    else:
        e.update({"achieved": None, "traffic": None, "frac": None, "table_over_gather_ceiling": tab / 7200.0,
                  "achieved_note": pmc_why or "no counter pass for this launch shape: only the table-row rate is stated"})
    out.append(e)
    h.close()
    return out


def few_chains_regime(device, theta=THETA):
    """Latency of ONE chain when the card is almost empty (64 chains: an annealing run of the reference is one chain and its
    Theta ladder is sequential): sweeps/s per chain with one proposal per step (free_stage_kernel) and with two
    (free_stage_spec_kernel), at the headline shape and at the reference's own input shape."""
    from sac_amd.free import FreeProblem
    from sac_amd.synth import synthetic_timedata
    out = []
    for n_tau, n_omega in ((N_TAU, N_OMEGA), (22, 1000)):
        td = synthetic_timedata(n_tau=n_tau, seed=SEED)
        prob = FreeProblem(td, n_omega, 0.0, OMEGA_M, D_OMEGA, par=1, kernel_type="bosonic")
        for pps in (1, 2):
            h = prob.make_handle(n_chains=64, seed=SEED, device=device, device_tables=True)
            h.set_chains_per_wave(1)
            h.set_proposals_per_step(pps)
            h.adjust_windows(200, theta)
            h.sample(100, theta)
            h.sample(200, theta)
            ms = h.stats().last_kernel_ms
            out.append({"workload": f"sac/free bosonic, N_tau={n_tau}, {n_omega} deltas, {prob.n_grid}-column table, 64 chains",
                        "proposals_per_step": h.proposals_per_step(), "sweeps_per_s_per_chain": 200 / (ms / 1e3)})
            h.close()
    return out


# ------------------------------------------------------------------------------------------------

def self_launch(args, argv):
    """`python bench.py --gpus N` as typed: N fresh rank processes under torch.distributed.run, started before this
    process has touched torch or the GPU; rank 0's JSON line is relayed.  (SAC_BENCH_REHEARSE=1: the ranks share the
    GPUs that exist and merge over gloo -- development only.)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    sys.exit(p.returncode if p.returncode else (0 if line else 1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
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
    ap.add_argument("--no-hbm-regime", action="store_true", help="skip the extra HBM-resident runs")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the CPU baseline (0 = the box's core share)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SAC_BENCH_REHEARSE=1 (development only, never a reported number): ranks share the GPUs that exist and merge
    # over gloo, to walk the N>1 code path on a one-GPU box; the driver's runs use RCCL, one GPU per rank.
    rehearse = os.environ.get("SAC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # SAC_BENCH_FORCE_DIST=1 under torch.distributed.run with ONE rank: the N>1 code path (RCCL communicator on the GPU,
    # the per-stage all-reduces, barrier, max-over-ranks timing) on a one-GPU box (tests/test_gpu_rccl_one_rank.py)
    dist_on = world > 1 or (os.environ.get("SAC_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
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
    # RCCL ranks: the per-stage merge runs BEHIND the C ABI (sac_comm_init: one fused ncclAllReduce(sum) over
    # [histogram | sum of Gbar | scalar sums | chain count] + one ncclAllReduce(min) at the end of every stage call, on the
    # handle's stream); torch only carries the 128-byte unique id to the ranks.  The rehearsal (ranks sharing a GPU) cannot
    # open an RCCL communicator and keeps the host-side merge below.
    abi_merge = dist_on and not rehearse
    abi_error = None
    if abi_merge:
        from sac_amd.dist import Comm
        try:
            Comm(device=torch.device("cuda", local_rank)).attach(h)
        except Exception as ex:              # e.g. librccl not loadable: the same on every rank
            abi_error = str(ex)
        # all ranks take the same path: if the communicator did not come up everywhere, the per-stage merge stays with
        # torch.distributed (RCCL all-reduces on device buffers, below) and the output line says so
        ok = torch.tensor([0.0 if abi_error else 1.0], dtype=torch.float64, device="cuda")
        all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.cpu()[0]) < 0.5:
            abi_merge = False
            abi_error = abi_error or "sac_comm_init failed on another rank"
    if abi_merge:
        h.adjust_windows(100, theta)         # one more untimed tuning round: from here on h.stats() is the merged view
                                             # (counters summed over ALL ranks' chains), before and after the timed region

    def step():
        h.sample(args.sweeps_per_step, theta)
        if dist_on and not abi_merge:
            # per-stage merge (the only collective on the path): histogram + chi^2 statistics
            h.histogram_sum_to_device(hist_dev.data_ptr())
            st = h.stats()
            stat_dev.copy_(torch.tensor([st.sampled_chi2, *st.accept_rates, 1.0], dtype=torch.float64))
            min_dev.fill_(st.chi2_min)
            all_reduce(hist_dev)
            all_reduce(stat_dev)
            all_reduce(min_dev, op=dist.ReduceOp.MIN)

    def fence():
        if dist_on:
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
    if dist_on:
        tmax = local[:1].clone()
        all_reduce(tmax, op=dist.ReduceOp.MAX)
        if not abi_merge:
            all_reduce(local)                # behind the ABI the handle's counters already are the sums over all ranks
        local[0] = tmax[0]
    dt_max, moves, props, sweeps, cols, evald = (float(x) for x in local.cpu())
    # this rank's share of the counters (weak scaling: every rank runs the same number of chains and sweeps)
    per_rank = 1.0 / world if abi_merge else 1.0

    if rank == 0:
        # Roofline of the dominant kernel (free_stage_kernel), this rank, measured with HIP events on the library's
        # stream around every launch of the timed region.  The 51 MB table of this configuration is L2 / Infinity-Cache
        # resident (TCC hit 2/3), so HBM does not bind it: the binding resource is VALU issue on the SIMDs (PMC:
        # SQ_ACTIVE_INST_VALU, profiles/traffic.json).  bound/achieved/peak/frac are stated against that ceiling; the
        # algorithmic gather rate and the counter-based HBM fraction are extra keys.
        cols_r0 = (s1.columns_gathered - s0.columns_gathered) * per_rank
        sweeps_r0 = (s1.sweeps - s0.sweeps) * per_rank
        props_r0 = (s1.proposals - s0.proposals) * per_rank
        n_l = max(launches, 1)
        alg_bytes = cols_r0 * N_TAU * 8 + sweeps_r0 * N_OMEGA * 16   # K columns + histogram RMW
        per_launch_s = kernel_ms / 1e3 / n_l
        alg_gbps = alg_bytes / n_l / per_launch_s / 1e9
        traffic = valu_qc = None
        # PMC counters of the same launch shape, collected in separate rocprofv3 passes ON THIS BUILD (else withheld)
        pmc, pmc_why = load_traffic_json()
        scale = 1.0
        if pmc is not None:
            try:
                w = pmc["workload"]
                if (w["n_tau"], w["n_omega"], w["n_grid"], w["chains"]) == (N_TAU, N_OMEGA, prob.n_grid, args.chains):
                    # per launch of THIS run: the profiled launch scaled by the sweeps a launch ran here
                    scale = sweeps_r0 / args.chains / n_l / w["sweeps_per_launch"]
                    traffic = (2.0 * pmc["fetch_size_kb_median"] + pmc["write_size_kb_median"]) * 1024.0 * scale
                    valu_qc = pmc["valu_active_quad_cycles_median"] * scale
                else:
                    pmc_why = "profiles/traffic.json holds another launch shape"
            except KeyError as ex:
                pmc_why = f"profiles/traffic.json lacks {ex}"
        if valu_qc is not None:
            ach = valu_qc / per_launch_s / 1e9
            roof = {"bound": "valu-issue", "achieved": ach, "peak": VALU_ISSUE_PEAK,
                    "unit": "G wave64 VALU issue quad-cycles/s (peak = 1024 SIMDs x 2.4 GHz / 4)",
                    "frac": ach / VALU_ISSUE_PEAK,
                    "valu_insts_per_proposal": pmc["valu_insts_median"] * scale / (props_r0 / n_l),
                    "salu_insts_per_proposal": pmc["salu_insts_median"] * scale / (props_r0 / n_l)}
        else:
            roof = {"bound": "valu-issue", "achieved": None, "peak": VALU_ISSUE_PEAK,
                    "unit": "G wave64 VALU issue quad-cycles/s", "frac": None, "achieved_null_reason": pmc_why}
        roof.update({"traffic": traffic,
                     "traffic_note": "bytes per launch beyond L2, (2*FETCH_SIZE+WRITE_SIZE)*1024 from profiles/traffic.json",
                     "hbm_frac": None if traffic is None else traffic / per_launch_s / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_GBps": alg_gbps, "algorithmic_over_hbm_peak": alg_gbps / HBM_PEAK_GBS,
                     "gather_ceiling_GBps": GATHER_CEILING_GBS, "algorithmic_over_gather_ceiling": alg_gbps / GATHER_CEILING_GBS,
                     # round 2's run of the same probe read 17.37 TB/s (profiles/r02_gather_probe.txt); round 3's runs read
                     # 16.47 - 16.62 on three settings of the rows in flight: both are kept, the stricter reading is this one
                     "algorithmic_over_gather_ceiling_round2_probe": alg_gbps / 17368.0,
                     "gather_ceiling_source": GATHER_CEILING_SOURCE + " (gather-only probe on the sampler's own column distribution, "
                                              "48.8 MiB table, 8 B per lane as the kernel loads; 16 B per lane / two rows per load: 18.8 TB/s)",
                     # instruction-independent cross-check of the counter file: every gathered column is one wave load
                     "columns_gathered_per_chain_sweep": cols_r0 / max(sweeps_r0, 1),
                     "vmem_rd_per_chain_sweep_pmc": None if (pmc is None or "vmem_rd_insts_median" not in pmc) else
                     pmc["vmem_rd_insts_median"] / (pmc["workload"]["chains"] * pmc["workload"]["sweeps_per_launch"]),
                     "library_build_id": __import__("sac_amd.lib", fromlist=["build_id"]).build_id(),
                     "note": "table is cache resident: the algorithmic gather rate exceeds the HBM peak and is not a "
                             "roofline fraction; see hbm_regime for the HBM-bound shapes",
                     "kernel": "free_stage_kernel<1,false,true,false>", "chains_per_wave": h.chains_per_wave(), "launches": launches,
                     "avg_launch_ms": per_launch_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes / n_l,
                     "pmc_source": None if pmc is None else pmc.get("source")})
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
                       "theta": theta, "parallelism": f"chains sharded over {world} GPU(s), " +
                                                      ("one fused RCCL all-reduce(sum) + one all-reduce(min) per stage behind the C ABI"
                                                       if (abi_merge or not dist_on) else "per-stage RCCL all-reduces through torch.distributed")},
            "sweeps_per_s": sweeps / dt_max, "proposals_per_s": props / dt_max,
            "acceptance": {"evaluated_frac": evald / props, "accepted_moves_per_proposal": moves / props},
            "roofline": roof,
        }
    h_comm_info = h.comm_info() if abi_merge else None
    h.close()
    if rank == 0:
        if world == 1 and not args.no_hbm_regime and not rehearse:
            out["hbm_regime"] = hbm_regime(local_rank, args.chains, theta)
            out["few_chains_regime"] = few_chains_regime(local_rank, theta)
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(prob, win0, wi0, theta, procs=args.cpu_procs or None)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
            out["speedup_note"] = f"one MI355X against {cb['cores']} host cores running the CPU port concurrently"
        if abi_merge:
            out["merge"] = dict(h_comm_info, where="behind the C ABI (sac_comm_init)", n_chains_total=int(s1.n_chains_total),
                                n_chains_expected=world * args.chains)
            if int(s1.n_chains_total) != world * args.chains:
                out["merge"]["warning"] = "the merged chain count is not world x chains: the counters of this line are suspect"
        elif abi_error:
            out["merge"] = {"where": "torch.distributed all-reduces (fallback)", "sac_comm_init_error": abi_error}
        if dist_on and world == 1:
            out["collectives"] = "forced on with one rank (SAC_BENCH_FORCE_DIST=1): RCCL communicator, per-stage all-reduces and barrier executed"
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
