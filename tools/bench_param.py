#!/usr/bin/env python3
"""Secondary throughput measurements (not the headline bench.py line): the peak and edge parameterisations
at BASELINE.json configs[3] / configs[2] shapes on one MI355X.  Prints one JSON line per workload.

    python tools/bench_param.py [peak|edge|both] [--chains N]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sac_amd.synth import synthetic_timedata  # noqa: E402

SEED = 777


def bench_peak(chains, d_omega=1e-5, n_omega=1000, sweeps=100, two_axes=False):
    """sac/peak, N_tau=64, 10^6+1-column table(s) (488 MiB each, beyond L2 + Infinity Cache); bosonic = one
    frequency axis, two_axes = fermionic finiteT with both axes and the +/- weight-transfer moves."""
    from sac_amd.peak import PeakProblem
    if two_axes:
        td = synthetic_timedata(n_tau=64, seed=SEED, kernel_type="finiteT")
        prob = PeakProblem(td, n_omega, 1, 0.5, 10.0, d_omega, kernel_type="finiteT", approx_ratio=2.5)
    else:
        td = synthetic_timedata(n_tau=64, seed=SEED)
        prob = PeakProblem(td, n_omega, 1, 0.5, 10.0, d_omega, kernel_type="bosonic")
    h = prob.make_handle(n_chains=chains, seed=SEED)
    h.adjust_windows(100, 1.0)
    h.sample(20, 1.0)
    s0 = h.peak_stats()
    t0 = time.perf_counter()
    h.sample(sweeps, 1.0)
    s1 = h.peak_stats()
    dt = time.perf_counter() - t0
    cols = s1.columns_gathered - s0.columns_gathered
    kms, nl = s1.last_kernel_ms, s1.last_kernel_launches
    return {"workload": f"sac/peak {'finiteT two axes' if two_axes else 'bosonic'}, N_tau=64, {n_omega}+1 deltas, {prob.n_grid}-column table ({prob.n_grid * 512 / 2**20:.0f} MiB), "
                        f"{chains} chains, Theta=1", "chain_sweeps_per_s": chains * sweeps / dt,
            "proposals_per_s": (s1.proposals - s0.proposals) / dt, "accepted_moves_per_s": (s1.accepted_moves - s0.accepted_moves) / dt,
            "kernel": f"peak_stage_kernel<1,false,{'true' if two_axes else 'false'}>", "launches": nl, "avg_launch_ms": kms / max(nl, 1),
            "algorithmic_GBps": cols * 64 * 8 / (kms / 1e3) / 1e9, "hbm_peak_GBps": 8000.0,
            "frac": cols * 64 * 8 / (kms / 1e3) / 1e9 / 8000.0, "wall_s": dt}


def bench_free2(chains, d_omega=1e-4, n_omega=2000, sweeps=100):
    """sac/free with amplitude updates (par = 2: A_w_move, double_A_move, renormalisation), BASELINE configs[1] shape."""
    from sac_amd.free import FreeProblem
    td = synthetic_timedata(n_tau=64, seed=SEED)
    prob = FreeProblem(td, n_omega, 0.0, 10.0, d_omega, par=2, kernel_type="bosonic")
    h = prob.make_handle(n_chains=chains, seed=SEED)
    h.adjust_windows(100, 1.0)
    h.sample(20, 1.0)
    s0 = h.stats()
    t0 = time.perf_counter()
    h.sample(sweeps, 1.0)
    s1 = h.stats()
    dt = time.perf_counter() - t0
    cols = s1.columns_gathered - s0.columns_gathered
    kms, nl = s1.last_kernel_ms, s1.last_kernel_launches
    return {"workload": f"sac/free par=2 bosonic, N_tau=64, {n_omega} deltas, {prob.n_grid}-column table, {chains} chains, Theta=1",
            "chain_sweeps_per_s": chains * sweeps / dt, "proposals_per_s": (s1.proposals - s0.proposals) / dt,
            "accepted_moves_per_s": (s1.accepted_moves - s0.accepted_moves) / dt,
            "kernel": "free_stage_kernel<1,false,false,true>", "launches": nl, "avg_launch_ms": kms / max(nl, 1),
            "algorithmic_GBps": cols * 64 * 8 / (kms / 1e3) / 1e9, "wall_s": dt}


def bench_free_hbm(chains, d_omega=1e-5, n_omega=2000, sweeps=100):
    """sac/free par=1 at bench.py's hbm_regime shapes: N_tau=64, 2000 deltas, 10^6+1 (d_omega=1e-5) or 4*10^6+1
    (d_omega=2.5e-6) column table built on the device; same stage sequence as bench.py's hbm_regime (for the PMC passes)."""
    from sac_amd.free import FreeProblem
    td = synthetic_timedata(n_tau=64, seed=12345)
    prob = FreeProblem(td, n_omega, 0.0, 10.0, d_omega, par=1, kernel_type="bosonic")
    h = prob.make_handle(n_chains=chains, seed=12345, device_tables=True)
    h.adjust_windows(200, 1.0)
    h.sample(sweeps, 1.0)
    s0 = h.stats()
    kms, nl = 0.0, 0
    for _ in range(2):
        h.sample(sweeps, 1.0)
        s1 = h.stats()
        kms += s1.last_kernel_ms
        nl += s1.last_kernel_launches
    cols = s1.columns_gathered - s0.columns_gathered
    return {"workload": f"sac/free bosonic, N_tau=64, {n_omega} deltas, {prob.n_grid}-column table ({prob.n_grid * 512 / 2**20:.0f} MiB), {chains} chains, Theta=1",
            "chain_sweeps_per_s": (s1.sweeps - s0.sweeps) / (kms / 1e3), "accepted_moves_per_s": (s1.accepted_moves - s0.accepted_moves) / (kms / 1e3),
            "kernel": "free_stage_kernel<1,false,true,false>", "launches": nl, "avg_launch_ms": kms / max(nl, 1),
            "algorithmic_bytes_per_launch": cols * 64 * 8 / max(nl, 1), "algorithmic_GBps": cols * 64 * 8 / (kms / 1e3) / 1e9}


def bench_peak_hbm(chains, sweeps=100, d_omega=1e-5):
    """bench.py's hbm_regime peak entry (BASELINE configs[3]: bosonic, 1000+1 deltas, 10^6+1-column table built on the device),
    same seed and stage sequence, for the PMC passes."""
    from sac_amd.peak import PeakProblem
    td = synthetic_timedata(n_tau=64, seed=12345)
    prob = PeakProblem(td, 1000, 1, 0.5, 10.0, d_omega, kernel_type="bosonic")
    h = prob.make_handle(n_chains=chains, seed=12345, device_tables=True)
    h.adjust_windows(200, 1.0)
    h.sample(sweeps, 1.0)
    s0 = h.peak_stats()
    kms, nl = 0.0, 0
    for _ in range(2):
        h.sample(sweeps, 1.0)
        s1 = h.peak_stats()
        kms += s1.last_kernel_ms
        nl += s1.last_kernel_launches
    cols = s1.columns_gathered - s0.columns_gathered
    return {"workload": f"sac/peak bosonic, N_tau=64, 1000+1 deltas, {prob.n_grid}-column table, {chains} chains, Theta=1 (bench.py hbm_regime)",
            "chain_sweeps_per_s": (s1.sweeps - s0.sweeps) / (kms / 1e3), "launches": nl, "avg_launch_ms": kms / max(nl, 1),
            "algorithmic_bytes_per_launch": cols * 64 * 8 / max(nl, 1), "algorithmic_GBps": cols * 64 * 8 / (kms / 1e3) / 1e9}


def bench_free_packed(chains=8192, n_tau=22, n_omega=1000, sweeps=100):
    """sac/free par=1 at the reference's own input shape (N_tau=22, 1000 deltas, 100001-column table), two chains per
    wavefront (free_stage_packed_kernel), 8192 chains."""
    from sac_amd.free import FreeProblem
    td = synthetic_timedata(n_tau=n_tau, seed=12345)
    prob = FreeProblem(td, n_omega, 0.0, 10.0, 1e-4, par=1, kernel_type="bosonic")
    h = prob.make_handle(n_chains=chains, seed=12345, device_tables=True)
    h.adjust_windows(200, 1.0)
    h.sample(sweeps, 1.0)
    s0 = h.stats()
    kms, nl = 0.0, 0
    for _ in range(2):
        h.sample(sweeps, 1.0)
        s1 = h.stats()
        kms += s1.last_kernel_ms
        nl += s1.last_kernel_launches
    return {"workload": f"sac/free bosonic, N_tau={n_tau}, {n_omega} deltas, {prob.n_grid}-column table, {chains} chains, {h.chains_per_wave()} chains per wave",
            "chain_sweeps_per_s": (s1.sweeps - s0.sweeps) / (kms / 1e3), "accepted_moves_per_s": (s1.accepted_moves - s0.accepted_moves) / (kms / 1e3),
            "proposals_per_s": (s1.proposals - s0.proposals) / (kms / 1e3), "launches": nl, "avg_launch_ms": kms / max(nl, 1)}


def bench_tables(d_omega=1e-5):
    """Row f1: the 10^6+1-column K table of BASELINE configs[3]'s regime, built on the host (numpy, as the reference's
    initialize! does) and uploaded, against built on the device (exp + fp64 MFMA basis change)."""
    import torch
    from sac_amd.free import FreeProblem
    td = synthetic_timedata(n_tau=64, seed=SEED)
    prob = FreeProblem(td, 2000, 0.0, 10.0, d_omega, par=1, kernel_type="bosonic")
    t0 = time.perf_counter()
    K = prob.K
    t_host = time.perf_counter() - t0
    h = prob.make_handle(n_chains=64, seed=SEED)
    t0 = time.perf_counter()
    h.upload_problem(K, td.G, td.sigma_inv)
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    h.build_table(0, "bosonic", td.tau, td.beta, td.cov, prob.wi_0, d_omega)      # warm-up (module load)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        h.build_table(0, "bosonic", td.tau, td.beta, td.cov, prob.wi_0, d_omega)
        ts.append(time.perf_counter() - t0)
    t_dev = min(ts)
    dev = h.get_table(0)
    err = float((np.abs(dev - K) / (np.abs(K).max(axis=1, keepdims=True))).max())
    out_bytes = prob.n_grid * 64 * 8
    flops = 2.0 * prob.n_grid * 64 * 64
    return {"workload": f"K table {prob.n_grid} x 64 (bosonic), host numpy + upload vs device build", "host_build_s": t_host,
            "upload_s": t_up, "device_build_s": t_dev, "device_write_GBps": out_bytes / t_dev / 1e9,
            "mfma_TFLOPs": flops / t_dev / 1e12, "max_rel_diff_vs_host": err}


def bench_edge(chains, d_omega=1e-5, n_e=80, steps=40, bins=4, n_tau=64):
    """sac/edge single_edge, N_tau=64, N_e=80, three 10^6+1-column tables interleaved (1.5 GB)."""
    from sac_amd.edge import EdgeProblem
    td = synthetic_timedata(n_tau=n_tau, seed=SEED)
    prob = EdgeProblem(td, "single_edge", "bosonic", n_e, 0, 0.5, 0.0, 1.0, 0.0, 10.0, d_omega)
    h = prob.make_handle(n_chains=chains, seed=SEED)
    h.run_bins(20, 2, 1.0)
    r0 = h.results(-1)
    t0 = time.perf_counter()
    h.run_bins(steps, bins, 1.0)
    r1 = h.results(-1)
    dt = time.perf_counter() - t0
    look = r1["lookups"] - r0["lookups"]
    ms = r1["last_stage_ms"]
    bytes_alg = look * 5 * n_tau * 8         # 3 table columns + K_curr read + K_new write per lookup
    bytes_tab = look * 3 * n_tau * 8         # the table part alone: what comes from HBM (K_curr / K_new are L2 resident)
    return {"workload": f"sac/edge single_edge bosonic, N_tau={n_tau}, N_e={n_e}, 3 x {prob.kernel.n_grid}-column tables "
                        f"({3 * prob.kernel.n_grid * 8 * ((n_tau + 7) // 8 * 8) / 2**20:.0f} MiB), {chains} chains, Theta=1",
            "chain_sweeps_per_s": chains * steps * bins / dt, "proposals_per_s": (r1["proposals"] - r0["proposals"]) / dt,
            "lookups_per_s": look / dt, "accepted_per_s": (r1["accepted"] - r0["accepted"]) / dt,
            "kernel": f"edge_stage_kernel<{1 if n_tau <= 64 else 2},false>", "stage_ms": ms, "algorithmic_GBps": bytes_alg / (ms / 1e3) / 1e9,
            "table_GBps": bytes_tab / (ms / 1e3) / 1e9, "hbm_peak_GBps": 8000.0,
            # HBM fraction on the table bytes only (round 1 counted all five columns and overstated it: 0.93 against 0.74 by counters)
            "frac": bytes_tab / (ms / 1e3) / 1e9 / 8000.0, "frac_all_columns": bytes_alg / (ms / 1e3) / 1e9 / 8000.0,
            "violations": r1["violations"], "wall_s": dt}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="both")
    ap.add_argument("--chains", type=int, default=4096)
    ap.add_argument("--d-omega", type=float, default=1e-5)
    ap.add_argument("--n-tau", type=int, default=64)
    a = ap.parse_args()
    if a.which in ("peak", "both"):
        print(json.dumps(bench_peak(a.chains, d_omega=a.d_omega)), flush=True)
    if a.which in ("peak2",):
        print(json.dumps(bench_peak(a.chains, d_omega=a.d_omega, two_axes=True)), flush=True)
    if a.which in ("tables",):
        print(json.dumps(bench_tables(a.d_omega)), flush=True)
    if a.which in ("free_hbm",):
        print(json.dumps(bench_free_hbm(a.chains, d_omega=a.d_omega)), flush=True)
    if a.which in ("peak_hbm",):
        print(json.dumps(bench_peak_hbm(a.chains, d_omega=a.d_omega)), flush=True)
    if a.which in ("free_packed",):
        print(json.dumps(bench_free_packed(2 * a.chains)), flush=True)
    if a.which in ("free2",):
        print(json.dumps(bench_free2(a.chains)), flush=True)
    if a.which in ("edge", "both"):
        print(json.dumps(bench_edge(a.chains, d_omega=a.d_omega, n_tau=a.n_tau)), flush=True)
