"""N>1 path on CPU: world_size-2 gloo.  Chains are sharded over ranks, each rank runs its chains
(here through the CPU oracle -- the merge logic under test is backend independent), and the
per-stage merge (sac_amd/dist.py) must give every rank the statistics and histogram of the
single-process run over all chains."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers

N_CHAINS, STEPS, THETA, SEED = 5, 6, 2.0, 77


def _run_chains(prob, first, count):
    specs, chi2s, mins, ars, wins = [], [], [], [], []
    for c in range(first, first + count):
        o = helpers.oracle_chain(prob, SEED, c)
        o.adjust_windows(10, THETA)
        o.sample(STEPS, THETA)
        specs.append(o.spec() * STEPS)
        chi2s.append(o.sampled_chi2())
        mins.append(o.get_gbar()[2])
        ars.append(o.accept_rates())
        wins.append(o.get_state()[2])
    return specs, chi2s, mins, ars, wins


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sac_amd.dist import Comm, shard_chains
    prob = helpers.problem_golden_bosonic(n_omega=40, omega_m=3.0, d_omega=2e-3)
    first, count = shard_chains(N_CHAINS, rank, world)
    specs, chi2s, mins, ars, wins = _run_chains(prob, first, count)
    comm = Comm()
    merged = comm.merge_scalars(count, float(np.mean(chi2s)), float(np.min(mins)),
                                list(np.mean(ars, axis=0)), list(np.mean(wins, axis=0)))
    hist = comm.merge_histogram_sum(torch.from_numpy(np.sum(specs, axis=0)), float(STEPS * count))
    out[rank] = dict(first=first, count=count, chi2=merged.sampled_chi2, cmin=merged.chi2_min,
                     ar=merged.accept_rates, win=merged.windows, n=merged.n_chains,
                     hist=hist.numpy().copy())
    dist.destroy_process_group()


def test_two_rank_merge_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert (r0["first"], r0["count"], r1["first"], r1["count"]) == (0, 3, 3, 2)
    prob = helpers.problem_golden_bosonic(n_omega=40, omega_m=3.0, d_omega=2e-3)
    specs, chi2s, mins, ars, wins = _run_chains(prob, 0, N_CHAINS)
    for r in (r0, r1):                       # every rank holds the same merged result
        assert r["n"] == N_CHAINS
        assert r["cmin"] == min(mins)
        np.testing.assert_allclose(r["chi2"], np.mean(chi2s), rtol=1e-14)
        np.testing.assert_allclose(r["ar"], np.mean(ars, axis=0), rtol=1e-13)
        np.testing.assert_allclose(r["win"], np.mean(wins, axis=0), rtol=1e-13)
        np.testing.assert_allclose(r["hist"], np.sum(specs, axis=0) / (STEPS * N_CHAINS), rtol=1e-13, atol=0)
    assert np.array_equal(r0["hist"], r1["hist"]) and r0["chi2"] == r1["chi2"]


def test_shard_chains_partitions():
    from sac_amd.dist import shard_chains
    for n, w in ((4096, 8), (10, 3), (5, 8), (7, 1)):
        spans = [shard_chains(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1


def _named_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sac_amd.dist import Comm
    rng = np.random.default_rng(100 + rank)
    n = 3 + 2 * rank                                  # unequal shards: 3 and 5 chains
    per_chain = rng.normal(size=(n, 3, 2))
    m = Comm().merge_named(n, {"res": per_chain.mean(axis=0), "chi2": float(per_chain[:, 0, 0].mean())},
                           mins={"chi2_min": float(per_chain.min())}, sums={"violations": rank + 1})
    out[rank] = dict(m, per_chain=per_chain)
    dist.destroy_process_group()


def test_merge_named_two_ranks():
    """Generic merge of the peak / edge drivers: chain-weighted means, minima and sums."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_named_worker, args=(2, port, out), nprocs=2, join=True)
    allc = np.concatenate([out[0]["per_chain"], out[1]["per_chain"]])
    for r in (out[0], out[1]):
        assert r["n_chains"] == 8
        np.testing.assert_allclose(r["res"], allc.mean(axis=0), rtol=1e-13, atol=1e-15)
        assert r["chi2"] == pytest.approx(allc[:, 0, 0].mean(), rel=1e-13)
        assert r["chi2_min"] == allc.min()
        assert r["violations"] == 3


# ---- cross-validation units partitioned over ranks (BASELINE configs[4]) -----------------------------------

def _cv_worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import shutil
    from sac_amd.cross_val import cv_curves, cv_units, run_cv_partition
    cv = os.path.join(os.path.dirname(__file__), "golden", "cv")
    ran = []

    def runner(unit, **kw):
        # stands in for free_cv on a GPU: "produces" the reference's committed GSAC.csv of that unit
        prm, rep, ab = unit
        folder = os.path.join(tmp, "%03i%s" % (rep, ab), prm)
        os.makedirs(folder, exist_ok=True)
        shutil.copyfile("%s/GSAC_%s_%03i%s.csv" % (cv, prm, rep, ab), folder + "/GSAC.csv")
        ran.append(unit)
        return folder

    units = cv_units([1, 2], params=("free",))
    folders = run_cv_partition(units, rank, world, runner=runner)
    curves = cv_curves(folders, cv, params=("free",)) if rank == 0 else {}
    out[rank] = dict(ran=ran, n=len(folders), curves={k: (v[0].tolist(), v[1].tolist()) for k, v in curves.items()})
    dist.destroy_process_group()


def test_cv_units_round_robin_over_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cv_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    # 2 replicas x 2 halves x 1 parameterisation = 4 units: disjoint, covering, alternating
    assert sorted(r0["ran"] + r1["ran"]) == sorted([("free", 1, "a"), ("free", 1, "b"), ("free", 2, "a"), ("free", 2, "b")])
    assert len(r0["ran"]) == len(r1["ran"]) == 2 and not set(map(tuple, r0["ran"])) & set(map(tuple, r1["ran"]))
    assert r0["n"] == r1["n"] == 4                              # every rank learns every unit's folder
    exp = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "cv", "free_v_expected.csv"), delimiter=",", ndmin=2)
    for rep in (1, 2):                                          # root scores them: the reference's free_v.csv
        th, v = r0["curves"][("free", rep)]
        n = min(len(v), exp.shape[1])
        np.testing.assert_allclose(th[:n], exp[0][:n], rtol=1e-9)
        np.testing.assert_allclose(v[:n], exp[rep][:n], rtol=1e-9)


def _scan_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sac_amd.scan import best_point, run_scan_partition, scan_units, scan_values
    ran = []

    def runner(unit, **kw):
        # stands in for a complete sac_edge run on this rank's GPU at that parameter point
        prog, par = unit
        par = dict(par)
        ran.append(par["p"])
        return dict(folder=f"out/p_{par['p']:.3f}", chi2_avg=(par["p"] - 0.4) ** 2 + 1.0, theta_opt=0.1)

    units = scan_units("edge", "p", scan_values(0.3, 0.1, 5), A_c=0.0, A_r=0.5)     # qsub_p_scan.sh with N_p = 5
    res = run_scan_partition(units, rank, world, runner=runner)
    best = best_point(res)
    out[rank] = dict(ran=ran, n=len(res), best=dict(best[0][1])["p"], keys=[dict(u[1])["p"] for u in units])
    dist.destroy_process_group()


def test_scan_points_round_robin_over_two_ranks():
    """Parameter-scan points (qsub_files/edge/qsub_p_scan.sh:13-19: p = p_0 + delta_p (task - 1)) as whole-GPU units."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_scan_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["keys"] == [0.3, 0.4, 0.5, 0.6, 0.7]
    assert r0["ran"] == [0.3, 0.5, 0.7] and r1["ran"] == [0.4, 0.6]            # disjoint, covering, alternating
    assert r0["n"] == r1["n"] == 5 and r0["best"] == r1["best"] == 0.4         # every rank learns every point


def _late_worker(rank, world, port, out):
    import datetime
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # the default group's collectives give up after 2 s; the partition's gather must not depend on that
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=2))
    from sac_amd.scan import run_scan_partition, scan_units, scan_values

    def runner(unit, **kw):
        if rank == 1:
            time.sleep(6.0)          # this rank's point takes far longer than the default group's timeout
        return dict(chi2_avg=1.0 + rank, rank=rank)

    units = scan_units("edge", "p", scan_values("0.5", "0.025", 2), A_c=0.0, A_r=0.5)
    t0 = time.time()
    res = run_scan_partition(units, rank, world, runner=runner)
    out[rank] = dict(n=len(res), ranks=sorted(r["rank"] for r in res.values()), waited=time.time() - t0)
    dist.destroy_process_group()


def test_partition_gather_survives_a_rank_that_finishes_late():
    """Scan points / cross-validation units are whole runs dealt round-robin; the early ranks wait in the final gather for
    the slow one far longer than a collective watchdog allows.  The gather runs on its own gloo group with a long timeout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_late_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0]["n"] == out[1]["n"] == 2 and out[0]["ranks"] == out[1]["ranks"] == [0, 1]
    assert out[0]["waited"] > 4.0                                              # rank 0 really sat in the gather


def test_scan_values_keep_the_operands_decimals():
    """qsub_p_scan.sh: `bc` keeps the operands' scale for + and *: p_0 = 0.5, delta = 0.025 gives .525 (not 0.52), and the
    output folders p_%.3f follow."""
    from sac_amd.scan import scan_values
    assert scan_values("0.5", "0.025", 4) == [0.5, 0.525, 0.55, 0.575]
    assert scan_values(0.5, 0.025, 4) == [0.5, 0.525, 0.55, 0.575]
    assert scan_values("0.3", "0.1", 5) == [0.3, 0.4, 0.5, 0.6, 0.7]           # no binary drift (0.30000000000000004)
    assert ["%.3f" % v for v in scan_values("0.5", "0.025", 2)] == ["0.500", "0.525"]


def test_theta_scan_points_follow_the_reference_ladder():
    from sac_amd.scan import scan_units, theta_scan_points
    pts = theta_scan_points(1.0, 0.1)                                          # edge:1805-1809, f_scan = 1.15
    assert len(pts) == 17 and pts[0] == 1.0 and pts[-1] > 0.1 > pts[-1] / 1.15
    assert np.allclose(np.array(pts[:-1]) / np.array(pts[1:]), 1.15)
    with pytest.raises(ValueError):
        scan_units("peak", "p", [0.1])
    u = scan_units("peak", "A_0", [0.5, 0.6], N_p=1)
    assert [dict(x[1]) for x in u] == [{"A_0": 0.5, "N_p": 1}, {"A_0": 0.6, "N_p": 1}]


def test_units_of_a_rank_side_by_side():
    """side_by_side: a rank's units run in host threads, several at a time; same {unit: result} map, really overlapped."""
    import threading
    import time
    from sac_amd.scan import run_scan_partition, scan_units, scan_values
    units = scan_units("edge", "p", scan_values("0.3", "0.1", 6), A_c=0.0, A_r=0.5)
    live, peak, lock = [0], [0], threading.Lock()

    def runner(unit, **kw):
        with lock:
            live[0] += 1
            peak[0] = max(peak[0], live[0])
        time.sleep(0.3)
        with lock:
            live[0] -= 1
        return dict(chi2_avg=dict(unit[1])["p"], thread=threading.get_ident())

    t0 = time.time()
    seq = run_scan_partition(units, runner=runner)
    t_seq, peak_seq = time.time() - t0, peak[0]
    peak[0] = 0
    t0 = time.time()
    par = run_scan_partition(units, runner=runner, side_by_side=3)
    t_par = time.time() - t0
    assert peak_seq == 1 and peak[0] == 3
    assert list(par) == list(seq) == units and [r["chi2_avg"] for r in par.values()] == [r["chi2_avg"] for r in seq.values()]
    assert t_par < 0.75 * t_seq

    def failing(unit, **kw):
        if dict(unit[1])["p"] > 0.55:
            raise RuntimeError("this point failed")
        return dict(chi2_avg=1.0)
    with pytest.raises(RuntimeError, match="this point failed"):
        run_scan_partition(units, runner=failing, side_by_side=3)
