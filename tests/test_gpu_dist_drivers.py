"""N>1 path of the free, peak and edge drivers on the GPU box: two processes (gloo for the merge; both use the one
MI355X of the box), chains sharded by chain_offset = rank * n_chains.  Every rank must end a stage with the
statistics of the single-process run over all chains (a chain's trajectory depends only on its global id)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _edge(comm, n_chains, folder):
    from sac_amd.edge import EdgeSAC
    s = EdgeSAC.from_infile(f"{GOLD}/edge/t2_symm/in_edge.in", n_chains=n_chains, seed=31, comm=comm)
    s.G_file, s.output_folder, s.d_omega = f"{GOLD}/edge/t2_symm/t.in", folder, 1e-3
    os.makedirs(folder, exist_ok=True)
    s.write_log("start", "w")
    s.init_outfiles()
    s.initialize()
    s.run_bins(12, 3, 5.0)
    s.write_res(1, 5.0)
    s.write_spec(1, 36, 0)
    r = s.res
    return {k: np.array(r[k]) for k in ("chi2_res", "edge_res", "n0_res", "wn0_res", "amp_ar", "eps0")} | {"chi2_min": r["chi2_min"]}


def _peak(comm, n_chains, folder):
    from sac_amd.peak import PeakSAC
    from sac_amd.tin import read_tin
    td = read_tin(f"{GOLD}/cv/t_001a.in")
    s = PeakSAC(60, 2, 0.6, 6.0, 2e-3, 0.05, 5.0, 1.1, 0.5, 5, 20, 20, "", folder, 0, 0, "finiteT",
                n_chains=n_chains, seed=32, timedata=td, comm=comm)
    os.makedirs(folder, exist_ok=True)
    s.init_outfiles()
    s.initialize()
    s.adjust_windows(20, 5.0)
    s.sample(10, 5.0)
    spec = s.write_spec(1)
    out = dict(sampled_chi2=s.sampled_chi2, chi2_min=s.chi2_min, accept_rates=np.array(s.accept_rates),
               windows=np.array(s.update_windows), weights=np.array(s.weights))
    if spec is not None:
        out["spec"] = spec[0]
    return out


def _free(comm, n_chains, folder):
    """FreeSAC + Comm: window tuning, a sample stage, the merged histogram (Comm.merge_histogram: replica sum on the
    handle's GPU, staged through host for gloo) and write_spec (free:959-1019)."""
    from sac_amd.free import FreeSAC
    from sac_amd.tin import read_tin
    td = read_tin(f"{GOLD}/cv/t_001a.in")
    s = FreeSAC(1, 60, 0.0, 4.0, 2e-3, 0.02, 5.0, 1.1, 1.05, 0.5, 0.5, 5, 20, 20, "", folder, 0, "bosonic",
                n_chains=n_chains, seed=33, timedata=td, comm=comm)
    s.init_outfiles()
    s.initialize()
    s.adjust_windows(20, 5.0)
    s.sample(10, 5.0)
    spec = s.sampled_spec()
    coarse = s.write_spec(1)
    out = dict(sampled_chi2=s.sampled_chi2, chi2_min=s.chi2_min, accept_rates=np.array(s.accept_rates),
               windows=np.array(s.update_windows), hist=spec, gbar=s.mean_gbar())
    if coarse is not None:
        out["spec"] = coarse
    return out


def _worker(rank, world, port, base, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sac_amd.dist import Comm
    comm = Comm()
    out[("free", rank)] = _free(comm, 4, f"{base}/free2")
    out[("edge", rank)] = _edge(comm, 4, f"{base}/edge2")
    out[("peak", rank)] = _peak(comm, 4, f"{base}/peak2")
    dist.destroy_process_group()


def test_two_rank_drivers_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    f1 = _free(None, 8, str(tmp_path / "free1"))
    assert f1["hist"].sum() > 0
    e1 = _edge(None, 8, str(tmp_path / "edge1"))
    p1 = _peak(None, 8, str(tmp_path / "peak1"))
    for rank in (0, 1):
        e2, p2, f2 = out[("edge", rank)], out[("peak", rank)], out[("free", rank)]
        for k, v in f1.items():
            if k in f2:
                np.testing.assert_allclose(f2[k], v, rtol=1e-11, atol=1e-13, err_msg=f"free {k}")
        assert ("spec" in f2) == (rank == 0)
        for k, v in e1.items():
            np.testing.assert_allclose(e2[k], v, rtol=1e-11, atol=1e-13, err_msg=f"edge {k}")
        for k, v in p1.items():
            if k in p2:
                np.testing.assert_allclose(p2[k], v, rtol=1e-11, atol=1e-13, err_msg=f"peak {k}")
    # only rank 0 wrote files, in the single-process layout
    a = open(tmp_path / "edge2" / "anneal.csv").read().split("\n")
    b = open(tmp_path / "edge1" / "anneal.csv").read().split("\n")
    assert len(a) == len(b) == 3 and a[0] == b[0]
    assert os.path.exists(tmp_path / "edge2" / "sw001_1.dat") and os.path.exists(tmp_path / "peak2" / "sw001.csv")
    two, one = (np.loadtxt(tmp_path / d / "sw001.csv", delimiter=",", skiprows=1) for d in ("free2", "free1"))
    np.testing.assert_allclose(two, one, rtol=1e-11, atol=0)         # the merge adds the ranks' sums in another order


def _theta_scan(rank, world, folder, gather=None):
    from sac_amd.edge import EdgeSAC
    from sac_amd.scan import sharded_theta_scan
    s = EdgeSAC.from_infile(f"{GOLD}/edge/hchain_single/in_edge.in", n_chains=16, seed=41)
    s.G_file, s.output_folder, s.d_omega = f"{GOLD}/edge/hchain_single/t.in", folder, 1e-4
    s.anneal_steps, s.sample_steps, s.bins, s.f_anneal = 400, 150, 10, 1.25
    os.makedirs(folder, exist_ok=True)
    s.write_log("start", "w")
    s.init_outfiles()
    s.initialize()
    s.run_bins(5 * s.anneal_steps, s.bins, s.theta_0)
    rows = sharded_theta_scan(s, theta_1=2.0, theta_2=0.8, rank=rank, world=world, gather=gather)
    s.h.close()
    return rows


def _scan_worker(rank, world, port, base, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out[rank] = _theta_scan(rank, world, f"{base}/scan2_rank{rank}")
    dist.destroy_process_group()


def test_theta_scan_points_sharded_over_two_ranks(tmp_path):
    """anneal_and_scan's Theta points (edge:1783-1846) dealt round-robin to two ranks (scan.sharded_theta_scan) against the
    sequential walk of one process: the same ladder, every point sampled by exactly one rank, every rank learns all rows,
    and <chi^2> at each point agrees with the sequential walk within the runs' own error bars (each point is an
    equilibrium sample at its own Theta: that is what makes the sharding legitimate)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_scan_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    seq = _theta_scan(0, 1, str(tmp_path / "scan1"))
    assert out[0] == out[1] and list(out[0]) == list(seq) == list(range(1, len(seq) + 1)) and len(seq) == 7
    for i, row in seq.items():
        two = out[0][i]
        assert two["theta"] == pytest.approx(row["theta"], rel=1e-12)
        err = np.hypot(row["chi2_sigma"], two["chi2_sigma"])
        print(i, row["theta"], row["chi2_avg"], two["chi2_avg"], err)
        assert abs(two["chi2_avg"] - row["chi2_avg"]) < 5 * err + 0.03 * row["chi2_avg"]
        assert abs(two["edge"] - row["edge"]) < 0.01
    # a point's files are written by the rank that sampled it: odd points by rank 0, even ones by rank 1
    for i in seq:
        owner = (i - 1) % 2
        assert os.path.exists(tmp_path / f"scan2_rank{owner}" / f"sw{i:03d}_1.dat")
        assert not os.path.exists(tmp_path / f"scan2_rank{1 - owner}" / f"sw{i:03d}_1.dat")
