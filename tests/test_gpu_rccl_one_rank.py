"""The RCCL code path on the one-GPU box: a process group with backend "nccl" (= RCCL on ROCm) of ONE rank.  The 8-GPU
runs are the driver's; what can be checked here is that everything the N>1 path does on the device executes and is
right -- communicator creation on the GPU, the drivers' per-stage all-reduces through `Comm(device=cuda)` (device
buffers, no host staging), barrier, and bench.py's merged stage -- and that one rank with a communicator reproduces the
run without one exactly."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.test_gpu_dist_drivers import _edge, _free, _peak

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, port, base, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from sac_amd.dist import Comm
    comm = Comm(device=torch.device("cuda", 0))
    t = torch.arange(4, dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    dist.barrier()
    out["allreduce"] = t.cpu().numpy()
    out["free"] = _free(comm, 8, f"{base}/free_c")
    out["edge"] = _edge(comm, 8, f"{base}/edge_c")
    out["peak"] = _peak(comm, 8, f"{base}/peak_c")
    dist.destroy_process_group()


def test_one_rank_rccl_drivers_equal_plain_run(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, str(tmp_path), out), nprocs=1, join=True)
    np.testing.assert_array_equal(out["allreduce"], np.arange(4.0))
    plain = {"free": _free(None, 8, str(tmp_path / "free_p")), "edge": _edge(None, 8, str(tmp_path / "edge_p")),
             "peak": _peak(None, 8, str(tmp_path / "peak_p"))}
    for prog, ref in plain.items():
        got = out[prog]
        for k, v in ref.items():
            assert k in got, f"{prog}: {k} missing with a communicator"
            np.testing.assert_allclose(got[k], v, rtol=1e-12, atol=1e-14, err_msg=f"{prog} {k}")


def test_bench_merged_stage_on_one_rank_rccl():
    """bench.py under torch.distributed.run with one rank and SAC_BENCH_FORCE_DIST=1: the stage merge of the N>1 bench
    (histogram replicas summed on the device, three RCCL all-reduces, barrier + max-over-ranks timing)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SAC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--chains", "512", "--sweeps-per-step", "50", "--no-cpu", "--no-hbm-regime"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and "collectives" in d
