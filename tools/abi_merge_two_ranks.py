#!/usr/bin/env python3
"""Two ranks, ONE GPU: does the merge behind the C ABI (sac_comm_init) sum over ranks?  RCCL normally refuses two ranks on one
device ("Duplicate GPU detected"); when it does, this prints that and exits 0 -- the check then has to wait for the 8-GPU node.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/abi_merge_two_ranks.py"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")                      # only carries the unique id and the check values
from sac_amd.lib import comm_unique_id              # noqa: E402
from tests import helpers                            # noqa: E402

prob = helpers.problem_golden_bosonic(n_omega=120, omega_m=4.0, d_omega=1e-3)
n = 6
h = prob.make_handle(n_chains=n, seed=5, chain_offset=rank * n, hist_copies=2)
ref = prob.make_handle(n_chains=world * n, seed=5, hist_copies=2)      # all chains in one handle, no communicator
box = [comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(box, src=0)
try:
    h.comm_init(box[0], rank, world)
except Exception as ex:
    print(json.dumps({"rank": rank, "comm_init": "refused", "why": str(ex)[:300]}), flush=True)
    dist.destroy_process_group()
    sys.exit(0)
for hh in (h, ref):
    hh.adjust_windows(30, 2.0)
    hh.sample(20, 2.0)
a, b = h.stats(), ref.stats()
out = {"rank": rank, "n_chains_total": a.n_chains_total, "merged": [a.sampled_chi2, a.chi2_min, a.chi2_min_mean, a.proposals, a.accepted_moves, a.sweeps],
       "one_handle": [b.sampled_chi2, b.chi2_min, b.chi2_min_mean, b.proposals, b.accepted_moves, b.sweeps],
       "hist_max_abs_diff": float(np.abs(h.histogram() - ref.histogram()).max()), "comm": h.comm_info()}
out["agree"] = bool(np.allclose(out["merged"], out["one_handle"], rtol=1e-12)) and out["hist_max_abs_diff"] < 1e-12
print(json.dumps(out), flush=True)
h.close()
ref.close()
dist.destroy_process_group()
