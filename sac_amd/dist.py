"""Multi-GPU merge for the sampling stages: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

With the "nccl" backend the merge itself lives BEHIND the C ABI (sac_comm_init / sac_edge_comm_init,
include/sac_hip.h): `Comm.attach(handle)` hands the handle an RCCL communicator, every stage call then ends
with one fused ncclAllReduce(sum) + one ncclAllReduce(min) on the handle's stream, and the handle's reduced
getters answer for all ranks -- this module only carries the 128-byte unique id to the ranks.  The torch
collectives below remain for host-side groups (gloo: the CPU tests, several ranks sharing one GPU) and for
what is not a stage statistic.

The path shards: independent chains are block-partitioned over ranks (the Philox key carries the
global chain id, so a chain's trajectory does not depend on the rank count) and no collective
runs inside a stage.  After a stage every rank holds partial sums; ONE fused all-reduce(sum) of
[histogram | chi2 sum | accept rates | windows | chain count] plus one all-reduce(min) for
chi2_min merge them, so every rank evaluates the same Theta-schedule predicates
(free:1087, 1128, 1249-1252).  The reference has no counterpart: its only fan-out is one Julia
process per parameter point / replica with files as the hand-off.
"""
from types import SimpleNamespace

import torch
import torch.distributed as dist


def shard_chains(n_total, rank, world):
    """Block partition: (first global chain id, number of chains) of `rank`."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class Comm:
    """Stage-level merge over a process group."""

    def __init__(self, group=None, device="cpu"):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        # the merge can go behind the C ABI when the ranks own one GPU each and talk RCCL
        self.abi = self.device.type == "cuda" and dist.get_backend(group) == "nccl"

    def attach(self, handle):
        """Puts the per-stage merge of `handle` (SacHandle / SacEdgeHandle) behind the C ABI: rank 0 draws the RCCL
        unique id (sac_comm_unique_id), every rank joins with sac_comm_init.  Collective.  Returns False (and leaves the
        torch path in charge) for host-side groups."""
        if not self.abi:
            return False
        from .lib import comm_unique_id
        box = [comm_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                   group=self.group, device=self.device)
        handle.comm_init(box[0], self.rank, self.world)
        return True

    def merge_scalars(self, n_chains, sampled_chi2, chi2_min, accept_rates, windows, chi2_min_mean=None):
        """Chain-weighted means (min for chi2_min) over all ranks."""
        cmm = chi2_min if chi2_min_mean is None else chi2_min_mean
        buf = torch.tensor([sampled_chi2 * n_chains, *[a * n_chains for a in accept_rates],
                            *[w * n_chains for w in windows], cmm * n_chains, float(n_chains)],
                           dtype=torch.float64, device=self.device)
        mn = torch.tensor([chi2_min], dtype=torch.float64, device=self.device)
        dist.all_reduce(buf, group=self.group)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=self.group)
        buf = buf.cpu()
        n = float(buf[-1])
        return SimpleNamespace(sampled_chi2=float(buf[0]) / n, chi2_min=float(mn.cpu()[0]),
                               accept_rates=[float(x) / n for x in buf[1:7]],
                               windows=[float(x) / n for x in buf[7:11]], chi2_min_mean=float(buf[11]) / n,
                               n_chains=int(n), n_chains_total=int(n))

    def merge_named(self, n_chains, means, mins=None, sums=None):
        """Generic per-stage merge used by the peak and edge drivers: `means` are this rank's
        chain-means (weighted by n_chains across ranks), `mins` take the minimum, `sums` are added.
        Values are scalars or arrays; one all-reduce(sum) + one all-reduce(min).  Returns a dict."""
        import numpy as np
        mins, sums = mins or {}, sums or {}
        parts, layout = [], []
        for kind, d, wgt in (("mean", means, float(n_chains)), ("sum", sums, 1.0)):
            for k, v in d.items():
                a = np.asarray(v, dtype=np.float64)
                layout.append((kind, k, a.shape, a.size))
                parts.append(a.ravel() * wgt)
        parts.append(np.array([float(n_chains)]))
        buf = torch.from_numpy(np.concatenate(parts)).to(self.device)
        dist.all_reduce(buf, group=self.group)
        buf = buf.cpu().numpy()
        n_tot = buf[-1]
        out, pos = {}, 0
        for kind, k, shape, size in layout:
            a = buf[pos:pos + size].reshape(shape)
            out[k] = a / n_tot if kind == "mean" else a
            if shape == ():
                out[k] = float(out[k])
            pos += size
        if mins:
            keys = list(mins)
            mn = torch.tensor([float(mins[k]) for k in keys], dtype=torch.float64, device=self.device)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=self.group)
            for k, v in zip(keys, mn.cpu().numpy()):
                out[k] = float(v)
        out["n_chains"] = int(round(n_tot))
        return out

    def sum_vector(self, v):
        """All-reduce(sum) of a small host vector (e.g. the chain-summed Gbar of the cross-validation dump)."""
        t = torch.as_tensor(v, dtype=torch.float64).to(self.device)
        dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def merge_stats(self, st, n_chains=None):
        """st: sac_stats_t-like (sampled_chi2, chi2_min, accept_rates[6], windows[4])."""
        n = n_chains if n_chains is not None else getattr(st, "n_chains", 1)
        return self.merge_scalars(n, st.sampled_chi2, st.chi2_min, list(st.accept_rates)[:6],
                                  list(st.windows)[:4], getattr(st, "chi2_min_mean", None))

    def merge_histogram_sum(self, hist_sum, divisor):
        """hist_sum: this rank's un-normalised histogram (tensor on self.device); divisor: its
        steps*chains.  Returns the merged, normalised histogram (tensor)."""
        d = torch.tensor([divisor], dtype=torch.float64, device=self.device)
        dist.all_reduce(hist_sum, group=self.group)
        dist.all_reduce(d, group=self.group)
        return hist_sum / d

    def merge_histogram(self, handle):
        """Histogram of a SacHandle merged over ranks (device-side sum, RCCL all-reduce).  The replica sum is
        always written on the handle's own GPU (sac_get_histogram_sum_device takes a DEVICE pointer); with a
        host-side group (gloo) it is staged through host memory for the all-reduce."""
        gpu = torch.device("cuda", int(handle.cfg.device))
        buf = torch.empty(handle.n_grid, dtype=torch.float64, device=gpu)
        divisor = handle.histogram_sum_to_device(buf.data_ptr())
        if self.device != gpu:
            buf = buf.to(self.device)
        return self.merge_histogram_sum(buf, divisor).cpu().numpy()
