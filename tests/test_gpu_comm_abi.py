"""The multi-GPU merge BEHIND the C ABI (sac_comm_init / sac_edge_comm_init, include/sac_hip.h) on the one-GPU box: an RCCL
communicator of ONE rank, created and used through the C ABI alone -- no torch.distributed anywhere in this file.  What is
checked: the communicator comes up on the GPU, every stage call issues exactly one all-reduce(sum) + one all-reduce(min), and
a handle with a communicator reports what the same handle without one reports (free and peak: bit for bit -- the device pack
kernels sum the chains in the host's order; edge: to 1e-12, its host path divides before it adds).  The N > 1 transport is the
driver's 8-GPU run; the partition logic is tests/test_dist_gloo.py."""
import numpy as np
import pytest

from sac_amd.lib import comm_unique_id
from tests import helpers

pytestmark = pytest.mark.gpu


def _stats_dict(st, names):
    out = {}
    for n in names:
        v = getattr(st, n)
        out[n] = np.array(v[:]) if hasattr(v, "__len__") else v
    return out


def test_free_merge_through_the_abi_one_rank():
    prob = helpers.problem_golden_bosonic(n_omega=120, omega_m=4.0, d_omega=1e-3)
    a = prob.make_handle(n_chains=12, seed=5, hist_copies=4)
    b = prob.make_handle(n_chains=12, seed=5, hist_copies=4)
    b.comm_init(comm_unique_id(), 0, 1)
    assert b.comm_info() == dict(rank=0, world=1, merges=0, allreduce_calls=0)
    names = ("sampled_chi2", "chi2_min", "chi2_min_mean", "accept_rates", "windows", "proposals", "evaluated", "accepted",
             "accepted_moves", "sweeps", "columns_gathered", "n_chains_total")
    stages = 0
    for kind, steps, theta in (("adjust", 40, 2.0), ("sample", 30, 2.0), ("adjust", 20, 1.0), ("sample", 25, 1.0)):
        for h in (a, b):
            (h.adjust_windows if kind == "adjust" else h.sample)(steps, theta)
        stages += 1
        sa, sb = _stats_dict(a.stats(), names), _stats_dict(b.stats(), names)
        for n in names:
            np.testing.assert_array_equal(sb[n], sa[n], err_msg=f"{kind} {n}")
        assert sb["n_chains_total"] == 12
        info = b.comm_info()
        assert info["merges"] == stages and info["allreduce_calls"] == 2 * stages      # one sum + one min per stage
        if kind == "sample":
            np.testing.assert_array_equal(b.histogram(), a.histogram())
            g = np.mean([a.get_gbar(c)[0] for c in range(12)], axis=0)
            np.testing.assert_allclose(a.gbar_mean(), g, rtol=1e-14)
            np.testing.assert_array_equal(b.gbar_mean(), a.gbar_mean())
    # chi2_min_mean is the mean of the chains' own minima, chi2_min their minimum
    own = a.chain_chi2()[1]
    assert a.stats().chi2_min == own.min() and a.stats().chi2_min_mean == pytest.approx(own.mean(), rel=1e-14)
    # host writes drop the merged view: the getters answer for the local chains until the next stage
    b.set_state(windows=np.full(4, 3.0))
    assert b.stats().windows[0] == 3.0
    a.close()
    b.close()


def test_peak_merge_through_the_abi_one_rank():
    pk = helpers.problem_peak(kind="golden", kernel_type="finiteT", approx_ratio=1.7)
    a = pk.make_handle(n_chains=6, seed=4)
    b = pk.make_handle(n_chains=6, seed=4)
    b.comm_init(comm_unique_id(), 0, 1)
    names = ("sampled_chi2", "chi2_min", "chi2_min_mean", "accept_rates", "windows", "bounds", "weights", "proposals",
             "evaluated", "accepted", "accepted_moves", "sweeps", "columns_gathered", "n_chains_total")
    for h in (a, b):
        h.adjust_windows(20, 5.0)
        h.sample(15, 5.0)
    sa, sb = _stats_dict(a.peak_stats(), names), _stats_dict(b.peak_stats(), names)
    for n in names:
        np.testing.assert_array_equal(sb[n], sa[n], err_msg=n)
    np.testing.assert_array_equal(np.array([list(r) for r in b.peak_stats().edge_res]), np.array([list(r) for r in a.peak_stats().edge_res]))
    for w in range(4):
        np.testing.assert_array_equal(b.histogram_peak(w), a.histogram_peak(w))
    # the binned final sampling (peak:1370-1381): accumulators zeroed, bins accumulate, edge_res sums over bins
    for h in (a, b):
        h.zero_accumulators()
        h.sample_bin(10, 5.0)
        h.sample_bin(10, 5.0)
    ea, eb = np.array([list(r) for r in a.peak_stats().edge_res]), np.array([list(r) for r in b.peak_stats().edge_res])
    np.testing.assert_array_equal(eb, ea)
    assert ea[1, 0] != 0.0
    for w in range(4):
        np.testing.assert_array_equal(b.histogram_peak(w), a.histogram_peak(w))
    info = b.comm_info()
    assert info["merges"] == 4 and info["allreduce_calls"] == 8
    a.close()
    b.close()


def test_edge_merge_through_the_abi_one_rank():
    from tests.test_gpu_edge import make_handle
    from tests.test_oracle_edge import make_chain
    tin, tab, init, _ = make_chain("t2_symm", 2e-3, 24, seed=6)
    a = make_handle(tin, tab, init, 5, 6)
    b = make_handle(tin, tab, init, 5, 6)
    b.comm_init(comm_unique_id(), 0, 1)
    for h in (a, b):
        h.run_bins(12, 3, 5.0)
        h.run_bins(10, 2, 4.0)
    ra, rb = a.results(-1), b.results(-1)
    for k, v in ra.items():
        if k == "last_stage_ms":
            continue
        np.testing.assert_allclose(rb[k], v, rtol=1e-12, atol=1e-300, err_msg=k)
    assert rb["n_chains_total"] == 5 and rb["chi2_min"] == ra["chi2_min"]
    own = np.array([a.results(c)["chi2_min"] for c in range(5)])
    assert ra["chi2_min"] == own.min() and ra["chi2_min_mean"] == pytest.approx(own.mean(), rel=1e-14)
    for which in (0, 1):
        np.testing.assert_allclose(b.histogram(which, 0), a.histogram(which, 0), rtol=1e-12)
    for x, y in zip(b.avg(-1), a.avg(-1)):
        np.testing.assert_allclose(x, y, rtol=1e-12)
    for m in (0, 1):
        np.testing.assert_allclose(b.accept_rates(m, 0), a.accept_rates(m, 0), rtol=1e-12)
    np.testing.assert_allclose(b.gbar_mean(), a.gbar_mean(), rtol=1e-13)
    np.testing.assert_allclose(a.gbar_mean(), np.mean([a.get_gbar(c)[0] for c in range(5)], axis=0), rtol=1e-13)
    # the per-delta windows / cluster sizes write_res prints: chain means over ALL chains, one call
    states = [a.get_state(c) for c in range(5)]
    dw, nm, ext = a.move_means()
    np.testing.assert_allclose(dw, np.mean([s["dw"] for s in states], axis=0), rtol=1e-13)
    np.testing.assert_allclose(nm, np.mean([s["nm"] for s in states], axis=0), rtol=1e-13)
    for x, y in zip(b.move_means(), a.move_means()):
        np.testing.assert_allclose(x, y, rtol=1e-13)
    info = b.comm_info()
    assert info["merges"] == 2 and info["allreduce_calls"] == 4
    a.close()
    b.close()


def test_edge_move_means_with_continuum():
    """A_c > 0: the continuum's extent (lowest / highest continuum frequency, edge:1454-1460) as the mean over all chains."""
    from tests.test_gpu_edge import make_handle
    from tests.test_oracle_edge import make_chain
    tin, tab, init, _ = make_chain("t1_single", 2e-3, 20, n_c=10, A_c=0.3, seed=8)
    a = make_handle(tin, tab, init, 4, 8)
    b = make_handle(tin, tab, init, 4, 8)
    b.comm_init(comm_unique_id(), 0, 1)
    for h in (a, b):
        h.run_bins(10, 2, 5.0)
    states = [a.get_state(c) for c in range(4)]
    NE = init.n_e
    want = np.array([np.mean([s["w"][NE + 1:, 0].min() for s in states]), np.mean([s["w"][NE + 1:, 0].max() for s in states])])
    np.testing.assert_allclose(a.move_means()[2], want, rtol=1e-13)
    np.testing.assert_allclose(b.move_means()[2], want, rtol=1e-13)
    a.close()
    b.close()
