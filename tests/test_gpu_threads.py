"""Different handles driven from different host threads at the same time (INTEGRATION.md "Several runs per GPU"): every
handle's chains are, bit for bit, the chains it runs when the handles are driven one after the other -- free, peak and edge.
(The library's host copies are ordered on the handle's own stream, its error state is thread-local, and it keeps no other
mutable process state; tools/side_by_side_check.py is the same statement for complete cross-validation units.)"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
N_HANDLES = 4


def _free_job(seed):
    prob = helpers.problem_golden_bosonic(n_omega=200, omega_m=5.0, d_omega=1e-3)
    h = prob.make_handle(n_chains=6, seed=seed, hist_copies=6, sweeps_per_launch=5)
    for theta in (5.0, 1.0):
        h.adjust_windows(40, theta)
        h.sample(30, theta)
    out = [h.get_state(c)[0] for c in range(6)] + [h.get_gbar(c)[0] for c in range(6)] + [h.histogram()]
    st = h.stats()
    out.append(np.array([st.sampled_chi2, st.chi2_min, st.accepted, st.proposals], dtype=float))
    h.close()
    return out


def _peak_job(seed):
    prob = helpers.problem_peak(kind="golden", n_omega=80, n_p=2)
    h = prob.make_handle(n_chains=5, seed=seed, hist_copies=5, sweeps_per_launch=5)
    for theta in (5.0, 1.0):
        h.adjust_windows(30, theta)
        h.sample(20, theta)
    out = [h.get_state(c)[0] for c in range(5)] + [h.get_state(c)[1] for c in range(5)] + [h.get_gbar(c)[0] for c in range(5)]
    st = h.peak_stats()
    out.append(np.array([st.sampled_chi2, st.chi2_min, st.accepted, st.proposals], dtype=float))
    h.close()
    return out


def _edge_job(seed):
    from tests.test_gpu_edge import make_handle
    from tests.test_oracle_edge import make_chain
    tin, tab, init, _ = make_chain("hchain_single", 2e-3, 30, 0, 0.0, 0.5, seed=seed, chain=0)
    h = make_handle(tin, tab, init, 4, seed, sweeps_per_launch=4)
    for steps, bins, theta in ((12, 3, 10.0), (8, 2, 1.0)):
        h.run_bins(steps, bins, theta)
    out = []
    for c in range(4):
        s = h.get_state(c)
        out += [s["w"], s["A"], s["dw"], s["nm"], h.get_gbar(c)[0], h.results(c)["chi2_res"], h.results(c)["edge_res"]]
    h.close()
    return out


@pytest.mark.parametrize("job", [_free_job, _peak_job, _edge_job], ids=["free", "peak", "edge"])
def test_handles_driven_from_threads_equal_handles_driven_in_turn(job):
    seeds = [101 + 7 * i for i in range(N_HANDLES)]
    in_turn = [job(s) for s in seeds]
    with ThreadPoolExecutor(N_HANDLES) as ex:
        together = list(ex.map(job, seeds))
    for a, b in zip(in_turn, together):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    # and the handles really differ from each other (different seeds)
    assert not np.array_equal(np.asarray(in_turn[0][0]), np.asarray(in_turn[1][0]))
