"""Several proposals of ONE chain per step (free_sweep_spec.h, the kernel for runs with few chains): the chain is the
sequential chain of the one-proposal kernel and of the CPU oracle -- positions, Gbar, chi2, windows, accept rates,
counters, the histogram and every record of the move log bit for bit -- for 2 and 4 proposals per step and every tau-slot
mapping."""
import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_free_packed import CASES

pytestmark = pytest.mark.gpu


def _build(kw):
    kw = dict(kw)
    kind = kw.pop("kind")
    return helpers.problem_golden_bosonic(**kw) if kind == "golden" else helpers.problem_synth(**kw)


@pytest.mark.parametrize("name,kw,theta,n_adj,n_smp,cpws", CASES, ids=[c[0] for c in CASES])
def test_speculative_chain_equals_the_oracle(name, kw, theta, n_adj, n_smp, cpws):
    prob = _build(kw)
    seed, n_chains = 20250607, 8
    oracles = [helpers.oracle_chain(prob, seed, 3 + c) for c in range(n_chains)]
    log_chain = 5
    cap = (prob.n_omega + prob.n_omega // 2 + prob.n_omega // 3) * ((n_adj // 10) * 10 + n_smp)
    oracles[log_chain].enable_log(cap)
    for o in oracles:
        o.adjust_windows(n_adj, theta)
        o.sample(n_smp, theta)
    hist_o = sum(o.spec() for o in oracles) / n_chains
    for cpw in cpws:                                   # here: proposals of one chain per step
        h = prob.make_handle(n_chains=n_chains, seed=seed, chain_offset=3, hist_copies=n_chains, sweeps_per_launch=7)
        h.set_chains_per_wave(1)
        h.set_proposals_per_step(cpw)
        h.set_move_log(log_chain, cap)
        h.adjust_windows(n_adj, theta)
        assert h.chains_per_wave() == 1 and h.proposals_per_step() == cpw
        h.sample(n_smp, theta)
        log_g, n_g = h.move_log(cap)
        log_o = oracles[log_chain].log()
        assert n_g == len(log_o) == cap
        assert np.array_equal(log_g["move"], log_o["move"]) and np.array_equal(log_g["flags"], log_o["flags"])
        assert np.array_equal(log_g["chi2_new"], log_o["chi2_new"])     # every proposal's chi^2, bit for bit
        for c, o in enumerate(oracles):
            wi_g, A_g, win_g = h.get_state(c)
            wi_o, A_o, win_o = o.get_state()
            assert np.array_equal(wi_g, wi_o), (cpw, c)
            assert np.array_equal(win_g, win_o), (cpw, c)
            g_g, c_g, m_g = h.get_gbar(c)
            g_o, c_o, m_o = o.get_gbar()
            assert np.array_equal(g_g, g_o), (cpw, c)
            assert c_g == c_o and m_g == m_o, (cpw, c)
            cs = h.chain_stats(c)
            assert np.array_equal(cs["accept_rates"], o.accept_rates())
            assert cs["sampled_chi2"] == o.sampled_chi2()
            assert cs["sweeps"] == o.sweep_counter()
        np.testing.assert_allclose(h.histogram(), hist_o, rtol=1e-13, atol=1e-300)
        st = h.stats()
        assert st.proposals == sum(o.counters()[0] for o in oracles)
        assert st.evaluated == sum(o.counters()[1] for o in oracles)
        assert st.accepted == sum(o.counters()[2] for o in oracles)
        assert st.accepted_moves == sum(o.counters()[3] for o in oracles)
        assert st.chi2_min == min(o.get_gbar()[2] for o in oracles)
        h.close()


def test_speculation_is_the_default_for_few_chains_only():
    prob = helpers.problem_golden_bosonic(n_omega=60)
    h = prob.make_handle(n_chains=8, seed=1)
    h.sample(3, 1.0)
    assert h.proposals_per_step() == 2                 # few chains: the chip is empty, latency counts
    h.set_proposals_per_step(1)
    h.sample(3, 1.0)
    assert h.proposals_per_step() == 1
    h.close()
    prob = helpers.problem_synth(n_tau=80, n_omega=100, kernel_type="zeroT", omega_0=-2.0, omega_m=8.0, d_omega=1e-3)
    h = prob.make_handle(n_chains=4, seed=1)
    h.set_proposals_per_step(4)                        # more than 64 time points: one proposal per step
    h.sample(3, 1.0)
    assert h.proposals_per_step() == 1
    h.close()
    prob = helpers.problem_golden_bosonic(n_omega=60)
    h = prob.make_handle(n_chains=2048, seed=1)        # enough chains to fill the chip: throughput kernels
    h.sample(2, 1.0)
    assert h.proposals_per_step() == 1
    h.close()


def test_speculative_equals_plain_over_a_long_run():
    """5 chains, 300 sweeps with window tuning at a low and a high acceptance: 1, 2 and 4 proposals per step end in the
    same state (a long run meets the rare paths: proposals decoded again inside a step, refetched columns)."""
    for kw, theta in ((dict(n_tau=22, n_omega=40, omega_m=6.0, d_omega=1e-3), 0.05), (dict(n_tau=64, n_omega=64, omega_m=8.0, d_omega=1e-3), 20.0)):
        prob = helpers.problem_synth(**kw)
        ref = None
        for pps in (1, 2, 4):
            h = prob.make_handle(n_chains=5, seed=4242, hist_copies=5)
            h.set_proposals_per_step(pps)
            h.adjust_windows(200, theta)
            h.sample(100, theta)
            assert h.proposals_per_step() == pps
            out = ([h.get_state(c) for c in range(5)], [h.get_gbar(c) for c in range(5)], h.histogram(), h.stats())
            if ref is None:
                ref = out
            else:
                for (w, A, win), (w0, A0, win0) in zip(out[0], ref[0]):
                    assert np.array_equal(w, w0) and np.array_equal(win, win0)
                for (g, c2, cm), (g0, c20, cm0) in zip(out[1], ref[1]):
                    assert np.array_equal(g, g0) and c2 == c20 and cm == cm0
                assert np.array_equal(out[2], ref[2])
                for k in ("proposals", "evaluated", "accepted", "accepted_moves", "columns_gathered"):
                    assert getattr(out[3], k) == getattr(ref[3], k), k
            h.close()
