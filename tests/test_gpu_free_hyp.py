"""Hypothesis pairs (free_sweep_hyp.h, the second kernel for runs with few chains): two consecutive proposals of a chain are
resolved per step -- chi2(a), chi2(b | a rejected), chi2(b | a accepted) evaluated together, a committed first, then b under
a's outcome -- and the chain is the sequential chain of the one-proposal kernel and of the CPU oracle: positions, Gbar, chi2,
windows, accept rates, counters, the histogram and every record of the move log bit for bit.  Shapes: every tau mapping
(N_tau 13 ... 128), 60 ... 2048 deltas (few deltas: nearly every pair shares a delta or goes stale -> the fallback paths)."""
import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_free_packed import CASES

pytestmark = pytest.mark.gpu

HYP_CASES = [(c[0], c[1], c[2], c[3], c[4]) for c in CASES] + [
    ("synth_ntau20_n7", dict(kind="synth", n_tau=20, n_omega=7, omega_m=8.0, d_omega=1e-3), 2.0, 40, 30),
    ("synth_ntau80_n100_zeroT", dict(kind="synth", n_tau=80, n_omega=100, kernel_type="zeroT", omega_0=-2.0, omega_m=8.0,
                                    d_omega=1e-3), 1.0, 20, 10),
    ("synth_ntau128_n2048", dict(kind="synth", n_tau=128, n_omega=2048, kernel_type="zeroT", omega_0=-2.0, omega_m=8.0,
                                 d_omega=1e-3), 1.0, 10, 2),
]


def _build(kw):
    kw = dict(kw)
    kind = kw.pop("kind")
    return helpers.problem_golden_bosonic(**kw) if kind == "golden" else helpers.problem_synth(**kw)


@pytest.mark.parametrize("name,kw,theta,n_adj,n_smp", HYP_CASES, ids=[c[0] for c in HYP_CASES])
def test_hypothesis_pair_chain_equals_the_oracle(name, kw, theta, n_adj, n_smp):
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
    h = prob.make_handle(n_chains=n_chains, seed=seed, chain_offset=3, hist_copies=n_chains, sweeps_per_launch=7)
    h.set_chains_per_wave(1)
    h.set_hypothesis_pairs(1)
    h.set_move_log(log_chain, cap)
    h.adjust_windows(n_adj, theta)
    assert h.chains_per_wave() == 1 and h.hypothesis_pairs() == 1 and h.proposals_per_step() == 2
    h.sample(n_smp, theta)
    log_g, n_g = h.move_log(cap)
    log_o = oracles[log_chain].log()
    assert n_g == len(log_o) == cap
    assert np.array_equal(log_g["move"], log_o["move"]) and np.array_equal(log_g["flags"], log_o["flags"])
    assert np.array_equal(log_g["chi2_new"], log_o["chi2_new"])     # every proposal's chi^2, bit for bit
    for c, o in enumerate(oracles):
        wi_g, A_g, win_g = h.get_state(c)
        wi_o, A_o, win_o = o.get_state()
        assert np.array_equal(wi_g, wi_o), c
        assert np.array_equal(win_g, win_o), c
        g_g, c_g, m_g = h.get_gbar(c)
        g_o, c_o, m_o = o.get_gbar()
        assert np.array_equal(g_g, g_o), c
        assert c_g == c_o and m_g == m_o, c
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


def test_hypothesis_pairs_off_restores_the_other_kernels():
    prob = helpers.problem_golden_bosonic(n_omega=60)
    h = prob.make_handle(n_chains=8, seed=1)
    h.set_hypothesis_pairs(1)
    h.sample(3, 1.0)
    assert h.hypothesis_pairs() == 1
    h.set_hypothesis_pairs(-1)
    h.set_proposals_per_step(1)
    h.sample(3, 1.0)
    assert h.hypothesis_pairs() == 0 and h.proposals_per_step() == 1
    h.close()
