"""Several Markov chains per wavefront (free_sweep_packed.h): the packed kernels run the SAME chains as the
one-chain-per-wave kernel and the CPU oracle -- positions, Gbar, chi2, windows, accept rates, counters and the
per-chain histogram bit for bit -- for every (chains per wave, tau slots per lane) mapping."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

CASES = [
    # name, problem kwargs, theta, adjust steps, sample steps, chains per wave to try
    ("golden_ntau22", dict(kind="golden", n_omega=60), 2.0, 40, 25, (2, 4)),
    ("golden_ntau22_n1000", dict(kind="golden", n_omega=1000, omega_m=5.0, d_omega=1e-3), 1.0, 20, 8, (2, 4)),
    ("synth_ntau32", dict(kind="synth", n_tau=32, n_omega=500, omega_m=8.0, d_omega=1e-3), 1.0, 20, 8, (2, 4)),
    ("synth_ntau13", dict(kind="synth", n_tau=13, n_omega=130, omega_m=8.0, d_omega=1e-3), 3.0, 20, 8, (2, 4)),
    ("synth_ntau64_n2000", dict(kind="synth", n_tau=64, n_omega=2000), 1.0, 10, 6, (2, 4)),
    ("synth_ntau40_finiteT", dict(kind="synth", n_tau=40, n_omega=333, kernel_type="finiteT",
                                  omega_0=-6.0, omega_m=6.0, d_omega=1e-3), 5.0, 20, 10, (2, 4)),
    ("synth_ntau64_n2048", dict(kind="synth", n_tau=64, n_omega=2048, omega_m=10.0, d_omega=1e-3), 1.0, 10, 3, (2, 4)),
    ("synth_ntau16_n333", dict(kind="synth", n_tau=16, n_omega=333, omega_m=8.0, d_omega=1e-3), 2.0, 20, 8, (2, 4)),
]


def _build(kw):
    kw = dict(kw)
    kind = kw.pop("kind")
    return helpers.problem_golden_bosonic(**kw) if kind == "golden" else helpers.problem_synth(**kw)


@pytest.mark.parametrize("name,kw,theta,n_adj,n_smp,cpws", CASES, ids=[c[0] for c in CASES])
def test_packed_chains_equal_the_oracle(name, kw, theta, n_adj, n_smp, cpws):
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
    for cpw in cpws:
        h = prob.make_handle(n_chains=n_chains, seed=seed, chain_offset=3, hist_copies=n_chains, sweeps_per_launch=7)
        h.set_chains_per_wave(cpw)
        h.set_move_log(log_chain, cap)
        h.adjust_windows(n_adj, theta)
        assert h.chains_per_wave() == cpw
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


def test_packing_falls_back_where_it_does_not_apply():
    prob = helpers.problem_golden_bosonic(n_omega=60)
    h = prob.make_handle(n_chains=3, seed=1)          # 3 chains do not fill whole waves of 2
    h.set_chains_per_wave(2)
    h.sample(3, 1.0)
    assert h.chains_per_wave() == 1
    h.close()
    prob = helpers.problem_synth(n_tau=80, n_omega=100, kernel_type="zeroT", omega_0=-2.0, omega_m=8.0, d_omega=1e-3)
    h = prob.make_handle(n_chains=4, seed=1)
    h.set_chains_per_wave(4)                           # more than 64 time points: one chain per wave
    h.sample(3, 1.0)
    assert h.chains_per_wave() == 1
    h.close()
    import sac_amd
    with pytest.raises(sac_amd.SacError, match="chains_per_wave"):
        h2 = prob.make_handle(n_chains=4, seed=1)
        h2.set_chains_per_wave(3)


def test_packed_equals_unpacked_at_the_baseline_shape():
    """BASELINE configs[1] shape, 64 chains: 2 and 4 chains per wave against one chain per wave, after 30 sweeps."""
    prob = helpers.problem_synth(n_tau=64, n_omega=2000)
    ref = None
    for cpw in (1, 2, 4):
        h = prob.make_handle(n_chains=64, seed=77)
        h.set_chains_per_wave(cpw)
        h.adjust_windows(20, 1.0)
        h.sample(10, 1.0)
        assert h.chains_per_wave() == cpw
        out = ([h.get_state(c)[0] for c in (0, 31, 63)], [h.get_gbar(c) for c in (0, 31, 63)], h.histogram(),
               h.stats())
        if ref is None:
            ref = out
        else:
            for a, b in zip(out[0], ref[0]):
                assert np.array_equal(a, b)
            for (g, c2, cm), (g0, c20, cm0) in zip(out[1], ref[1]):
                assert np.array_equal(g, g0) and c2 == c20 and cm == cm0
            np.testing.assert_allclose(out[2], ref[2], rtol=1e-12, atol=1e-300)
            assert out[3].accepted_moves == ref[3].accepted_moves and out[3].evaluated == ref[3].evaluated
            assert out[3].columns_gathered == ref[3].columns_gathered
        h.close()
