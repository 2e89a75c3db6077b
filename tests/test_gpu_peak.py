"""GPU parity for the peak parameterisation: libsac_hip.so (SAC_KIND_PEAK, through the C ABI)
against the CPU oracle of sac_peak.jl -- move for move, bit for bit."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

CASES = [
    # name, problem kwargs, indiv_update, theta, adjust steps, sample steps, bin steps
    ("bosonic_ntau22", dict(kind="golden", kernel_type="bosonic"), False, 5.0, 20, 10, 5),
    ("symm_finiteT_ntau22", dict(kind="golden", kernel_type="finiteT", symm=1), False, 5.0, 20, 10, 5),
    ("both_axes_ntau22", dict(kind="golden", kernel_type="finiteT", approx_ratio=1.7), False, 20.0, 20, 10, 5),
    ("both_axes_indiv_ntau22", dict(kind="golden", kernel_type="finiteT", approx_ratio=1.7, n_p=1), True,
     20.0, 20, 10, 5),
    ("both_axes_ntau64_n400", dict(kind="synth", kernel_type="finiteT", approx_ratio=2.5, n_tau=64, n_omega=400,
                                   n_p=3, omega_m=8.0, d_omega=1e-3), True, 10.0, 10, 6, 3),
    ("zeroT_ntau80_fixedge", dict(kind="synth", kernel_type="zeroT", approx_ratio=np.inf, n_tau=80, n_omega=200,
                                  n_p=1, omega_m=8.0, d_omega=1e-3, fix_edge=1, omega_0=0.4), False, 5.0, 10, 6, 3),
    # a dozen deltas per group: nearly every accepted move touches a delta of a later, already decoded proposal of the same group of
    # 64 (re-decodes, stale ring entries, proposals entering and leaving the bounds, ring restarts of the live-proposal loops)
    ("both_axes_tiny_n12", dict(kind="golden", kernel_type="finiteT", approx_ratio=1.7, n_omega=12, n_p=2), True, 20.0, 60, 40, 10),
    ("bosonic_tiny_n9_cold", dict(kind="golden", kernel_type="bosonic", n_omega=9, n_p=1), False, 0.5, 60, 40, 10),
    # the limits of the handle: N_omega = SAC_MAX_N_OMEGA = 2048 continuum deltas per axis (+ 11 peak deltas), both axes, N_tau = 128
    ("max_state_both_axes_ntau128", dict(kind="synth", kernel_type="finiteT", approx_ratio=2.0, n_tau=128, n_omega=2048,
                                         n_p=11, omega_m=8.0, d_omega=1e-3), True, 10.0, 10, 2, 1),
]


def _compare(h, o, chain, sampled):
    wi_g, A_g, win_g = h.get_state(chain)
    wi_o, A_o, win_o = o.get_state()
    assert np.array_equal(wi_g, wi_o)
    assert np.array_equal(A_g, A_o)
    assert np.array_equal(win_g, win_o)
    g_g, c_g, m_g = h.get_gbar(chain)
    g_o, c_o, m_o = o.get_gbar()
    assert np.array_equal(g_g, g_o) and c_g == c_o and m_g == m_o
    assert np.array_equal(h.bounds(chain), o.bounds())
    cs = h.chain_stats(chain)
    assert np.array_equal(cs["accept_rates"], o.accept_rates())
    assert cs["sweeps"] == o.sweep_counter()
    if sampled:
        assert cs["sampled_chi2"] == o.sampled_chi2()
        assert np.array_equal(cs["edge_mean"], o.edge_res()[0])


@pytest.mark.parametrize("name,kw,indiv,theta,n_adj,n_smp,n_bin", CASES, ids=[c[0] for c in CASES])
def test_peak_move_for_move_parity(name, kw, indiv, theta, n_adj, n_smp, n_bin):
    prob = helpers.problem_peak(**kw)
    seed, n_chains, log_chain = 777, 3, 2
    h = prob.make_handle(n_chains=n_chains, seed=seed, chain_offset=4, hist_copies=n_chains, sweeps_per_launch=4)
    h.set_indiv_update(indiv)
    oracles = [helpers.oracle_peak_chain(prob, seed, 4 + c, indiv) for c in range(n_chains)]
    cap = 400000
    h.set_move_log(log_chain, cap)
    oracles[log_chain].enable_log(cap)
    for c, o in enumerate(oracles):
        g_g, c_g, m_g = h.get_gbar(c)
        g_o, c_o, m_o = o.get_gbar()
        assert np.array_equal(g_g, g_o) and c_g == c_o and m_g == m_o
        assert np.array_equal(h.bounds(c), o.bounds())

    h.adjust_windows(n_adj, theta)
    for o in oracles:
        o.adjust_windows(n_adj, theta)
    for c, o in enumerate(oracles):
        _compare(h, o, c, False)

    h.sample(n_smp, theta)
    for o in oracles:
        o.sample(n_smp, theta)
    for c, o in enumerate(oracles):
        _compare(h, o, c, True)
    for which in range(4):
        ref = sum(o.spec(which) for o in oracles) / n_chains
        np.testing.assert_allclose(h.histogram_peak(which), ref, rtol=1e-12, atol=1e-300)

    # final-sampling bins (peak:1370-1377): accumulators zeroed, then bins without weight transfer
    h.zero_accumulators()
    for o in oracles:
        o.zero_spec()
        o.set_edge_res(np.zeros(6))
    for _ in range(2):
        h.sample_bin(n_bin, theta)
        for o in oracles:
            o.sample(n_bin, theta, bin=True)
    for c, o in enumerate(oracles):
        _compare(h, o, c, True)
    for which in range(4):
        ref = sum(o.spec(which) for o in oracles) / n_chains
        np.testing.assert_allclose(h.histogram_peak(which), ref, rtol=1e-12, atol=1e-300)
    st = h.peak_stats()
    np.testing.assert_allclose(np.array(st.edge_res).reshape(3, 2), np.mean([o.edge_res() for o in oracles], axis=0),
                               rtol=1e-13)
    assert st.chi2_min == min(o.get_gbar()[2] for o in oracles)
    assert st.proposals == sum(o.counters()[0] for o in oracles)
    assert st.evaluated == sum(o.counters()[1] for o in oracles)
    assert st.accepted == sum(o.counters()[2] for o in oracles)
    assert st.accepted_moves == sum(o.counters()[3] for o in oracles)
    assert st.columns_gathered == sum(o.counters()[4] for o in oracles)
    wts = np.array(st.weights)
    N, P = prob.n_omega, prob.n_p
    A_mean = np.mean([o.get_state()[1] for o in oracles], axis=0)
    np.testing.assert_allclose(wts, [A_mean[:P].sum(), A_mean[P:P + N].sum(), A_mean[P + N:2 * P + N].sum(),
                                     A_mean[2 * P + N:].sum()], rtol=1e-12, atol=1e-300)

    log_g, n_g = h.move_log(cap)
    log_o = oracles[log_chain].log()
    assert n_g == len(log_o) and n_g < cap
    assert np.array_equal(log_g["move"], log_o["move"])
    assert np.array_equal(log_g["flags"], log_o["flags"])
    assert np.array_equal(log_g["chi2_new"], log_o["chi2_new"])
    assert (log_g["flags"] & 2).sum() > 0
    h.close()
