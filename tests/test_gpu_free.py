"""GPU parity tests proper: libsac_hip.so (through the C ABI) against the CPU oracle.

Bit-exact where the arithmetic is defined (per-move chi^2, accept decisions, positions, Gbar,
windows, per-chain histogram); size-independent invariants at the BASELINE size.
"""
import numpy as np
import pytest

from tests import helpers
from sac_amd.lib import HIST_ACCUMULATE, SacError

pytestmark = pytest.mark.gpu


def _compare_chain(h, o, chain, steps_sampled=True):
    wi_g, A_g, win_g = h.get_state(chain)
    wi_o, A_o, win_o = o.get_state()
    assert np.array_equal(wi_g, wi_o)
    assert np.array_equal(A_g, A_o)
    assert np.array_equal(win_g, win_o)
    g_g, c_g, m_g = h.get_gbar(chain)
    g_o, c_o, m_o = o.get_gbar()
    assert np.array_equal(g_g, g_o)
    assert c_g == c_o and m_g == m_o
    cs = h.chain_stats(chain)
    assert np.array_equal(cs["accept_rates"], o.accept_rates())
    if steps_sampled:
        assert cs["sampled_chi2"] == o.sampled_chi2()
    assert cs["sweeps"] == o.sweep_counter()


CASES = [
    # name, problem builder kwargs, theta, adjust steps, sample steps
    ("golden_ntau22", dict(kind="golden", n_omega=60), 2.0, 40, 25),
    ("golden_ntau22_n1000", dict(kind="golden", n_omega=1000, omega_m=5.0, d_omega=1e-3), 1.0, 20, 8),
    ("synth_ntau64_n2000", dict(kind="synth", n_tau=64, n_omega=2000), 1.0, 10, 6),
    ("synth_ntau40_finiteT", dict(kind="synth", n_tau=40, n_omega=333, kernel_type="finiteT",
                                  omega_0=-6.0, omega_m=6.0, d_omega=1e-3), 5.0, 20, 10),
    # BASELINE.json configs[0]'s shape: fermionic (finite-T) data, N_tau = 40, 1000 delta functions
    ("configs0_ntau40_finiteT_n1000", dict(kind="synth", n_tau=40, n_omega=1000, kernel_type="finiteT",
                                           omega_0=-6.0, omega_m=6.0, d_omega=1e-3), 2.0, 10, 5),
    ("synth_ntau80_zeroT", dict(kind="synth", n_tau=80, n_omega=500, kernel_type="zeroT",
                                omega_0=-2.0, omega_m=8.0, d_omega=1e-3), 3.0, 20, 10),
    ("synth_ntau102_symm", dict(kind="synth", n_tau=102, n_omega=100, kernel_type="finiteT",
                                symm=1, omega_m=8.0, d_omega=2e-3), 10.0, 20, 10),
    ("synth_ntau64_par3", dict(kind="synth", n_tau=64, n_omega=300, par=3, omega_m=8.0, d_omega=1e-3),
     2.0, 20, 10),
    # par = 2: amplitude updates (A_w_move, double_A_move, renormalisation after each)
    ("golden_ntau22_par2", dict(kind="golden", n_omega=60, par=2), 2.0, 40, 25),
    ("synth_ntau64_par2", dict(kind="synth", n_tau=64, n_omega=700, par=2, omega_m=8.0, d_omega=1e-3),
     1.0, 20, 8),
    ("synth_ntau80_par2_symm", dict(kind="synth", n_tau=80, n_omega=150, par=2, kernel_type="finiteT",
                                    symm=1, omega_m=6.0, d_omega=2e-3), 5.0, 20, 10),
    # the limits of the handle: SAC_MAX_N_TAU = 128 time points, SAC_MAX_N_OMEGA = 2048 delta functions
    ("max_ntau128_n2048", dict(kind="synth", n_tau=128, n_omega=2048, omega_m=10.0, d_omega=1e-3), 1.0, 10, 3),
    ("max_ntau128_n2048_par2", dict(kind="synth", n_tau=128, n_omega=2048, par=2, omega_m=10.0, d_omega=1e-3),
     1.0, 10, 2),
]


def _build(kw):
    kw = dict(kw)
    kind = kw.pop("kind")
    return helpers.problem_golden_bosonic(**kw) if kind == "golden" else helpers.problem_synth(**kw)


@pytest.mark.parametrize("name,kw,theta,n_adj,n_smp", CASES, ids=[c[0] for c in CASES])
def test_move_for_move_parity(name, kw, theta, n_adj, n_smp):
    prob = _build(kw)
    seed, n_chains, log_chain = 20250511, 3, 1
    h = prob.make_handle(n_chains=n_chains, seed=seed, chain_offset=7, hist_copies=n_chains,
                         sweeps_per_launch=7)
    h.set_proposals_per_step(1)          # this test is free_stage_kernel's; tests/test_gpu_free_spec.py covers 2 and 4
    per_sweep = prob.n_omega + prob.n_omega // 2 + prob.n_omega // 3
    if prob.par == 2:
        per_sweep += prob.n_omega // 2 + prob.n_omega
    n_sweeps = (n_adj // 10) * 10 + n_smp
    cap = per_sweep * n_sweeps
    h.set_move_log(log_chain, cap)
    oracles = [helpers.oracle_chain(prob, seed, 7 + c) for c in range(n_chains)]
    oracles[log_chain].enable_log(cap)

    # initial Gbar / chi2 (free:1213-1216)
    for c, o in enumerate(oracles):
        g_g, c_g, m_g = h.get_gbar(c)
        g_o, c_o, m_o = o.get_gbar()
        assert np.array_equal(g_g, g_o) and c_g == c_o and m_g == m_o

    h.adjust_windows(n_adj, theta)
    for o in oracles:
        o.adjust_windows(n_adj, theta)
    for c, o in enumerate(oracles):
        _compare_chain(h, o, c, steps_sampled=False)

    h.sample(n_smp, theta)
    for o in oracles:
        o.sample(n_smp, theta)
    for c, o in enumerate(oracles):
        _compare_chain(h, o, c)

    log_g, n_g = h.move_log(cap)
    log_o = oracles[log_chain].log()
    assert n_g == len(log_o) == cap
    assert np.array_equal(log_g["move"], log_o["move"])
    assert np.array_equal(log_g["flags"], log_o["flags"])
    assert np.array_equal(log_g["chi2_new"], log_o["chi2_new"])     # bit for bit
    assert (log_g["flags"] & 2).sum() > 0 and (log_g["flags"] & 1).sum() > (log_g["flags"] & 2).sum() // 2

    # histogram: mean over chains of per-chain sweep averages
    hist_o = sum(o.spec() * n_smp for o in oracles) / n_smp / n_chains
    hist_g = h.histogram()
    np.testing.assert_allclose(hist_g, hist_o, rtol=1e-13, atol=1e-300)
    st = h.stats()
    assert st.chi2_min == min(o.get_gbar()[2] for o in oracles)
    np.testing.assert_allclose(st.sampled_chi2, np.mean([o.sampled_chi2() for o in oracles]), rtol=1e-14)
    assert st.proposals == sum(o.counters()[0] for o in oracles)
    assert st.evaluated == sum(o.counters()[1] for o in oracles)
    assert st.accepted == sum(o.counters()[2] for o in oracles)
    assert st.accepted_moves == sum(o.counters()[3] for o in oracles)
    h.close()


@pytest.mark.parametrize("mode", [0, HIST_ACCUMULATE])
def test_single_chain_histogram_bit_exact(mode):
    # duplicates are frequent here: 400 deltas on a 501-point grid
    prob = helpers.problem_golden_bosonic(n_omega=400, omega_m=1.0, d_omega=2e-3)
    h = prob.make_handle(n_chains=1, seed=99, hist_mode=mode)
    o = helpers.oracle_chain(prob, 99, 0, hist_mode=mode)
    h.sample(12, 0.5)
    o.sample(12, 0.5)
    wi, _, _ = h.get_state(0)
    assert len(np.unique(wi)) < len(wi), "test needs coincident deltas"
    assert np.array_equal(h.histogram(), o.spec())
    if mode == HIST_ACCUMULATE:
        assert abs(h.histogram().sum() - 1.0) < 1e-12
    else:
        assert h.histogram().sum() < 1.0
    h.close()


def test_shared_histogram_replicas_atomics():
    # more chains than replicas: fp64 hardware atomics, order not fixed -> tolerance 1e-12
    prob = helpers.problem_golden_bosonic(n_omega=200, omega_m=2.0, d_omega=1e-3)
    n_chains = 12
    h = prob.make_handle(n_chains=n_chains, seed=5, hist_copies=2)
    h.sample(10, 1.0)
    oracles = [helpers.oracle_chain(prob, 5, c) for c in range(n_chains)]
    for o in oracles:
        o.sample(10, 1.0)
    ref = sum(o.spec() for o in oracles) / n_chains
    np.testing.assert_allclose(h.histogram(), ref, rtol=1e-12, atol=1e-300)
    h.close()


def test_baseline_size_invariants():
    """BASELINE config 2 shape (N_tau=64, 2000 deltas, 10^5-point grid, 4096 chains): properties
    that hold at any size -- bounds, Gbar-incremental == Gbar-from-scratch, chi2 == tree(Gbar),
    chain independence from the launch shape (chain 0 equals a 1-chain handle's chain)."""
    prob = helpers.problem_synth(n_tau=64, n_omega=2000)
    h = prob.make_handle(n_chains=4096, seed=1)
    h.adjust_windows(10, 1.0)
    h.sample(5, 1.0)
    st = h.stats()
    per_sweep = 2000 + 1000 + 666
    assert st.sweeps == 4096 * 15 and st.proposals == 4096 * 15 * per_sweep
    assert st.accepted <= st.evaluated <= st.proposals
    hist = h.histogram()
    assert hist.min() >= 0 and 0.5 < hist.sum() <= 1.0 + 1e-12
    from oracle import oracle as ora
    for c in (0, 1777, 4095):
        wi, A, win = h.get_state(c)
        assert wi.min() >= prob.wi_0 and wi.max() <= prob.wi_m
        g, c2, cmin = h.get_gbar(c)
        ref = prob.K[wi - prob.wi_0].T @ A
        assert np.abs(g - ref).max() <= 1e-11 * np.abs(ref).max()
        assert c2 == ora.chi2(g, prob.td.G, prob.td.sigma_inv) and cmin <= c2
    h1 = prob.make_handle(n_chains=1, seed=1)
    h1.adjust_windows(10, 1.0)
    h1.sample(5, 1.0)
    assert np.array_equal(h1.get_state(0)[0], h.get_state(0)[0])
    assert h1.get_gbar(0)[1] == h.get_gbar(0)[1]
    h.close()
    h1.close()


def test_error_paths_do_not_exit():
    import sac_amd
    with pytest.raises(sac_amd.SacError, match="n_omega"):
        sac_amd.SacHandle(22, 5000, 100, 0, 99)
    with pytest.raises(sac_amd.SacError, match="n_tau"):
        sac_amd.SacHandle(200, 50, 100, 0, 99)
    h = sac_amd.SacHandle(22, 50, 100, 0, 99)
    with pytest.raises(sac_amd.SacError, match="first"):
        h.sample(10, 1.0)
    with pytest.raises(sac_amd.SacError, match="outside"):
        h.set_state(wi=np.full(50, 500, dtype=np.int64))
    h.close()


def test_new_positions_invalidate_gbar():
    """Writing positions / amplitudes / a problem on an initialised handle leaves Gbar and chi2 stale (the free
    kernels never rebuild them, free:876): the stages refuse to run (SAC_E_STATE) until sac_init_gbar has run again."""
    prob = helpers.problem_golden_bosonic(n_omega=60, omega_m=3.0, d_omega=1e-3)
    h = prob.make_handle(n_chains=2, seed=1)
    h.sample(3, 1.0)
    wi, A, win = h.get_state(0)
    h.set_state(windows=win)                     # windows only: Gbar stays valid
    h.sample(3, 1.0)
    for write in (lambda: h.set_state(wi=wi, chain=1), lambda: h.set_state(A=A),
                  lambda: h.upload_problem(prob.K, prob.td.G, prob.td.sigma_inv)):
        write()
        with pytest.raises(SacError, match="sac_init_gbar"):
            h.sample(3, 1.0)
        with pytest.raises(SacError, match="sac_init_gbar"):
            h.adjust_windows(10, 1.0)
        h.init_gbar()
        h.sample(3, 1.0)
    # the replica sum is written by a kernel: a host pointer is refused, not written through
    host = np.zeros(prob.n_grid)
    with pytest.raises(SacError, match="device"):
        h.histogram_sum_to_device(host.ctypes.data)
    h.close()


def test_replay_of_an_explicit_proposal_stream():
    """sac_replay: a foreign proposal stream (indices, new positions, accept draws) evaluated by
    the library agrees with a direct fp64 evaluation to <= 1e-12 relative per move (the north-star tolerance),
    takes the same accept decisions and leaves a consistent chain behind."""
    from sac_amd.lib import PROPOSAL
    prob = helpers.problem_synth(n_tau=64, n_omega=200, omega_m=6.0, d_omega=1e-3)
    h = prob.make_handle(n_chains=3, seed=5)
    h.adjust_windows(20, 2.0)
    chain, theta = 1, 2.0
    wi, A, _ = h.get_state(chain)
    gbar, chi2, _ = h.get_gbar(chain)
    rng = np.random.default_rng(11)
    n = 400
    props = np.zeros(n, dtype=PROPOSAL)
    wi_sim, g_sim, chi2_sim = wi.copy(), gbar.copy(), chi2
    G, sinv, K = prob.td.G, prob.td.sigma_inv, prob.K
    exp_chi2, exp_acc = np.empty(n), np.empty(n, dtype=np.int32)
    for i in range(n):
        m = int(rng.integers(1, 4))
        ks = rng.choice(prob.n_omega, size=m, replace=False)
        new = np.clip(wi_sim[ks] + rng.integers(-300, 301, size=m), prob.wi_0, prob.wi_m)
        props[i]["move"], props[i]["u"] = m, rng.random()
        props[i]["k"][:m], props[i]["w_new"][:m] = ks, new
        gn = g_sim.copy()
        for e in range(m):
            gn = gn + A[ks[e]] * (K[new[e] - prob.wi_0] - K[wi_sim[ks[e]] - prob.wi_0])
        c2 = float((((gn - G) * sinv) ** 2).sum())
        acc = props[i]["u"] <= np.exp(min((chi2_sim - c2) / (2 * theta), 50.0))
        exp_chi2[i], exp_acc[i] = c2, acc
        if acc:
            wi_sim[ks], g_sim, chi2_sim = new, gn, c2
    c2_gpu, acc_gpu = h.replay(props, theta, chain=chain)
    np.testing.assert_allclose(c2_gpu, exp_chi2, rtol=1e-12)
    assert np.array_equal(acc_gpu, exp_acc) and 0 < acc_gpu.sum() < n
    wi_g, _, _ = h.get_state(chain)
    assert np.array_equal(wi_g, wi_sim)
    g_g, chi2_g, _ = h.get_gbar(chain)
    np.testing.assert_allclose(g_g, g_sim, rtol=1e-11, atol=1e-15)
    assert chi2_g == pytest.approx(chi2_sim, rel=1e-12)
    # the neighbouring chains are untouched; bad proposals are refused, not evaluated
    bad = props[:1].copy()
    bad[0]["w_new"][0] = prob.wi_m + 5
    with pytest.raises(SacError):
        h.replay(bad, theta, chain=chain)
    h.sample(5, theta)                       # the chain keeps running from the replayed state
    h.close()
