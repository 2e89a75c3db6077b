"""GPU parity for the edge parameterisation: libsac_hip.so (sac_edge_*, through the C ABI) against the CPU
oracle of sac_edge.jl -- move for move, bit for bit (Taylor lookup, ordering checks, cluster moves with
their sort, amplitude profile with the shared log/exp/sqrt sequence, histograms, per-bin retuning)."""
import numpy as np
import pytest

from sac_amd.lib import SacEdgeHandle, SacError
from tests.test_oracle_edge import make_chain

pytestmark = pytest.mark.gpu

CASES = [
    # name, run, d_omega, n_e, n_c, A_c, p, stages [(steps, bins, theta)]
    ("symm_ntau31", "t2_symm", 2e-3, 24, 0, 0.0, 0.5, [(12, 3, 10.0), (8, 2, 1.0)]),
    ("single_bosonic_p04", "hchain_single", 2e-3, 30, 0, 0.0, 0.4, [(12, 3, 10.0), (8, 2, 0.5)]),
    ("double_out", "t4_out", 2e-3, 20, 0, 0.0, 0.5, [(10, 3, 10.0), (6, 2, 2.0)]),
    ("double_in_p03", "t3_in", 2e-3, 20, 0, 0.0, 0.3, [(10, 3, 10.0), (6, 2, 2.0)]),
    ("single_continuum_ntau80", "t1_single", 2e-3, 24, 12, 0.3, 0.6, [(10, 3, 10.0), (6, 2, 2.0)]),
    ("single_ne80", "hchain_single", 1e-3, 80, 0, 0.0, 0.5, [(6, 2, 5.0)]),
    # the limit of the handle: SAC_MAX_N_EDGE = 256 edge deltas (N_tau = 80)
    ("max_ne256_ntau80", "t1_single", 1e-3, 256, 0, 0.0, 0.5, [(4, 2, 5.0)]),
]


def make_handle(tin, tab, init, n_chains, seed, **kw):
    h = SacEdgeHandle(tin["n_tau"], init.n_e, init.n_omega, tab.n_grid, tab.wi0, tab.dw, init.w_m, init.w_floor, init.c,
                      n_rl=tab.n_rl, double_in=int(init.mode == "double_edge_in"), fix_edge=init.fix_edge,
                      A_c=init.A_c, A_r=init.A_r, n_chains=n_chains, seed=seed, hist_copies=n_chains, **kw)
    h.upload_problem(tab.full(), tin["G_D"], tin["sigma_inv"])
    h.set_state(init.w, init.A, init.dw, init.nm, init.scalars())
    h.init_gbar()
    return h


def compare(h, o, chain, n_rl):
    sg, so = h.get_state(chain), o.get_state()
    for key in ("w", "A", "dw", "nm", "scal"):
        assert np.array_equal(sg[key], so[key]), key
    gg, go = h.get_gbar(chain), o.get_gbar()
    assert np.array_equal(gg[0], go[0]) and gg[1] == go[1] and gg[2] == go[2]
    rg, ro = h.results(chain), o.results()
    for key in ("chi2_res", "edge_res", "n0_res", "wn0_res", "amp_ar"):
        assert np.array_equal(rg[key], ro[key]), key
    assert rg["violations"] == o.violations() == 0
    assert rg["sweeps"] == o.sweep_counter()
    cnt = o.counters()
    assert (rg["proposals"], rg["evaluated"], rg["accepted"], rg["lookups"]) == (cnt[0], cnt[1], cnt[2], cnt[4])
    for rl in range(n_rl):
        for m in range(2):
            assert np.array_equal(h.accept_rates(m, rl, chain), o.accept_rates(m, rl))
    wa_g, aa_g = h.avg(chain)
    wa_o, aa_o = o.avg()
    assert np.array_equal(wa_g, wa_o) and np.array_equal(aa_g, aa_o)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_edge_chain_matches_oracle_move_for_move(case):
    name, run, dw, n_e, n_c, A_c, p, stages = case
    seed = 4242
    tin, tab, init, o = make_chain(run, dw, n_e, n_c, A_c, p, seed=seed, chain=0)
    h = make_handle(tin, tab, init, 1, seed, sweeps_per_launch=5)
    # initialisation: Taylor lookup of every delta, Gbar, chi2
    g0, o0 = h.get_gbar(0), o.get_gbar()
    assert np.array_equal(g0[0], o0[0]) and g0[1] == o0[1]
    cap = 400000
    h.set_move_log(0, cap)
    o.enable_log(cap)
    for steps, bins, theta in stages:
        h.run_bins(steps, bins, theta)
        o.run_bins(steps, bins, theta)
        lg, n = h.move_log(cap)
        lo = o.log()
        assert n == len(lo) <= cap
        bad = np.flatnonzero((lg["move"] != lo["move"]) | (lg["flags"] != lo["flags"]) | (lg["chi2_new"] != lo["chi2_new"]))
        assert len(bad) == 0, (bad[:5], lg[bad[:3]], lo[bad[:3]])
        compare(h, o, 0, tab.n_rl)
        for rl in range(tab.n_rl):
            for which in range(2):
                assert np.array_equal(h.histogram(which, rl), o.spec(which, rl))
    acc = (lo["flags"] & 2) > 0
    assert acc.any() and not acc.all()
    h.close()


def test_large_clusters_register_and_lds_paths():
    """Cluster sizes from 1 to 100 deltas: the register-resident path (n <= 64, several pair blocks and column
    batches) and the LDS path for longer clusters, both against the oracle."""
    seed = 77
    tin, tab, init, o = make_chain("hchain_single", 1e-3, 150, seed=seed, chain=0)
    NE = init.n_e
    nm = init.nm.copy()
    for i in range(2, NE):                                   # Julia index i, array row i - 1
        nm[i - 1, 0] = max(1, min(100, NE - i - 1, 3 + (i * 7) % 97))
    nm[1, 0] = 100                                           # the edge cluster of multi_1
    o.set_state(init.w, init.A, init.dw, nm, init.scalars())
    o.init_gbar()
    h = make_handle(tin, tab, init, 1, seed, sweeps_per_launch=3)
    h.set_state(init.w, init.A, init.dw, nm, init.scalars())
    h.init_gbar()
    cap = 200000
    h.set_move_log(0, cap)
    o.enable_log(cap)
    for steps, bins, theta in [(4, 2, 20.0), (4, 2, 2.0)]:
        h.run_bins(steps, bins, theta)
        o.run_bins(steps, bins, theta)
        lg, n = h.move_log(cap)
        lo = o.log()
        assert n == len(lo) <= cap
        bad = np.flatnonzero((lg["move"] != lo["move"]) | (lg["flags"] != lo["flags"]) | (lg["chi2_new"] != lo["chi2_new"]))
        assert len(bad) == 0, (bad[:5], lg[bad[:3]], lo[bad[:3]])
        compare(h, o, 0, tab.n_rl)
    multi = lo[(lo["move"] == 43)]
    assert ((multi["flags"] & 1) > 0).any()                  # cluster moves were evaluated ...
    assert ((multi["flags"] & 2) > 0).any()                  # ... and some accepted
    h.close()


def test_chains_are_independent_of_their_neighbours():
    tin, tab, init, _ = make_chain("t2_symm", 2e-3, 24, seed=9)
    h = make_handle(tin, tab, init, 70, 9, sweeps_per_launch=7)
    h.run_bins(10, 3, 5.0)
    for chain in (0, 37, 69):
        o = make_chain("t2_symm", 2e-3, 24, seed=9, chain=chain)[3]
        o.run_bins(10, 3, 5.0)
        compare(h, o, chain, 1)
    merged = h.results(-1)
    per = [h.results(c) for c in range(70)]
    assert merged["chi2_min"] == min(r["chi2_min"] for r in per)
    assert merged["chi2_res"][1] == pytest.approx(np.mean([r["chi2_res"][1] for r in per]), rel=1e-12)
    assert merged["violations"] == 0
    h.close()


def test_error_paths():
    tin, tab, init, _ = make_chain("t2_symm", 2e-3, 24)
    with pytest.raises(SacError):
        SacEdgeHandle(tin["n_tau"], 3, 3, tab.n_grid, tab.wi0, tab.dw, init.w_m, init.w_floor, init.c)
    h = SacEdgeHandle(tin["n_tau"], init.n_e, init.n_omega, tab.n_grid, tab.wi0, tab.dw, init.w_m, init.w_floor, init.c)
    with pytest.raises(SacError):
        h.run_bins(10, 2, 1.0)                 # nothing uploaded
    h.upload_problem(tab.full(), tin["G_D"], tin["sigma_inv"])
    bad = init.nm.copy()
    bad[2, 0] = 1000
    with pytest.raises(SacError):
        h.set_state(init.w, init.A, init.dw, bad, init.scalars())
    h.set_state(init.w, init.A, init.dw, init.nm, init.scalars())
    with pytest.raises(SacError):
        h.run_bins(10, 2, 1.0)                 # Gbar not initialised
    h.init_gbar()
    with pytest.raises(SacError):
        h.run_bins(10, 1, 1.0)                 # bins >= 2 (sigma / sqrt(bins - 1))
    h.run_bins(4, 2, 1.0)
    h.close()
