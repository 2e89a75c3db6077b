"""Edge oracle: (1) the deterministic initial scan against the (j_opt, k_opt) and edge endpoints the
reference logged for its committed runs (known answers: read_G! edge_guess, kernel tables with
derivative stencils, set_amplitudes!, init_config_jk!, Taylor lookup, chi^2, 50x50 argmin; no RNG);
(2) invariants of the C restatement of the moves."""
import os
import re

import numpy as np
import pytest

from oracle import oracle as ora

EDGE = os.path.join(os.path.dirname(__file__), "golden", "edge")


def load_run(name, dw=None, n_e=None, n_c=None, A_c=None, p=None):
    rows = [ln.split() for ln in open(f"{EDGE}/{name}/in_edge.in")]
    w0, wm, dwh, dw_file = (float(x) for x in rows[1][:4])  # edge:1888 (this order)
    p_file, A_c_file, A_r = (float(x) for x in rows[2][:3])
    n_e = int(rows[0][0]) if n_e is None else n_e
    n_c = int(rows[0][1]) if n_c is None else n_c
    dw = dw_file if dw is None else dw
    A_c = A_c_file if A_c is None else A_c
    p = p_file if p is None else p
    fix_edge, ktype, mode = float(rows[6][0]), rows[6][1], rows[7][0]
    if mode == "single_edge":                               # edge:1909-1921
        A_r = 1.0
    elif mode == "double_edge_out":
        w0 = 0.0
    elif mode == "double_edge_symm":
        w0, A_r = 0.0, 0.5
    if ktype == "bosonic":
        w0 = 0.0
    tin = ora.read_tin_edge(f"{EDGE}/{name}/t.in")
    tab = ora.EdgeTables(tin, mode, ktype, w0, wm, dw)
    init = ora.EdgeInit(tin, tab, n_e, n_c if A_c > 0 else 0, p, A_c, A_r, w0, wm, fix_edge_value=fix_edge)
    return tin, tab, init


def logged(name):
    txt = open(f"{EDGE}/{name}/init_edges.txt").read()
    m = re.search(r"Rightward Edge: \[([-\d.e]+), ([-\d.e]+)\] \((\d+), (\d+)\)", txt)
    right = (float(m.group(1)), float(m.group(2)), int(m.group(3)), int(m.group(4)))
    m = re.search(r"Leftward Edge: \[([-\d.e]+), ([-\d.e]+)\]", txt)
    left = (float(m.group(1)), float(m.group(2))) if m else None
    return right, left


@pytest.mark.parametrize("name", ["hchain_single", "t1_single", "t2_symm", "t3_in", "t4_out", "tj_in_05", "tj_in_07"])
def test_initial_scan_matches_reference_log(name):
    tin, tab, init = load_run(name)
    j_opt, k_opt, chi2 = init.config_dual()
    (w_first, w_last, j_ref, k_ref), left = logged(name)
    assert (j_opt, k_opt) == (j_ref, k_ref)
    assert init.w[1, 0] == pytest.approx(w_first, abs=1e-12)
    assert init.w[init.n_e, 0] == pytest.approx(w_last, abs=1e-12)
    if left is not None:
        assert -init.w[1, 1] == pytest.approx(left[0], abs=1e-12)
        assert -init.w[init.n_e, 1] == pytest.approx(left[1], abs=1e-12)


# ---- the moves (C restatement) --------------------------------------------------------------------

def make_chain(name, dw, n_e, n_c=0, A_c=0.0, p=0.5, seed=5, chain=0):
    tin, tab, init = load_run(name, dw=dw, n_e=n_e, n_c=n_c, A_c=A_c, p=p)
    init.config_dual()
    o = ora.OracleEdge(tab.full(), tin["G_D"], tin["sigma_inv"], init.n_e, init.n_omega, tab.wi0, tab.dw, init.w_m,
                       init.w_floor, init.c, A_c=init.A_c, A_r=init.A_r, double_in=int(init.mode == "double_edge_in"))
    o.set_state(init.w, init.A, init.dw, init.nm, init.scalars())
    o.init_gbar()
    o.use_philox(seed, chain)
    return tin, tab, init, o


def check_invariants(tin, tab, init, o, steps, bins):
    assert o.violations() == 0
    st = o.get_state()
    NE, N = init.n_e, init.n_omega
    for rl in range(tab.n_rl):
        w = st["w"][1:NE + 1, rl]
        d = np.diff(w)
        assert (d >= -1e-12).all() and (np.diff(d) >= -1e-10).all()      # ordered, non-decreasing spacings
        assert w.max() <= init.w_m
        f1 = (init.A_r if rl == 0 else 1 - init.A_r) / (1 + init.A_c)
        assert st["A"][1:NE + 1, rl].sum() == pytest.approx(f1, rel=1e-12)
    if N > NE:
        assert st["A"][NE + 1:, 0].sum() == pytest.approx(init.A_c / (1 + init.A_c), rel=1e-10)
        wc = st["w"][NE + 1:, 0]
        assert wc.min() >= st["w"][1, 0] and wc.max() <= st["w"][NE, 0]
    # running Gbar / chi2 against a from-scratch evaluation of the final configuration
    g, c2, c2min = o.get_gbar()
    g0 = tab.taylor(0, st["w"][:, 0]).T @ st["A"][:, 0]
    if tab.n_rl == 2:
        g0 = g0 + tab.taylor(1, st["w"][:, 1]).T @ st["A"][:, 1]
    assert np.abs(g - g0).max() < 1e-11
    assert c2 == pytest.approx(float((((g0 - tin["G_D"]) * tin["sigma_inv"]) ** 2).sum()), rel=1e-6)
    assert c2min <= c2
    # histogram mass = sweeps x total amplitude
    for rl in range(tab.n_rl):
        tot = o.spec(0, rl).sum() + o.spec(1, rl).sum()
        assert tot == pytest.approx(steps * bins * st["A"][1:, rl].sum(), rel=1e-9)


@pytest.mark.parametrize("name,n_e,n_c,A_c,p", [("t2_symm", 24, 0, 0.0, 0.5), ("hchain_single", 30, 0, 0.0, 0.4),
                                                ("t4_out", 20, 0, 0.0, 0.5), ("t3_in", 20, 0, 0.0, 0.3),
                                                ("t1_single", 24, 12, 0.3, 0.6)])
def test_moves_keep_the_invariants(name, n_e, n_c, A_c, p):
    tin, tab, init, o = make_chain(name, 2e-3, n_e, n_c, A_c, p)
    chi2_0 = o.get_gbar()[1]
    o.enable_log(200000)
    for theta in (10.0, 1.0):
        o.run_bins(40, 4, theta)
    check_invariants(tin, tab, init, o, 40, 4)
    log = o.log()
    acc = (log["flags"] & 2) > 0
    moves = set(np.unique(log["move"]))
    assert {41, 43, 47} <= moves
    if A_c > 0:
        assert {49, 50} <= moves
    if tab.n_rl == 2:
        assert {42, 44, 48} <= moves
    assert acc.any() and not acc.all()
    assert o.get_gbar()[2] < chi2_0                      # the anneal found something better than the start
    r = o.results()
    assert r["chi2_res"][1] > 0 and r["chi2_res"][2] >= 0
    assert o.sweep_counter() == 2 * 160


def test_chain_is_reproducible_and_seed_dependent():
    a = make_chain("t2_symm", 2e-3, 24)[3]
    b = make_chain("t2_symm", 2e-3, 24)[3]
    c = make_chain("t2_symm", 2e-3, 24, seed=6)[3]
    for o in (a, b, c):
        o.run_bins(20, 3, 2.0)
    assert np.array_equal(a.get_state()["w"], b.get_state()["w"])
    assert not np.array_equal(a.get_state()["w"], c.get_state()["w"])


def test_sac_log_exp_accuracy():
    import math
    rng = np.random.default_rng(3)
    for x in np.concatenate([rng.uniform(1e-3, 80.0, 2000), rng.uniform(0.9, 1.1, 500)]):
        ref = math.log(x)
        assert abs(ora.sac_log(x) - ref) <= 2 * np.spacing(abs(ref)) + 1e-300
    for x in np.concatenate([rng.uniform(-20.0, 5.0, 2000), rng.uniform(-0.4, 0.4, 500)]):
        ref = math.exp(x)
        assert abs(ora.sac_exp(x) - ref) <= 2 * np.spacing(ref)


def test_oracle_follows_the_reference_run_statistically():
    """t2_double_symm with the reference's own equilibration (5 x 600 sweeps x 20 bins at Theta = 10) and its
    first anneal stage, on a 100x coarser kernel grid: the committed anneal.csv of the real Julia run says
    chi2_min 0.861, <chi2> 2.73 +- 0.07, edge 0.649, n0/N_e 0.499 (rows j = 0, 1)."""
    tin, tab, init, o = make_chain("t2_symm", 1e-3, 80, seed=5)
    nt = tin["n_tau"]
    ref = np.genfromtxt(f"{EDGE}/t2_symm/anneal.csv", delimiter=",", skip_header=1)
    o.run_bins(3000, 20, 10.0)
    r = o.results()
    assert o.get_gbar()[2] / nt == pytest.approx(ref[0, 2], rel=0.1)          # chi2_min / N_tau
    assert r["edge_res"][1, 0] == pytest.approx(ref[0, 5], abs=0.03)
    o.run_bins(588, 20, 10.0)
    r = o.results()
    assert r["chi2_res"][1] / nt == pytest.approx(ref[1, 3], rel=0.15)        # <chi2> / N_tau
    assert r["edge_res"][1, 0] == pytest.approx(ref[1, 5], abs=0.015)
    assert r["n0_res"][1, 0] == pytest.approx(ref[1, 7], abs=0.05)
    assert o.violations() == 0
