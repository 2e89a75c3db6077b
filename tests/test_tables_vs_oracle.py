"""The product's host-built problem (sac_amd/{free,peak,edge}.py: grid bounds, initial ladder, amplitudes, windows,
K tables in the covariance eigenbasis) against the oracle's independent restatement of the same reference code
(oracle/oracle.py: initialize free:134-218, initialize_peak peak:149-285, EdgeTables edge:167-242) on the golden
t.in files -- and both against a scalar evaluation of a few table columns written out from the formula
K[:, i] = transpose(cov) * kernel(omega_i, tau) (free:199-213).  Every move-parity test hands the oracle the product's
tables; this closes the hole a wrong table build (cov orientation, mirror, omega_0 < 0 offset) would slip through.
CPU only."""
import math
import os

import numpy as np
import pytest

from oracle import oracle as ora
from sac_amd.edge import EdgeProblem
from sac_amd.free import FreeProblem
from sac_amd.peak import PeakProblem
from sac_amd.tin import read_tin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def scalar_kernel(kind, w, t, beta):
    """free:220-239, one (omega, tau) at a time with math.exp."""
    if kind == "zeroT":
        return math.exp(-w * t)
    if kind == "bosonic":
        return (math.exp(-w * t) + math.exp(-w * (beta - t))) / (1.0 + math.exp(-beta * w))
    a, b = -w * t, -beta * w
    m = max(a, b, 0.0)
    return math.exp(a - m) / (math.exp(-m) + math.exp(b - m))


def scalar_column(tin, kind, w, mirror=False):
    """transpose(cov) * K[:, i] written out: out[t] = sum_j cov[j, t] * kernel(w, tau_j)."""
    n = tin["n_tau"]
    k = [scalar_kernel(kind, w, tin["tau"][j], tin["beta"]) + (scalar_kernel(kind, -w, tin["tau"][j], tin["beta"]) if mirror else 0.0)
         for j in range(n)]
    return np.array([sum(tin["cov"][j, t] * k[j] for j in range(n)) for t in range(n)])


def assert_table(got, want, what):
    assert got.shape == want.shape, what
    scale = np.abs(want).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(got - want) / scale).max() <= 1e-13, what


def test_timedata_reader_matches_oracle():
    for path in (f"{GOLD}/cv/t_001a.in", f"{GOLD}/peak_f2/t.in", f"{GOLD}/edge/t1_single/t.in"):
        td, t = read_tin(path), ora.read_tin(path)
        assert (td.beta, td.n_tau, td.norm) == (t["beta"], t["n_tau"], t["norm"])
        for a, b in ((td.tau, t["tau"]), (td.G_raw, t["G"]), (td.cov, t["cov"]), (td.sigma_inv, t["sigma_inv"])):
            assert np.array_equal(a, b)
        np.testing.assert_allclose(td.G, t["G_D"], rtol=0, atol=1e-15 * np.abs(t["G"]).max())
        assert td.omega_window == pytest.approx(t["omega_window"], rel=1e-15)


@pytest.mark.parametrize("tin_file,kind,symm,w0,wm,dw,par", [
    ("cv/t_001a.in", "bosonic", 0, 0.0, 4.0, 2e-3, 1),
    ("peak_f2/t.in", "finiteT", 1, -3.0, 5.0, 5e-3, 3),        # symm: omega_0 forced to 0, K(w) + K(-w), A sums to 1/2
    ("peak_f2/t.in", "finiteT", 0, -3.0, 5.0, 5e-3, 2),        # omega_0 < 0: table starts at wi_0, ladder offset
    ("edge/t1_single/t.in", "zeroT", 0, -1.0, 3.0, 4e-3, 1),
])
def test_free_problem_equals_oracle_initialize(tin_file, kind, symm, w0, wm, dw, par):
    path = f"{GOLD}/{tin_file}"
    prob = FreeProblem(read_tin(path), 40, w0, wm, dw, par=par, symm=symm, kernel_type=kind)
    tin = ora.read_tin(path)
    init = ora.initialize(tin, 40, w0, wm, dw, par=par, symm=symm, kernel_type=kind)
    assert (prob.wi_0, prob.wi_m, prob.n_grid) == (init["wi0"], init["wim"], init["n_grid"])
    assert np.array_equal(prob.wi, init["wi"])
    np.testing.assert_allclose(prob.A, init["A"], rtol=1e-15)
    assert prob.A.sum() == pytest.approx(0.5 if symm == 1 else 1.0, rel=1e-14)
    assert prob.window == pytest.approx(init["window"], rel=1e-15)
    assert_table(prob.K, init["K"], "K")
    for col in (0, 1, prob.n_grid // 3, prob.n_grid - 1):       # scalar formula, independent of both numpy builds
        want = scalar_column(tin, kind, (prob.wi_0 + col) * dw, mirror=(symm == 1))
        np.testing.assert_allclose(prob.K[col], want, rtol=0, atol=1e-13 * np.abs(want).max())


@pytest.mark.parametrize("kind,symm,fix_edge", [("finiteT", 0, 0), ("finiteT", 1, 0), ("bosonic", 0, 0), ("finiteT", 0, 1)])
def test_peak_problem_equals_oracle_initialize(kind, symm, fix_edge):
    path = f"{GOLD}/peak_f2/t.in" if kind == "finiteT" else f"{GOLD}/cv/t_001a.in"
    prob = PeakProblem(read_tin(path), 50, 2, 0.7, 6.0, 4e-3, fix_edge=fix_edge, symm=symm, kernel_type=kind)
    tin = ora.read_tin_peak(path, symm=symm, kernel_type=kind)
    init = ora.initialize_peak(tin, 50, 2, 0.7, 6.0, 4e-3, fix_edge=fix_edge, symm=symm, kernel_type=kind)
    assert (prob.wi_0, prob.wi_m, prob.n_grid) == (init["wi0"], init["wim"], init["n_grid"])
    assert prob.approx_ratio == tin["approx_ratio"]
    assert np.array_equal(prob.wi, init["wi"])
    np.testing.assert_allclose(prob.A, init["A"], rtol=1e-15, atol=0)
    assert prob.window == pytest.approx(init["window"], rel=1e-15)
    assert_table(prob.Kp, init["Kp"], "Kp")
    if prob.sample_neg:
        assert_table(prob.Kn, init["Kn"], "Kn")
    else:
        assert prob.Kn is None and not init["Kn"].any()
    for col in (0, 7, prob.n_grid - 1):                         # half-offset grid (peak:253)
        w = (col + 0.5) * 4e-3
        want = scalar_column(tin, kind, w, mirror=(symm == 1))
        np.testing.assert_allclose(prob.Kp[col], want, rtol=0, atol=1e-13 * np.abs(want).max())
        if prob.sample_neg:
            want = scalar_column(tin, kind, -w)
            np.testing.assert_allclose(prob.Kn[col], want, rtol=0, atol=1e-13 * np.abs(want).max())


@pytest.mark.parametrize("case,mode,kind,w0,wm,dw", [
    ("t2_symm", "double_edge_symm", "finiteT", 0.0, 6.0, 4e-3),
    ("t1_single", "single_edge", "zeroT", -2.0, 4.0, 4e-3),    # omega_0 < 0: the row edge:235 addresses is interior
    ("t3_in", "double_edge_in", "finiteT", -3.0, 3.0, 5e-3),
])
def test_edge_problem_equals_oracle_tables(case, mode, kind, w0, wm, dw):
    path = f"{GOLD}/edge/{case}/t.in"
    prob = EdgeProblem(read_tin(path), mode, kind, 24, 0, 0.5, 0.0, 0.5 if mode == "double_edge_in" else 1.0, w0, wm, dw)
    tin = ora.read_tin_edge(path)
    tab = ora.EdgeTables(tin, mode, kind, w0, wm, dw)
    k = prob.kernel
    assert (k.wi0, k.wim, k.n_grid, k.n_rl) == (tab.wi0, tab.wim, tab.n_grid, tab.n_rl)
    assert prob.edge_guess == pytest.approx(tin["edge_guess"], rel=1e-15)
    for rl, (got, want) in enumerate(zip(k.tables, tab.full())):
        assert_table(got[0], want[0], f"K side {rl}")
        # the stencils divide differences of neighbouring columns by dw, dw^2: compare at the scale of K / dw^n
        for n, name in ((1, "dK"), (2, "d2K")):
            scale = np.abs(want[0]).max() / dw ** n
            assert np.abs(got[n] - want[n]).max() <= 1e-12 * scale, f"{name} side {rl}"
        for col in (0, k.n_grid // 2, k.n_grid - 1):
            w = (k.wi0 + col) * dw
            want_col = scalar_column(tin, kind, w if rl == 0 else -w, mirror=(mode == "double_edge_symm"))
            np.testing.assert_allclose(got[0][col], want_col, rtol=0, atol=1e-13 * np.abs(want_col).max())
    # the initial configuration of the deterministic scan (edge:487-583) agrees too
    init = ora.EdgeInit(tin, tab, 24, 0, 0.5, 0.0, 0.5 if mode == "double_edge_in" else 1.0, w0, wm)
    j, kk, c2 = init.config_dual()
    assert (prob.j_opt, prob.k_opt) == (j, kk) and prob.chi2_init == pytest.approx(c2, rel=1e-9)
    np.testing.assert_allclose(prob.w, init.w, rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(prob.A, init.A, rtol=1e-14, atol=0)
    np.testing.assert_allclose(prob.dwin, init.dw, rtol=1e-14, atol=1e-14)
    assert np.array_equal(prob.nm, init.nm)
