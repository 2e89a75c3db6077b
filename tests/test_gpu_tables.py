"""K tables built on the device (elementwise kernel + basis change by fp64 MFMA, edge
stencils) against the host-built tables of the reference's recipe (free:199-213, peak:248-276, edge:167-236).

The two are not bit-identical (other exp, other summation order): K agrees to 1e-13 of the row scale -- the
north-star tolerance on a move's delta chi^2 is 1e-12 -- and everything downstream of the table is checked
exactly on the table the device holds (read back through the ABI): the edge stencils, and a chain run on the
device-built table against the oracle run on that same table."""
import numpy as np
import pytest

from oracle import oracle as ora
from tests import helpers
from tests.test_oracle_edge import load_run

pytestmark = pytest.mark.gpu

TOL = 1e-13


def _close(dev, host):
    scale = np.abs(host).max(axis=1, keepdims=True) + 1e-300
    return float((np.abs(dev - host) / scale).max())


@pytest.mark.parametrize("name,kw", [
    ("bosonic_ntau22", dict(kind="golden")),
    ("bosonic_ntau64", dict(kind="synth", n_tau=64, n_omega=300, omega_m=10.0, d_omega=1e-3)),
    ("finiteT_ntau80_negative_grid", dict(kind="synth", n_tau=80, n_omega=100, omega_m=6.0, d_omega=2e-3, kernel_type="finiteT", omega_0=-4.0)),
    ("finiteT_symm_ntau37", dict(kind="synth", n_tau=37, n_omega=100, omega_m=6.0, d_omega=2e-3, kernel_type="finiteT", symm=1)),
    ("zeroT_ntau100", dict(kind="synth", n_tau=100, n_omega=100, omega_m=5.0, d_omega=2e-3, kernel_type="zeroT")),
])
def test_free_table_built_on_device(name, kw):
    kw = dict(kw)
    kind = kw.pop("kind")
    prob = helpers.problem_golden_bosonic() if kind == "golden" else helpers.problem_synth(**kw)
    h = prob.make_handle(n_chains=2, seed=3, device_tables=True)
    dev = h.get_table(0)
    rows = prob.wi_m - prob.wi_0 + 1                       # rows the reference fills (the rest is never addressed)
    assert _close(dev[:rows], prob.K[:rows]) < TOL
    # a chain on the device-built table == the oracle on that very table
    o = ora.OracleFree(dev, prob.td.G, prob.td.sigma_inv, prob.n_omega, prob.wi_0, prob.wi_m, par=prob.par, symm=prob.symm)
    o.set_state(prob.wi, prob.A, np.full(4, prob.window))
    o.init_gbar()
    o.use_philox(3, 1)
    h.adjust_windows(20, 2.0)
    o.adjust_windows(20, 2.0)
    h.sample(6, 2.0)
    o.sample(6, 2.0)
    wi_g, _, win_g = h.get_state(1)
    wi_o, _, win_o = o.get_state()
    assert np.array_equal(wi_g, wi_o) and np.array_equal(win_g, win_o)
    g, c2, c2m = h.get_gbar(1)
    go, c2o, c2mo = o.get_gbar()
    assert np.array_equal(g[:prob.td.n_tau], go[:prob.td.n_tau]) and c2 == c2o and c2m == c2mo
    h.close()


def test_peak_tables_built_on_device():
    prob = helpers.problem_peak(kind="synth", n_omega=80, n_p=1, A_0=0.5, omega_m=6.0, d_omega=2e-3,
                                kernel_type="finiteT", approx_ratio=2.5, n_tau=48)
    assert prob.sample_neg
    h = prob.make_handle(n_chains=1, seed=5, device_tables=True)
    assert _close(h.get_table(0), prob.Kp) < TOL
    assert _close(h.get_table(1), prob.Kn) < TOL
    h.adjust_windows(10, 5.0)                               # the stage runs on the built tables
    assert np.isfinite(h.peak_stats().chi2_min)
    h.close()


@pytest.mark.parametrize("run,n_e", [("hchain_single", 30), ("t3_in", 20), ("t2_symm", 24)])
def test_edge_tables_built_on_device(run, n_e):
    tin, tab, init = load_run(run, dw=2e-3, n_e=n_e)
    init.config_dual()
    from tests.test_gpu_edge import make_handle
    h = make_handle(tin, tab, init, 1, 11)                  # host tables first (the usual path) ...
    host = [h.get_tables(rl) for rl in range(tab.n_rl)]
    for rl in range(tab.n_rl):                              # ... then rebuilt on the device
        h.build_tables(rl, tab.kernel_type, tin["tau"], tin["beta"], tin["cov"], tab.wi0, tab.dw,
                       mirror=int(tab.mode == "double_edge_symm"), negate=int(rl == 1))
    h.upload_data(tin["G_D"], tin["sigma_inv"])
    for rl in range(tab.n_rl):
        K, dK, d2K = h.get_tables(rl)
        assert _close(K, host[rl][0]) < TOL
        # the stencils, exactly, on the device's own K (edge:230-235)
        n, dw = tab.n_grid, tab.dw
        e1 = np.zeros_like(K)
        e2 = np.zeros_like(K)
        e1[1:n - 1] = (K[2:] - K[:n - 2]) / (2 * dw)
        e2[1:n - 1] = (K[2:] - (2 * K[1:n - 1]) + K[:n - 2]) / (2 * dw ** 2)
        e1[0] = (K[1] - K[0]) / dw
        wim = tab.wi0 + n - 1
        e1[wim] = (K[wim] - K[wim - 1]) / dw
        assert np.array_equal(dK, e1) and np.array_equal(d2K, e2)
    h.init_gbar()
    h.run_bins(4, 2, 5.0)
    assert h.results(0)["violations"] == 0
    h.close()
