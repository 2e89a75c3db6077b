"""Size-independent properties at BASELINE.json's full sizes for the peak and edge parameterisations
(configs[3]: 10^6-column table; configs[2]: edge with thousands of chains) -- where the oracle would take
hours: conservation laws, ordering invariants, histogram mass, Gbar/chi2 from scratch, chain independence."""
import numpy as np
import pytest

from sac_amd.synth import synthetic_timedata

pytestmark = pytest.mark.gpu


def test_peak_million_column_table_4096_chains():
    from sac_amd.peak import PeakProblem
    td = synthetic_timedata(n_tau=64, seed=777)
    prob = PeakProblem(td, 1000, 1, 0.5, 10.0, 1e-5, kernel_type="bosonic")
    assert prob.n_grid == 1000001
    h = prob.make_handle(n_chains=4096, seed=777)
    h.adjust_windows(50, 1.0)
    steps = 20
    h.sample(steps, 1.0)
    st = h.peak_stats()
    assert st.sweeps == 70 * 4096 and st.evaluated <= st.proposals and 0 < st.accepted < st.evaluated
    assert abs(sum(st.weights[:2]) - 1.0) < 1e-12                        # peak + continuum weight (one axis)
    mass = sum(h.histogram_peak(w).sum() for w in range(4))              # chain- and sweep-averaged
    assert mass == pytest.approx(1.0, rel=1e-9)
    for c in (0, 1777, 4095):
        wi, A, _ = h.get_state(c)
        assert A.sum() == pytest.approx(1.0, rel=1e-12) and wi.min() >= 0 and wi.max() <= prob.wi_m
        b = h.bounds(c)
        assert b[0] == wi[:1].max() and b[1] == wi[1:1001].min()
        g, chi2, cmin = h.get_gbar(c)
        g0 = prob.Kp[wi[:1001]].T @ A[:1001]
        # Gbar is refreshed from scratch every sweep, then updated incrementally within the last one
        assert np.abs(g - g0).max() < 1e-12 and cmin <= chi2
    # chains differ (independent streams) but share the statistics' scale
    c2, _ = h.chain_chi2()
    assert len(np.unique(c2)) > 4000 and c2.std() / c2.mean() < 1.0
    h.close()


def test_edge_million_column_tables_3072_chains():
    from sac_amd.edge import EdgeProblem
    td = synthetic_timedata(n_tau=64, seed=777)
    prob = EdgeProblem(td, "single_edge", "bosonic", 80, 0, 0.4, 0.0, 1.0, 0.0, 10.0, 1e-5)
    assert prob.kernel.n_grid == 1000001
    nc, steps, bins = 3072, 20, 3
    h = prob.make_handle(n_chains=nc, seed=777)
    h.run_bins(steps, bins, 2.0)
    r = h.results(-1)
    assert r["violations"] == 0 and r["sweeps"] == steps * bins
    assert r["chi2_res"][1] > 0 and r["chi2_res"][2] >= 0 and r["chi2_min"] <= r["chi2"]
    mass = h.histogram(0, 0).sum() + h.histogram(1, 0).sum()              # chain mean of the accumulated sums
    assert mass == pytest.approx(steps * bins * 1.0, rel=1e-9)
    w_avg, A_avg = h.avg(-1)
    assert A_avg[1:, 0].sum() == pytest.approx(steps * bins, rel=1e-9)
    for c in (0, 1500, nc - 1):
        s = h.get_state(c)
        w = s["w"][1:81, 0]
        d = np.diff(w)
        assert (d >= 0).all() and (np.diff(d) >= -1e-10).all() and w[0] >= 0 and w[-1] <= 10.0
        assert s["A"][1:81, 0].sum() == pytest.approx(1.0, rel=1e-12)
        assert (s["nm"][1:79, 0] >= 1).all() and (s["nm"][1:79, 0] <= 80 - np.arange(2, 80)).all()
        g, chi2, _ = h.get_gbar(c)
        g0 = prob.kernel.lookup(0, s["w"][:, 0]).T @ s["A"][:, 0]
        assert np.abs(g - g0).max() < 1e-11
        assert chi2 == pytest.approx(float((((g0 - td.G) * td.sigma_inv) ** 2).sum()), rel=1e-6)
    h.close()
