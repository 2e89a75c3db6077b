"""Pin the oracle against the reference's own artifacts.

calc_cv.jl (reference, sac/cross_validation/calc_cv.jl:29-58,82-107) scores every row of a
committed GSAC.csv against the *other* half's t.in.  Its committed output free_v.csv is a
deterministic known answer for: the t.in parser, U^T G, U^T G_SAC and the eigenbasis chi^2.
"""
import os

import numpy as np
import pytest

from oracle import oracle as ora

CV = os.path.join(os.path.dirname(__file__), "golden", "cv")


def _expected():
    rows = np.loadtxt(os.path.join(CV, "free_v_expected.csv"), delimiter=",", ndmin=2)
    return rows[0], rows[1:]


@pytest.mark.parametrize("rep", [1, 2])
def test_calc_cv_known_answer(rep):
    theta_exp, v_exp = _expected()
    # calc_cv.jl:83-93: GSAC(b) against t_a, GSAC(a) against t_b, averaged (calc_cv.jl:98)
    th_ab, v_ab = ora.calc_cv(f"{CV}/t_{rep:03d}a.in", f"{CV}/GSAC_free_{rep:03d}b.csv")
    th_ba, v_ba = ora.calc_cv(f"{CV}/t_{rep:03d}b.in", f"{CV}/GSAC_free_{rep:03d}a.csv")
    v = 0.5 * (v_ab + v_ba)
    n = min(len(v), v_exp.shape[1])
    assert n >= 100
    # cancellation in (G - Gbar)/sigma with sigma ~ 1e-7 sets a ~1e-10 floor
    np.testing.assert_allclose(v[:n], v_exp[rep - 1][:n], rtol=1e-9)


def test_tin_reader_fields():
    t = ora.read_tin(f"{CV}/t_001a.in")
    assert t["n_tau"] == 22 and t["beta"] == 2048.0 and t["n_boot"] == 10000.0
    assert abs(t["norm"] - 0.31828666324499966) < 1e-18
    U = t["cov"]
    assert np.abs(U.T @ U - np.eye(22)).max() < 1e-12      # orthonormal eigenvectors
    assert t["tau"][0] == 0.1 and t["G"][0] == 0.8943318941497335
    assert t["sigma"][0] == 6.493148619210861e-7           # column 4, not 3 (free:109)


def test_kernel_identities():
    tau = np.array([0.0, 0.3, 1.7])
    # finiteT_K(0, tau, beta) = 1/2, bosonic_K(0, tau, beta) = 1, zeroT_K(0, ...) = 1
    assert np.all(ora.kernel_values("finiteT", [0.0], tau, 8.0) == 0.5)
    assert np.all(ora.kernel_values("bosonic", [0.0], tau, 8.0) == 1.0)
    assert np.all(ora.kernel_values("zeroT", [0.0], tau, 8.0) == 1.0)
    # finiteT: K(w) = e^{-w tau}/(1+e^{-beta w}), stable for large |w|
    k = ora.kernel_values("finiteT", [-400.0, 400.0], np.array([1.0]), 4.0)
    assert np.isfinite(k).all()
    np.testing.assert_allclose(k[0, 0], np.exp(-400.0 * 3.0), rtol=1e-12, atol=0)
    np.testing.assert_allclose(k[1, 0], np.exp(-400.0), rtol=1e-12, atol=0)


def test_initialize_ladder_and_gbar():
    t = ora.read_tin(f"{CV}/t_001a.in")
    init = ora.initialize(t, n_omega=100, omega_0=0.0, omega_m=5.0, d_omega=1e-3,
                          par=1, symm=0, kernel_type="bosonic")
    assert init["wi0"] == 0 and init["wim"] == 5000 and init["n_grid"] == 5001
    assert init["wi"][0] == 50 and init["wi"][-1] == 5000        # free:166
    assert abs(init["A"].sum() - 1.0) < 1e-14
    o = ora.OracleFree(init["K"], t["G_D"], t["sigma_inv"], 100, init["wi0"], init["wim"])
    o.set_state(init["wi"], init["A"], np.full(4, init["window"]))
    o.init_gbar()
    g, c, cmin = o.get_gbar()
    ref = init["K"][init["wi"] - init["wi0"]].T @ init["A"]
    np.testing.assert_allclose(g, ref, rtol=1e-12)
    assert c == cmin == ora.chi2(g, t["G_D"], t["sigma_inv"])
