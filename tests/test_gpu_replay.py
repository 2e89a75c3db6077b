"""sac_replay_peak / sac_edge_replay: an explicit proposal stream evaluated by the library agrees with a
direct fp64 evaluation (numpy, the reference's formulas) to <= 1e-12 relative per move -- the north-star tolerance for
per-move chi^2 -- takes the same accept decisions and leaves a consistent chain behind."""
import numpy as np
import pytest

from tests import helpers
from sac_amd.lib import EDGE_PROPOSAL, PEAK_PROPOSAL, SacError

pytestmark = pytest.mark.gpu


def _chi2(g, G, sinv):
    return float((((g - G) * sinv) ** 2).sum())


@pytest.mark.parametrize("kernel_type", ["finiteT", "bosonic"])
def test_peak_replay(kernel_type):
    prob = helpers.problem_peak(kind="synth", n_omega=120, n_p=2, A_0=0.6, omega_m=6.0, d_omega=1e-3, kernel_type=kernel_type,
                                n_tau=48)
    h = prob.make_handle(n_chains=3, seed=5)
    h.adjust_windows(20, 2.0)
    h.init_gbar()                                     # Gbar from scratch, as at the start of a sweep (peak:1072)
    chain, theta = 1, 2.0
    wi, A, _ = h.get_state(chain)
    gbar, chi2, _ = h.get_gbar(chain)
    G, sinv = prob.td.G, prob.td.sigma_inv
    n_axis = prob.n_omega + prob.n_p
    both = prob.Kn is not None
    tabs = [prob.Kp, prob.Kn if both else prob.Kp]
    ref = sum(A[k] * tabs[k >= n_axis][wi[k]] for k in range(n_axis * (2 if both else 1)))
    np.testing.assert_allclose(gbar, ref, rtol=1e-11, atol=1e-15)
    rng = np.random.default_rng(3)
    n = 300
    props = np.zeros(n, dtype=PEAK_PROPOSAL)
    wi_s, A_s, g_s, c_s = wi.copy(), A.copy(), gbar.copy(), chi2
    exp_c, exp_a = np.empty(n), np.empty(n, dtype=np.int32)
    for i in range(n):
        transfer = both and rng.random() < 0.4
        if transfer:                                  # (+ axis, - axis) pairs with weight moving between them
            m = 2 * int(rng.integers(1, 3))
            ks = []
            for q in range(m // 2):
                while True:
                    kp, kn = int(rng.integers(0, n_axis)), int(rng.integers(n_axis, 2 * n_axis))
                    if kp not in ks and kn not in ks:
                        break
                ks += [kp, kn]
        else:
            m = int(rng.integers(1, 4))
            ks = [int(k) for k in rng.choice(n_axis * (2 if both else 1), size=m, replace=False)]
        new = np.clip(wi_s[ks] + rng.integers(-200, 201, size=m), 0, prob.wi_m)
        props[i]["n"], props[i]["transfer"], props[i]["u"] = m, int(transfer), rng.random()
        props[i]["k"][:m], props[i]["w_new"][:m] = ks, new
        gn = g_s.copy()
        An = A_s[ks].copy()
        if transfer:
            for q in range(0, m, 2):
                dA = rng.uniform(-0.5, 0.5) * min(A_s[ks[q]], A_s[ks[q + 1]])
                An[q], An[q + 1] = A_s[ks[q]] + dA, A_s[ks[q + 1]] - dA
                gn = ((((gn + An[q] * tabs[0][new[q]]) + An[q + 1] * tabs[1][new[q + 1]]) - A_s[ks[q]] * tabs[0][wi_s[ks[q]]])
                      - A_s[ks[q + 1]] * tabs[1][wi_s[ks[q + 1]]])
            props[i]["A_new"][:m] = An
        else:
            for e in range(m):
                t = tabs[ks[e] >= n_axis]
                gn = gn + A_s[ks[e]] * (t[new[e]] - t[wi_s[ks[e]]])
        c2 = _chi2(gn, G, sinv)
        acc = props[i]["u"] <= np.exp(min((c_s - c2) / (2 * theta), 50.0))
        exp_c[i], exp_a[i] = c2, acc
        if acc:
            wi_s[ks], g_s, c_s = new, gn, c2
            A_s[ks] = An
    c_gpu, a_gpu = h.replay_peak(props, theta, chain=chain)
    np.testing.assert_allclose(c_gpu, exp_c, rtol=1e-12)
    assert np.array_equal(a_gpu, exp_a) and 0 < a_gpu.sum() < n
    wi_g, A_g, _ = h.get_state(chain)
    assert np.array_equal(wi_g, wi_s)
    np.testing.assert_allclose(A_g, A_s, rtol=1e-15)
    g_g, c_g, _ = h.get_gbar(chain)
    np.testing.assert_allclose(g_g, g_s, rtol=1e-11, atol=1e-15)
    assert c_g == pytest.approx(c_s, rel=1e-12)
    bad = props[:1].copy()
    bad[0]["w_new"][0] = prob.wi_m + 5
    with pytest.raises(SacError):
        h.replay_peak(bad, theta, chain=chain)
    h.sample(5, theta)                                # the chain keeps running from the replayed state
    h.close()


def test_edge_replay():
    from tests.test_gpu_edge import make_handle
    from tests.test_oracle_edge import make_chain
    tin, tab, init, _ = make_chain("hchain_single", 2e-3, 30, 0, 0.0, 0.4, seed=9, chain=0)
    h = make_handle(tin, tab, init, 2, 9)
    h.run_bins(30, 2, 3.0)
    chain, theta = 1, 3.0
    st = h.get_state(chain)
    gbar, chi2, _ = h.get_gbar(chain)
    w, A = st["w"].copy(), st["A"]
    G, sinv = tin["G_D"], tin["sigma_inv"]
    NE = init.n_e
    look = lambda x: tab.taylor(0, np.atleast_1d(x))                   # noqa: E731  (edge:626-633)
    kc = look(w[:, 0])                                                 # K_curr, row r = Julia index r + 1
    np.testing.assert_allclose(gbar, kc.T @ A[:, 0], rtol=1e-10, atol=1e-15)
    rng = np.random.default_rng(4)
    n = 120
    props = np.zeros(n, dtype=EDGE_PROPOSAL)
    w_new = []
    g_s, c_s = gbar.copy(), chi2
    exp_c, exp_a = np.empty(n), np.empty(n, dtype=np.int32)
    for i in range(n):
        m = 1 if rng.random() < 0.5 else int(rng.integers(2, 9))
        k = int(rng.integers(2, NE + 2 - m + 1))                       # Julia index of the first moved delta
        new = w[k - 1:k - 1 + m, 0] + rng.normal(0, 0.01, size=m)
        props[i]["side"], props[i]["k"], props[i]["n"], props[i]["w_offset"], props[i]["u"] = 0, k, m, len(w_new), rng.random()
        w_new += list(new)
        kn = look(new)
        gn = g_s.copy()
        for j in range(m):
            gn = gn + A[k - 1 + j, 0] * (kn[j] - kc[k - 1 + j])
        c2 = _chi2(gn, G, sinv)
        acc = props[i]["u"] < np.exp(min((c_s - c2) / (2 * theta), 50.0))
        exp_c[i], exp_a[i] = c2, acc
        if acc:
            g_s, c_s = gn, c2
            w[k - 1:k - 1 + m, 0] = new
            kc[k - 1:k - 1 + m] = kn
    c_gpu, a_gpu = h.replay(props, np.array(w_new), theta, chain=chain)
    np.testing.assert_allclose(c_gpu, exp_c, rtol=1e-12)
    assert np.array_equal(a_gpu, exp_a) and 0 < a_gpu.sum() < n
    st2 = h.get_state(chain)
    np.testing.assert_array_equal(st2["w"][:, 0], w[:, 0])
    g_g, c_g, _ = h.get_gbar(chain)
    np.testing.assert_allclose(g_g, g_s, rtol=1e-11, atol=1e-15)
    assert c_g == pytest.approx(c_s, rel=1e-12)
    bad = props[:1].copy()
    bad[0]["k"] = 1
    with pytest.raises(SacError):
        h.replay(bad, np.array(w_new), theta, chain=chain)
    h.close()
