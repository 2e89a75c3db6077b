"""End-to-end peak driver on the GPU against a committed run of the reference (statistical oracle):
tests/golden/peak_f2 = sac/peak/out_files/fermionic2/Np_01/A0_0.700 (its t.in, in_peak.in and the
anneal.csv / sample.csv that real Julia run produced with 2 x 5000 sweeps per Theta).  The test walks
every 12th Theta of the reference's ladder with fewer sweeps but 32 chains."""
import os

import numpy as np
import pytest

from sac_amd.peak import PeakSAC

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(__file__), "golden", "peak_f2")


def test_peak_anneal_follows_reference_run(tmp_path):
    s = PeakSAC.from_infile(f"{REF}/in_peak.in", n_chains=32, seed=11, device_tables=True)     # two 1.5-million-column tables
    assert (s.N_omega, s.N_p, s.A_0, s.d_omega, s.kernel_type, s.theta_0) == (2000, 1, 0.7, 1e-5, "finiteT", 15.0)
    s.G_file = f"{REF}/t.in"
    s.output_folder = str(tmp_path / "out")
    stride = 18
    s.f_anneal = 1.03 ** stride
    s.N_anneal, s.anneal_steps, s.sample_steps = 12, 400, 100
    s.chi2_anneal = np.zeros(s.N_anneal)
    # run() of peak:1439-1569, with the final stage shortened (its ramp costs 110 x anneal_steps sweeps)
    s.init_outfiles()
    s.initialize()
    s.intial_sampling(s.theta_0)
    n_fixed = s.repair_chains()
    print("chains restarted after the initial sampling:", n_fixed)
    assert n_fixed < 32                      # at least one chain found the basin (reference: chi2_min < 1000 N_tau)
    s.run_anneal(s.theta_0)
    _, _, theta_opt = s.optimal_theta()
    s.anneal_steps = 40
    s.final_anneal(theta_opt)

    ref = np.loadtxt(f"{REF}/anneal.csv", delimiter=",", skiprows=1)[::stride]
    got = np.loadtxt(s.anneal_file, delimiter=",", skiprows=1)
    n = len(got)
    assert n == 12
    np.testing.assert_allclose(got[:, 1], ref[:n, 1], rtol=1e-6)              # same Theta values
    ratio = got[:, 3] / ref[:n, 3]                                           # <chi2>/N_tau
    print("chi2 ratio", ratio)
    print("chi2_min ratio", got[:, 2] / ref[:n, 2])
    # the first stages still carry the (much shorter than the reference's) initial sampling
    assert np.abs(np.log(ratio[2:])).max() < 0.25, ratio
    assert abs(np.log(ratio[2:]).mean()) < 0.15, np.log(ratio).mean()
    # edges and weights are fixed by the data: reference 1.00107 / -0.49961, 0.2144 / 0.4856 / 0.0959 / 0.2041
    last = got[-1]
    print("last row", last, "ref", ref[n - 1])
    assert abs(last[4] - 1.001) < 0.01 and abs(last[5] + 0.4996) < 0.01
    np.testing.assert_allclose(last[6:10], [0.2144, 0.4856, 0.0959, 0.2041], atol=0.02)
    assert np.allclose(got[:, 6:10].sum(axis=1), 1.0, atol=1e-6)
    smp = open(s.sample_file).read().split("\n")
    assert smp[0].startswith("i,a,theta,chi2_min,chi2_avg,edge_p") and len(smp) >= 2
    vals = [float(x) for x in smp[1].split(",")]
    print("sample.csv", vals)
    assert abs(vals[5] - 1.001) < 0.01 and abs(vals[6] + 0.4996) < 0.01 and vals[7] < 0.01
    sw = open(os.path.join(s.output_folder, "sw000.csv")).read().split("\n")
    assert sw[0] == "omega,S,S_p,S_c" and len(sw) > 50
    assert 0 < theta_opt <= 15.0
    # the final spectrum against the one the reference run wrote (peak:1170-1265: omega, S, S_p, S_c on the d_omega_h grid,
    # negative axis first): same layout, delta peaks in the same output bins, same weights per axis and part, and the
    # same continuum distribution (this run cools 18 x faster than the reference: the continuum is compared loosely)
    a = np.loadtxt(os.path.join(s.output_folder, "sw000.csv"), delimiter=",", skiprows=1)
    b = np.loadtxt(f"{REF}/sw000.csv", delimiter=",", skiprows=1)
    for t in (a, b):                                  # d_omega_h grid, the two axes meeting in the rows -0.0 and 0.0
        d = np.diff(t[:, 0])
        assert np.allclose(d[d > 1e-9], 0.005, atol=1e-9) and (d > 1e-9).sum() == len(d) - 1
    np.testing.assert_allclose(a[:, 1], a[:, 2] + a[:, 3], rtol=1e-12, atol=1e-300)
    for sign in (1, -1):
        ma, mb = a[:, 0] * sign > 0, b[:, 0] * sign > 0
        wa, wb = a[ma][np.argmax(a[ma, 2]), 0], b[mb][np.argmax(b[mb, 2]), 0]
        print("peak bin", wa, wb, "weights", a[ma, 1:].sum(axis=0) * 0.005, b[mb, 1:].sum(axis=0) * 0.005)
        assert abs(wa - wb) <= 0.0051                                       # the delta peak: same output bin or its neighbour
        ia, ib = a[ma, 1:].sum(axis=0) * 0.005, b[mb, 1:].sum(axis=0) * 0.005       # integrals of S, S_p, S_c on this axis
        assert ia[0] == pytest.approx(ib[0], rel=1e-4)                             # the axis' total weight is fixed by the data
        np.testing.assert_allclose(ia[1:] / ia[0], ib[1:] / ib[0], atol=0.03)      # its split into peak and continuum
        grid = np.linspace(0, 8, 401)                                               # |omega|

        def cdf(t, m):
            x, y = t[m, 0] * sign, t[m, 3]
            o = np.argsort(x)
            return np.interp(grid, x[o], np.cumsum(y[o]) / y.sum())
        ca, cb = cdf(a, ma), cdf(b, mb)
        print("continuum CDF difference", np.abs(ca - cb).max())
        assert np.abs(ca - cb).max() < 0.08


CVP = os.path.join(os.path.dirname(__file__), "golden", "peak_cv001a")
CVT = os.path.join(os.path.dirname(__file__), "golden", "cv", "t_001a.in")


def test_peak_anneal_reference_schedule_z_test(tmp_path):
    """tests/golden/peak_cv001a = cross_validation/out_files/bosonic_delta_peak/001a/peak/Np_01/A0_0.700 of the reference: a
    sac_peak.jl run (bosonic, fixed edge, 1000 + 1 deltas) short enough to repeat with ITS OWN schedule -- 500 + 500 sweeps
    per Theta, ladder 15 / 1.1^i -- on 64 chains, first 50 points.  The reference's anneal.csv holds one chain's 500-sweep
    average per Theta; with m, s the mean and scatter of the 64 GPU chains' 500-sweep averages,
        z = (m - ref) / (s * sqrt(1 + 1/64))
    is standard normal if both sample the same distribution.  Required: |z| < 4 everywhere, <z^2> < 2, |<z>| < 0.6
    (measured: 2.2, 1.15, 0.25)."""
    n_chains, n_pts = 64, 50
    s = PeakSAC.from_infile(f"{CVP}/in_peak.in", n_chains=n_chains, seed=11, device_tables=True)
    assert (s.N_omega, s.N_p, s.kernel_type, s.fix_edge, s.anneal_steps, s.f_anneal) == (1000, 1, "bosonic", 1, 500, 1.1)
    s.G_file = CVT
    s.output_folder = str(tmp_path / "out")
    s.N_anneal = n_pts
    s.chi2_anneal = np.zeros(n_pts)
    s.tol = 1e-6                                                    # cross_val.jl:200
    s.init_outfiles()
    s.initialize()
    s.intial_sampling(s.theta_0)
    assert s.repair_chains() == 0
    per_chain = []
    s.run_anneal(s.theta_0, on_stage=lambda i, th: per_chain.append(
        np.array([s.h.chain_stats(c)["sampled_chi2"] for c in range(n_chains)]) / s.td.n_tau))
    ref = np.loadtxt(f"{CVP}/anneal.csv", delimiter=",", skiprows=1)
    got = np.loadtxt(s.anneal_file, delimiter=",", skiprows=1)
    assert len(got) == n_pts == len(per_chain)
    np.testing.assert_allclose(got[:, 1], ref[:n_pts, 1], atol=6e-5)          # the reference rounds Theta to 4 digits
    v = np.array(per_chain)
    m, sd = v.mean(axis=1), v.std(axis=1, ddof=1)
    z = (m - ref[:n_pts, 3]) / (sd * np.sqrt(1 + 1 / n_chains))
    print("z", np.round(z, 2))
    assert np.abs(z).max() < 4.0, z
    assert (z ** 2).mean() < 2.0 and abs(z.mean()) < 0.6, ((z ** 2).mean(), z.mean())
    # fixed edge: the peak stays where initialize! put it, the weights are the input's (reference columns 5-10)
    np.testing.assert_allclose(got[:, 4:10], ref[:n_pts, 4:10], atol=1e-4)
    # the chi2_min column is the single-chain equivalent (mean over the chains of each chain's own minimum): it tracks the
    # reference's ONE chain; the minimum over all 64 chains lies below it
    print("chi2_min ratio", np.round(got[15:, 2] / ref[15:n_pts, 2], 3))
    assert np.abs(np.log(got[15:, 2] / ref[15:n_pts, 2])).max() < 0.12
    assert s.chi2_min_all <= s.chi2_min
