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
    s = PeakSAC.from_infile(f"{REF}/in_peak.in", n_chains=32, seed=11)
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
