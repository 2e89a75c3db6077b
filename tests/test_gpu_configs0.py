"""BASELINE.json configs[0] on the GPU: sac/free, fermionic (finite-T kernel) synthetic t.in with N_tau = 40, 1000 delta
functions, ONE complete Theta-annealing run -- FreeSAC.run as `julia sac_free.jl` runs it (free:1160-1261): in_free.in and t.in
in, the reference's output folder layout out, with the reference's stop rule deciding where the main anneal ends.  The "CPU
Julia" leg of configs[0] cannot run here (no Julia in the image); the move-for-move parity of this shape against the oracle is
tests/test_gpu_free.py::test_move_for_move_parity[configs0_ntau40_finiteT_n1000]."""
import os

import numpy as np
import pytest

from sac_amd.free import FreeSAC
from sac_amd.synth import synthetic_timedata
from sac_amd.tin import read_tin, write_tin

pytestmark = pytest.mark.gpu


def test_configs0_full_annealing_run(tmp_path):
    # a fermion problem as the reference runs it (free/out_files/fermionic_gaussians: beta = 8): tau up to beta / 2, so both signs of
    # omega are visible in G(tau) (K(omega < 0, tau) = exp(-|omega| (beta - tau)))
    td = synthetic_timedata(n_tau=40, beta=8.0, tau_max=4.0, rel_noise=1e-4, seed=40, n_boot=2000, kernel_type="finiteT")
    tin = str(tmp_path / "t.in")
    write_tin(tin, td.beta, td.n_boot, td.norm, td.tau, td.G_raw, td.sigma_bar, td.sigma, td.cov)
    back = read_tin(tin)
    assert back.n_tau == 40 and np.allclose(back.G, td.G, rtol=1e-12)
    out = str(tmp_path / "out" / "fermionic_synth")
    infile = str(tmp_path / "in_free.in")
    # in_free.in as the reference reads it (free:1163-1170): par / N_w w_0 w_m dw dw_h / theta_0 f_anneal f_final a1 a2 /
    # N_anneal anneal_steps sample_steps / G_file output_folder / symm kernel_type
    with open(infile, "w") as f:
        f.write("1\n1000 -8.0 8.0 0.00002 0.02\n20.0 1.25 1.05 0.5 0.5\n60 100 200\n" + f"{tin} {out}\n0 finiteT\n")
    s = FreeSAC.from_infile(infile, n_chains=4, seed=17)
    assert (s.par, s.N_omega, s.kernel_type, s.symm, s.N_anneal) == (1, 1000, "finiteT", 0, 60)
    theta_opt = s.run(infile)
    n_tau = 40
    # the reference's output folder (free:1182-1185, 931-955): copies of the inputs, the three CSVs, the log, the spectrum
    for name in ("in_free.in", "t.in", "anneal.csv", "accept_rate.csv", "sample.csv", "log.txt", "sw000.csv"):
        assert os.path.getsize(os.path.join(out, name)) > 0, name
    assert open(os.path.join(out, "anneal.csv")).readline().strip() == "i,theta,chi2_min,chi2_avg"
    assert open(os.path.join(out, "sample.csv")).readline().strip() == "i,a,theta,chi2_min,chi2_avg"
    assert open(os.path.join(out, "sw000.csv")).readline().strip() == "omega,S"
    log = open(os.path.join(out, "log.txt")).read()
    for msg in ("Beginning Initialization.", "Equillibration Finished.", "Main Anneal Finished.", "Final Anneal Finished."):
        assert msg in log
    a = np.loadtxt(os.path.join(out, "anneal.csv"), delimiter=",", skiprows=1)
    n = len(a)
    assert 5 < n <= 60 and np.array_equal(a[:, 0], np.arange(1, n + 1))
    np.testing.assert_allclose(a[:, 1], np.round(20.0 / 1.25 ** np.arange(n), 8), rtol=0, atol=1e-8)     # the Theta ladder
    # <chi2> falls along the anneal and stays above chi2_min; the stop rule (free:1087) is what ended a shortened anneal
    assert a[-1, 3] < a[0, 3] and (a[:, 3] >= a[:, 2] - 1e-4).all()
    if n < 60:
        assert (a[-1, 3] - a[-1, 2]) < s.tol + 1e-4 and ((a[:-1, 3] - a[:-1, 2]) >= s.tol - 1e-4).all()
    # Theta_opt is a ladder point: the one whose a-value is closest to a2 (free:1249-1252)
    ladder = 20.0 / 1.25 ** np.arange(60)
    assert np.isclose(ladder, theta_opt, rtol=1e-12).any()
    smp = np.loadtxt(os.path.join(out, "sample.csv"), delimiter=",", skiprows=1, ndmin=2)
    assert len(smp) == 1 and smp[0, 2] == pytest.approx(theta_opt) and smp[0, 4] >= smp[0, 3]     # a1 == a2: one final sample
    assert abs(smp[0, 1] - 0.5) < 0.6                                        # the a-criterion is met near a2 = 0.5
    # the spectrum: d_omega_h grid, normalised to norm * pi (finite-T: both signs of omega carry weight)
    sw = np.loadtxt(os.path.join(out, "sw000.csv"), delimiter=",", skiprows=1)
    assert np.allclose(np.diff(sw[:, 0]), 0.02, atol=1e-9) and (sw[:, 1] >= 0).all()
    # neighbouring coarse bins share one fine bin (the overlap quirk of free:980), so the sum may slightly exceed norm * pi; and
    # `sampled_spec[idx] .+= A` (free:918) adds only ONE of the deltas that share a fine-grid point, so a cooled chain with a sharp
    # peak loses weight there (the reference's behaviour, kept: sac_amd/csrc/hist_sort.h): the integral is bounded, not fixed
    # (0.21 on a 2e-4 grid, where the 700 deltas of the peak share a few dozen grid points; the reference's own bosonic_delta_peak
    # spectrum on its 1e-5 grid integrates to 0.988)
    integral = sw[:, 1].sum() * 0.02 / (td.norm * np.pi)
    print("integral of sw000.csv / (norm pi):", integral)
    assert 0.05 < integral < 1.06
    assert sw[0, 0] < 0 < sw[-1, 0]
    s.h.close()
