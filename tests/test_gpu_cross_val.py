"""cross_val.jl's free_cv on the GPU against the GSAC.csv a real reference run wrote
(tests/golden/cv/GSAC_free_001a.csv = cross_validation/out_files/bosonic_delta_peak/001a/free/GSAC.csv)."""
import os
import shutil

import numpy as np
import pytest

from sac_amd.cross_val import free_cv
from sac_amd.tin import read_tin

pytestmark = pytest.mark.gpu
CV = os.path.join(os.path.dirname(__file__), "golden", "cv")


def test_free_cv_gsac_follows_reference(tmp_path):
    # lay the inputs out as cross_val.jl expects: in_free.in names a folder of t_%03i%s.in files
    rows = open(f"{CV}/run001a_free_in_free.in").read().split("\n")
    rows[3] = "60 100 200"
    rows[4] = f"{CV} {tmp_path}/out"
    in_file = tmp_path / "in_free.in"
    in_file.write_text("\n".join(rows))
    s = free_cv(1, "a", in_file=str(in_file), n_chains=64, seed=77, device_tables=True)
    assert s.output_folder.endswith("/001a/free") and s.tol == 1e-6
    got = np.loadtxt(s.output_folder + "/GSAC.csv", delimiter=",")
    ref = np.loadtxt(f"{CV}/GSAC_free_001a.csv", delimiter=",")
    n = len(got)
    assert n == 60 and got.shape[1] == ref.shape[1] == 1 + s.td.n_tau
    assert np.array_equal(got[:, 0], ref[:n, 0])                     # Theta column, same rounding
    td = read_tin(f"{CV}/t_001a.in")
    # deviation of our chain-mean Gbar from the reference's single-chain snapshot, in the eigenbasis and
    # in units of the data's errors; the snapshot itself sits at chi2(Theta) from the data
    z = ((got[:, 1:] - ref[:n, 1:]) @ td.cov) * td.sigma_inv
    d2 = (z ** 2).sum(axis=1) / td.n_tau
    anneal = np.loadtxt(s.anneal_file, delimiter=",", skiprows=1)
    print("dev^2/N_tau", d2[::6], "chi2avg/N_tau", anneal[::6, 3])
    assert (d2 < 0.25 * anneal[:, 3]).all()
    assert np.abs(got[-1, 1:] - ref[n - 1, 1:]).max() < 2e-4         # tau basis, late stage


def test_peak_cv_and_chain_repair(tmp_path):
    """peak_cv (cv:147-312) writes one GSAC row per Theta stage; a chain that failed the reference's own test for a
    bad start (chi2_min > 1000 N_tau, peak:1522) takes over a healthy chain's configuration and then continues on
    its own random stream (sac_copy_chain)."""
    from sac_amd.cross_val import peak_cv
    from sac_amd.peak import PeakSAC
    rows = ["60 2", "0.6 6.0 0.002 0.05", "5 1.3 0.5", "6 40 40", f"{CV} {tmp_path}/out", "0 0 bosonic"]
    in_file = tmp_path / "in_peak.in"
    in_file.write_text("\n".join(rows) + "\n")
    s = peak_cv(1, "a", in_file=str(in_file), n_chains=8, seed=5)
    assert s.output_folder == f"{tmp_path}/out/001a/peak/Np_02/A0_0.600" and s.tol == 1e-6
    got = np.loadtxt(s.output_folder + "/GSAC.csv", delimiter=",")
    anneal = np.loadtxt(s.anneal_file, delimiter=",", skiprows=1)
    assert got.shape == (len(anneal), 1 + s.td.n_tau) and 1 <= len(anneal) <= 6
    np.testing.assert_allclose(got[:, 0], anneal[:, 1], rtol=1e-7)
    td = read_tin(f"{CV}/t_001a.in")
    assert np.abs(got[-1, 1:] - td.G_raw).max() < 0.05          # Gbar (tau basis) near the data after the short anneal

    p = PeakSAC.from_infile(str(in_file), n_chains=4, seed=9)
    p.G_file, p.output_folder = f"{CV}/t_001a.in", str(tmp_path / "repair")
    p.init_outfiles()
    p.initialize()
    p.intial_sampling(p.theta_0)
    wi, A, win = p.h.get_state(2)
    bad = wi.copy()
    bad[:62] = p.prob.wi_m - np.arange(62)                      # everything piled up at the far end of the grid
    p.h.set_state(bad, A, win, chain=2)
    p.h.init_gbar()
    _, cmin = p.h.chain_chi2()
    assert cmin[2] > 1000 * p.td.n_tau and (np.delete(cmin, 2) < 1000 * p.td.n_tau).all()
    assert p.repair_chains() == 1
    best = int(np.argmin(np.where(np.arange(4) == 2, np.inf, cmin)))
    for a, b in zip(p.h.get_state(2), p.h.get_state(best)):
        assert np.array_equal(a, b)
    assert p.h.get_gbar(2)[1] == p.h.get_gbar(best)[1]
    p.sample(20, 5.0)
    assert not np.array_equal(p.h.get_state(2)[0], p.h.get_state(best)[0])     # own Philox stream afterwards
    assert p.repair_chains() == 0


def test_edge_cv_gsac(tmp_path):
    """edge_cv (cv:315-509): equilibration + the whole Theta ladder, one GSAC row per stage."""
    from sac_amd.cross_val import edge_cv
    rows = ["24 0", "0.0 6.0 0.05 0.002", "0.5 0.0 1.0", "5 1.5 5 0.5", "60 100 4", f"{CV} {tmp_path}/out", "0 bosonic",
            "single_edge"]
    in_file = tmp_path / "in_edge.in"
    in_file.write_text("\n".join(rows) + "\n")
    s = edge_cv(1, "a", in_file=str(in_file), n_chains=8, seed=5)
    assert s.output_folder == f"{tmp_path}/out/001a/edge/p_0.500"
    got = np.loadtxt(s.output_folder + "/GSAC.csv", delimiter=",")
    anneal = np.genfromtxt(s.output_folder + "/anneal.csv", delimiter=",", skip_header=1)
    assert got.shape == (5, 1 + s.td.n_tau) and len(anneal) == 6          # row 0 = equilibration
    np.testing.assert_allclose(got[:, 0], anneal[1:, 1], rtol=1e-7)
    td = read_tin(f"{CV}/t_001a.in")
    assert np.abs(got[-1, 1:] - td.G_raw).max() < 0.05
    assert s.res["violations"] == 0
