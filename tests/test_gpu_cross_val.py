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
    s = free_cv(1, "a", in_file=str(in_file), n_chains=64, seed=77)
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
