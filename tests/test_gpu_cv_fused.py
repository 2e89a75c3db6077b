"""Cross-validation scoring on the device: sac_validation_chi2 computes what calc_cv.jl:29-58
computes from a GSAC.csv row, directly from the Gbar the chains hold."""
import os

import numpy as np
import pytest

from sac_amd import cross_val
from sac_amd.free import FreeProblem
from sac_amd.tin import read_tin

pytestmark = pytest.mark.gpu
CV = os.path.join(os.path.dirname(__file__), "golden", "cv")


def test_device_scoring_reproduces_the_references_known_answers():
    """The reference's committed GSAC.csv rows (real Julia runs) scored on the device against the other half's t.in reproduce
    the committed output of calc_cv.jl (chi2/free_v.csv) to 1e-9 -- the same known answer that pins the oracle."""
    exp = np.loadtxt(f"{CV}/free_v_expected.csv", delimiter=",", ndmin=2)
    thetas = exp[0]
    for r, rep in enumerate((1, 2)):
        td = {ab: read_tin(f"{CV}/t_{rep:03d}{ab}.in") for ab in "ab"}
        curves = []
        for fit, other in (("b", "a"), ("a", "b")):                   # GSAC of one half against the other half's data
            rows = np.loadtxt(f"{CV}/GSAC_free_{rep:03d}{fit}.csv", delimiter=",", ndmin=2)
            assert np.allclose(rows[:, 0], thetas[: len(rows)], rtol=1e-7)
            prob = FreeProblem(td[fit], 20, 0.0, 2.0, 1e-2, kernel_type="bosonic")
            h = prob.make_handle(n_chains=2, seed=1)
            v = np.empty(len(rows))
            for i, row in enumerate(rows):
                gbar = td[fit].cov.T @ row[1:]                         # the Gbar whose cov * Gbar the row holds (cv:106)
                h.set_gbar(gbar, 0.0, 0.0)
                v[i] = h.validation_chi2(td[fit].cov, td[other]) / td[other].n_tau
            h.close()
            curves.append(v)
        got = 0.5 * (curves[0] + curves[1])                           # calc_cv.jl:87-99
        np.testing.assert_allclose(got, exp[1 + r][: len(got)], rtol=1e-9)


def test_fused_scores_equal_the_file_route(tmp_path):
    """A cross-validation run with validate_tin: the per-stage validation chi^2 taken on the device equals calc_cv applied
    to the GSAC.csv the same run wrote (the chain-mean Gbar is scored in both), and per-chain scores are available."""
    in_file = tmp_path / "in_free.in"
    rows = open(f"{CV}/run001a_free_in_free.in").read().split("\n")
    rows[3] = "12 40 100"                                             # N_anneal, anneal_steps, sample_steps
    rows[4] = f"{CV} {tmp_path}/out"
    in_file.write_text("\n".join(rows))
    runs = {}
    for ab, other in (("a", "b"), ("b", "a")):
        runs[ab] = cross_val.free_cv(1, ab, in_file=str(in_file), n_chains=8, seed=3, device_tables=True,
                                     validate_tin=f"{CV}/t_001{other}.in")
    for ab, other in (("a", "b"), ("b", "a")):
        s = runs[ab]
        th, v_file = cross_val.calc_cv(f"{CV}/t_001{other}.in", s.output_folder + "/GSAC.csv")
        fused = np.array(s.validation)
        assert len(fused) == len(v_file) == 12
        np.testing.assert_allclose(fused[:, 0], th, rtol=1e-12)
        np.testing.assert_allclose(fused[:, 1], v_file, rtol=1e-8)
        on_disk = np.loadtxt(s.output_folder + "/validation.csv", delimiter=",", skiprows=1)
        np.testing.assert_allclose(on_disk, fused, rtol=1e-14)
        mean, chains = s.h.validation_chi2(s.td.cov, read_tin(f"{CV}/t_001{other}.in"), per_chain=True)
        assert chains.shape == (8,) and (chains > 0).all() and mean / s.td.n_tau == pytest.approx(fused[-1, 1], rel=1e-12)
        assert mean <= chains.mean() * (1 + 1e-12)                    # chi2 is convex: the mean Gbar scores below the mean score
    th, both = cross_val.cv_score_fused(runs["a"], runs["b"])
    _, ref = cross_val.cv_score(runs["a"].output_folder, runs["b"].output_folder, f"{CV}/t_001a.in", f"{CV}/t_001b.in")
    np.testing.assert_allclose(both, ref, rtol=1e-8)
    for s in runs.values():
        s.h.close()
