"""Host-side driver logic that needs no GPU: input parsing, Julia number formatting, the
write_spec post-processing (free:959-1019) and the Theta_opt rule (free:1249-1252)."""
import os

import numpy as np

from sac_amd.free import FreeSAC, jl_float, _jl_round
from sac_amd.tin import read_tin

CV = os.path.join(os.path.dirname(__file__), "golden", "cv")


def test_infile_parsing_matches_reference_layout():
    s = FreeSAC.from_infile(f"{CV}/run001a_free_in_free.in")
    assert (s.par, s.N_omega, s.omega_m, s.d_omega, s.d_omega_h) == (1, 1000, 15.0, 1e-5, 0.005)
    assert (s.theta_0, s.f_anneal, s.f_final, s.a1, s.a2) == (10.0, 1.1, 1.05, 0.5, 0.5)
    assert (s.N_anneal, s.anneal_steps, s.sample_steps) == (150, 500, 5000)
    assert s.kernel_type == "bosonic" and s.omega_0 == 0.0 and s.symm == 0      # free:1172-1174


def test_number_formatting_reproduces_reference_csv():
    # every number in the committed reference anneal.csv must round-trip through jl_float
    rows = open(f"{CV}/run001a_free_anneal.csv").read().split()[1:]
    for row in rows:
        for tok in row.split(","):
            assert jl_float(float(tok)) == tok, tok
    assert jl_float(_jl_round(10.0 / 1.1 ** 149, 8)) == rows[-1].split(",")[1]   # theta column


def test_theta_opt_rule_includes_unused_tail():
    s = FreeSAC(1, 10, 0, 1, 0.1, 0.2, 10.0, 2.0, 1.05, 0.5, 0.5, 5, 10, 10, "x", "y", 0, "bosonic")
    s.chi2_min = 8.0
    s.chi2_anneal = np.array([20.0, 12.0, 10.1, 0.0, 0.0])      # anneal stopped after 3 steps
    # a = (chi2 - 8)/4 = [3, 1, .525, -2, -2]; closest to 0.5 is index 2 -> theta_0 / 2^2
    assert s.optimal_theta() == 2.5


class _FakeHandle:
    def __init__(self, hist):
        self._h = hist

    def histogram(self):
        return self._h.copy()


def test_write_spec_rebin_overlap_trim_and_mirror(tmp_path):
    td = read_tin(f"{CV}/t_001a.in")
    s = FreeSAC(1, 10, 0.0, 1.0, 0.01, 0.05, 10.0, 1.1, 1.05, 0.5, 0.5, 5, 10, 10, "x", str(tmp_path), 0,
                "bosonic", timedata=td)
    s.td = td
    from sac_amd.free import FreeProblem
    s.prob = FreeProblem(td, 10, 0.0, 1.0, 0.01, kernel_type="bosonic")
    fine = np.zeros(s.prob.n_grid)
    fine[[3, 5, 10, 22]] = [0.1, 0.2, 0.3, 0.4]                 # fine bins (0-based)
    s.h = _FakeHandle(fine)
    coarse = s.write_spec(0)
    conv, N_h = 5, 20
    assert len(coarse) == N_h
    w = 0.01 * np.arange(101)
    scaled = fine * (td.norm * np.pi) / (1 + np.exp(-td.beta * w))
    expect = np.array([scaled[k * conv:(k + 1) * conv + 1].sum() / 0.05 for k in range(N_h)])
    # bins 0 and 1 both contain fine bin 5 and bins 1 and 2 both contain fine bin 10 (free:980)
    np.testing.assert_allclose(coarse, expect, rtol=1e-15)
    assert coarse[0] > 0 and coarse[1] > 0 and coarse[2] > 0 and coarse[4] > 0 and coarse[5] == 0
    lines = open(tmp_path / "sw000.csv").read().split("\n")
    assert lines[0] == "omega,S"
    # trimmed to the last bin above 1e-10 (free:987-992: the scan starts at N_h-1), mirrored for bosonic
    body = [ln.split(",") for ln in lines[1:] if ln]
    assert len(body) == 2 * 5
    assert body[0][0] == "-0.2" and body[4][0] == "-0.0" and body[5][0] == "0.0" and body[-1][0] == "0.2"
    assert float(body[5][1]) == coarse[0]
