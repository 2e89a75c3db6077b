"""Host-side driver logic that needs no GPU: input parsing, Julia number formatting, the
write_spec post-processing (free:959-1019) and the Theta_opt rule (free:1249-1252)."""
import os

import numpy as np
import pytest

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


# ---- the shipped in-files of the three programs (parameter rows as in sac/*/in_*.in) -----------------

_TAIL = "\n\n######################################################################\n\n1: a comment block follows the rows and is ignored\n"


def test_peak_infile_rows_and_folder(tmp_path):
    from sac_amd.peak import PeakSAC
    f = tmp_path / "in_peak.in"
    f.write_text("2000 1\n0.70 15. 0.00001 0.005\n15 1.03 0.5\n250 5000 5000\n../in_files/peak/t_f2.in out_files/fermionic2_v2\n"
                 "0 0 finiteT\n" + _TAIL)
    s = PeakSAC.from_infile(str(f))
    assert (s.N_omega, s.N_p, s.A_0, s.omega_m, s.d_omega, s.d_omega_h) == (2000, 1, 0.7, 15.0, 1e-5, 0.005)
    assert (s.theta_0, s.f_anneal, s.a_criterion, s.N_anneal, s.anneal_steps, s.sample_steps) == (15.0, 1.03, 0.5, 250, 5000, 5000)
    assert (s.fix_edge, s.symm, s.kernel_type) == (0, 0, "finiteT")
    assert s.output_folder == "out_files/fermionic2_v2/Np_01/A0_0.700"                   # peak:1473
    s = PeakSAC.from_infile(str(f), A_0_in=0.65, N_p_in=2)                             # peak:1573-1575
    assert s.output_folder.endswith("/Np_02/A0_0.650") and (s.A_0, s.N_p) == (0.65, 2)
    f.write_text("2000 1\n0.70 15. 0.00001 0.005\n15 1.03 0.5\n250 5000 5000\nx.in out\n0.8 1 finiteT\n")
    s = PeakSAC.from_infile(str(f))
    assert (s.fix_edge, s.omega_0, s.symm) == (1, 0.8, 1) and s.output_folder == "out_symm/Np_01/A0_0.700"


def test_edge_infile_rows_and_folder(tmp_path):
    from sac_amd.edge import EdgeSAC
    f = tmp_path / "in_edge.in"
    rows = "100 0\n-20.0 20.0 0.005 0.00001\n0.5 0.0 0.5\n10 1.05 200 0.7\n600 10000 20\n../in_files/hchain_beta2048.in out_files/hchain_beta2048\n0 bosonic\nsingle_edge\n"
    f.write_text(rows + _TAIL)
    s = EdgeSAC.from_infile(str(f))
    assert (s.N_e, s.N_c, s.omega_0, s.omega_m, s.d_omega_h, s.d_omega) == (100, 0, 0.0, 20.0, 0.005, 1e-5)   # edge:1888, 1968
    assert (s.p, s.A_c, s.A_r, s.theta_0, s.f_anneal, s.N_anneal, s.a_criterion) == (0.5, 0.0, 1.0, 10.0, 1.05, 200, 0.7)
    assert (s.anneal_steps, s.sample_steps, s.bins, s.mode, s.kernel_type) == (600, 10000, 20, "single_edge", "bosonic")
    assert s.output_folder == "out_files/hchain_beta2048_single/Nw100/Ac_0.000/p_0.500"          # edge:1910-1957
    f.write_text(rows.replace("0 bosonic\nsingle_edge", "0 finiteT\ndouble_edge_in").replace("0.5 0.0 0.5", "0.4 0.2 0.7"))
    s = EdgeSAC.from_infile(str(f))
    assert s.N_c == 50 and s.omega_0 == -20.0                                            # edge:1931-1934: N_c = ceil(N_e / 2)
    assert s.output_folder == "out_files/hchain_beta2048_double_in/Nw100/Ac_0.200/p_0.400/Ar_0.700"
    s = EdgeSAC.from_infile(str(f), A_c_in=0.0, A_r_in=0.5, p_in=0.3, scan=True)        # edge:2069-2071
    assert s.N_c == 0 and s.output_folder == "out_files/hchain_beta2048_double_in/Nw100_scan/Ac_0.000/p_0.300/Ar_0.500"
    f.write_text(rows.replace("single_edge", "double_edge_symm"))
    with pytest.raises(ValueError):
        EdgeSAC.from_infile(str(f))                                                      # edge:1964-1967


def test_edge_amplitude_profile_and_initial_scan_product_side():
    """The product's own host initialisation (sac_amd/edge.py) against the reference's logged answer for
    hchain_beta2048_single on a 10x coarser grid (the scan does not depend on the grid at this accuracy)."""
    import os
    from sac_amd.edge import EdgeProblem, set_amplitudes
    from sac_amd.tin import read_tin
    A = set_amplitudes(80, 100, 1 - 2 * 0.3, 60.0, 0.5, 0.25, 0.7, 0)
    assert A[0] == 0 and A[1:81].sum() == pytest.approx(0.7 / 1.25) and A[81:].sum() == pytest.approx(0.25 / 1.25)
    assert (np.diff(A[1:81]) > 0).all()                       # c = 1 - 2p > 0: amplitudes grow with the delta index
    td = read_tin(os.path.join(os.path.dirname(__file__), "golden", "edge", "hchain_single", "t.in"))
    pr = EdgeProblem(td, "single_edge", "bosonic", 80, 0, 0.5, 0.0, 1.0, 0.0, 20.0, 1e-4)
    assert (pr.j_opt, pr.k_opt) == (37, 3)                    # log.txt: "Rightward Edge: [1.44, 2.1745...] (37, 3)"
    assert pr.w[1, 0] == pytest.approx(1.44, abs=1e-12) and pr.w[80, 0] == pytest.approx(2.174545454545454, abs=1e-12)


@pytest.mark.parametrize("rep", [1, 2])
def test_calc_cv_product_side_known_answer(rep):
    """sac_amd.cross_val.calc_cv against the reference's committed free_v.csv (same known answer that pins the oracle)."""
    from sac_amd.cross_val import calc_cv
    cv = os.path.join(os.path.dirname(__file__), "golden", "cv")
    exp = np.loadtxt(f"{cv}/free_v_expected.csv", delimiter=",", ndmin=2)
    th, v_ab = calc_cv(f"{cv}/t_{rep:03d}a.in", f"{cv}/GSAC_free_{rep:03d}b.csv")
    _, v_ba = calc_cv(f"{cv}/t_{rep:03d}b.in", f"{cv}/GSAC_free_{rep:03d}a.csv")
    v = 0.5 * (v_ab + v_ba)
    n = min(len(v), exp.shape[1])
    np.testing.assert_allclose(th[:n], exp[0][:n], rtol=1e-9)
    np.testing.assert_allclose(v[:n], exp[rep][:n], rtol=1e-9)
