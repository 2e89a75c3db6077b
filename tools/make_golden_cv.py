#!/usr/bin/env python3
"""Extract the cross-validation known-answer fixture from the reference's own
artifacts into tests/golden/cv/ (data only: inputs and expected outputs).

Source (read-only, /root/reference/sac/cross_validation):
  in_files/bosonic_delta_peak/t_{rep}{a,b}.in            -- t.in inputs
  out_files/bosonic_delta_peak/{rep}{a,b}/free/GSAC.csv  -- G_SAC(tau) per Theta (cv:105-108)
  out_files/bosonic_delta_peak/chi2/free_v.csv           -- expected output of calc_cv.jl:82-107
                                                            (row 1 = Theta, row 1+rep = chi2_v/N_tau)
Run once in the build container; the outputs are committed.
"""
import os, shutil, sys

REF = "/root/reference/sac/cross_validation"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cv")
REPS = [1, 2]

def main():
    os.makedirs(OUT, exist_ok=True)
    for rep in REPS:
        for half in "ab":
            shutil.copyfile(f"{REF}/in_files/bosonic_delta_peak/t_{rep:03d}{half}.in",
                            f"{OUT}/t_{rep:03d}{half}.in")
            shutil.copyfile(f"{REF}/out_files/bosonic_delta_peak/{rep:03d}{half}/free/GSAC.csv",
                            f"{OUT}/GSAC_free_{rep:03d}{half}.csv")
    with open(f"{REF}/out_files/bosonic_delta_peak/chi2/free_v.csv") as f:
        rows = f.read().splitlines()
    with open(f"{OUT}/free_v_expected.csv", "w") as f:
        f.write(rows[0] + "\n")
        for rep in REPS:
            f.write(rows[rep] + "\n")
    # one committed reference run (statistical oracle): anneal curve + its parameters
    run = f"{REF}/out_files/bosonic_delta_peak/001a/free"
    for name in ("anneal.csv", "accept_rate.csv", "in_free.in"):
        shutil.copyfile(f"{run}/{name}", f"{OUT}/run001a_free_{name}")
    os.system(f"chmod -R u+w {OUT}")



def peak_fixture():
    """Committed reference run of sac_peak.jl (statistical oracle for the peak path):
    sac/peak/out_files/fermionic2/Np_01/A0_0.700/{t.in,in_peak.in,anneal.csv,sample.csv,accept_rate.csv,sw000.csv}
    -> tests/golden/peak_f2/ (data only; the other committed peak runs were written by older versions of
    sac_peak.jl whose accept_rate.csv header and fixed +/- weight split differ from the current code)."""
    src = "/root/reference/sac/peak/out_files/fermionic2/Np_01/A0_0.700"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "peak_f2")
    os.makedirs(out, exist_ok=True)
    for name in ("t.in", "in_peak.in", "anneal.csv", "sample.csv", "accept_rate.csv", "sw000.csv"):
        shutil.copyfile(f"{src}/{name}", f"{out}/{name}")
    os.system(f"chmod -R u+w {out}")
    with open(f"{out}/in_peak.in") as f:          # keep the six parameter rows, drop the commentary
        rows = f.readlines()[:6]
    with open(f"{out}/in_peak.in", "w") as f:
        f.writelines(rows)


def free_spectrum_fixture():
    """Committed complete reference run of sac_free.jl with its final spectrum (statistical oracle for S(omega)):
    sac/free/out_files/bosonic_delta_peak/{t.in,in_free.in,anneal.csv,sample.csv,sw000.csv} -> tests/golden/free_bdp/
    (data only; in_free.in cut to its six parameter rows)."""
    src = "/root/reference/sac/free/out_files/bosonic_delta_peak"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "free_bdp")
    os.makedirs(out, exist_ok=True)
    for name in ("t.in", "in_free.in", "anneal.csv", "sample.csv", "sw000.csv"):
        shutil.copyfile(f"{src}/{name}", f"{out}/{name}")
    os.system(f"chmod -R u+w {out}")
    with open(f"{out}/in_free.in") as f:
        rows = f.readlines()[:6]
    with open(f"{out}/in_free.in", "w") as f:
        f.writelines(rows)


def peak_cv_fixture():
    """A committed reference run of sac_peak.jl whose schedule is short enough to be repeated in full by a test (statistical
    oracle with the reference's OWN sweeps per Theta): cross_validation/out_files/bosonic_delta_peak/001a/peak/Np_01/A0_0.700/
    {in_peak.in, anneal.csv, accept_rate.csv} -> tests/golden/peak_cv001a/ (its t.in is tests/golden/cv/t_001a.in)."""
    src = f"{REF}/out_files/bosonic_delta_peak/001a/peak/Np_01/A0_0.700"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "peak_cv001a")
    os.makedirs(out, exist_ok=True)
    for name in ("in_peak.in", "anneal.csv", "accept_rate.csv"):
        shutil.copyfile(f"{src}/{name}", f"{out}/{name}")
    os.system(f"chmod -R u+w {out}")
    with open(f"{out}/in_peak.in") as f:
        rows = f.readlines()[:6]
    with open(f"{out}/in_peak.in", "w") as f:
        f.writelines(rows)


if __name__ == "__main__":
    main()
    peak_fixture()
    free_spectrum_fixture()
    peak_cv_fixture()
