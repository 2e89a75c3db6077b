#!/usr/bin/env python3
"""Data-prep fixture: the reference's committed QMC-style bins and the t.in its make_tin_zeroT.jl
wrote from them.
  sac/process_G/synthetic/bosonic_delta_peak/cor.dat   (8000 bins x (G(0), 22 tau points), text, "1" separators)
      -> tests/golden/tin_bdp/bins.npz (the same numbers as integers q = x * 1e8, compressed)
  .../tgrid.dat -> tests/golden/tin_bdp/tgrid.dat,  .../t.in -> tests/golden/tin_bdp/t_ref.in
Data only; run once in the build container."""
import os
import shutil

import numpy as np

SRC = "/root/reference/sac/process_G/synthetic/bosonic_delta_peak"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "tin_bdp")


def main():
    os.makedirs(OUT, exist_ok=True)
    tau = np.loadtxt(f"{SRC}/tgrid.dat")
    flat = np.loadtxt(f"{SRC}/cor.dat")
    n = len(tau)                                   # G(0) + N_tau points
    n_b = len(flat) // (n + 1)
    rows = flat.reshape(n_b, n + 1)
    assert (rows[:, 0] == 1).all()                 # the separators
    # every number in cor.dat is a multiple of 1e-8: kept as integers (x = q / 1e8 reproduces the parsed double exactly)
    q = np.round(rows[:, 1:] * 1e8).astype(np.int32)
    assert np.array_equal(q / 1e8, rows[:, 1:])
    np.savez_compressed(f"{OUT}/bins.npz", q=q)
    shutil.copyfile(f"{SRC}/tgrid.dat", f"{OUT}/tgrid.dat")
    shutil.copyfile(f"{SRC}/t.in", f"{OUT}/t_ref.in")
    # the true spectrum the reference's generator wrote next to its bins (omega, S; rows 1-2 mark the delta peak as a spike of
    # height A0 / spacing): data for the restated make_G_tau / add_noise (sac_amd/synth.py, tests/test_synth_recipe.py)
    aw = np.loadtxt(f"{SRC}/aw.dat", delimiter=",", skiprows=1)
    np.savez_compressed(f"{OUT}/aw.npz", omega=aw[:, 0], S=aw[:, 1])
    os.system(f"chmod -R u+w {OUT}")


if __name__ == "__main__":
    main()
