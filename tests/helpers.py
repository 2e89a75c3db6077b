"""Shared builders for the parity tests: the same problem on the HIP path and on the oracle."""
import os

import numpy as np

from oracle import oracle as ora
from sac_amd.free import FreeProblem
from sac_amd.synth import synthetic_timedata
from sac_amd.tin import read_tin

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def problem_golden_bosonic(n_omega=60, omega_m=4.0, d_omega=2e-3, par=1):
    td = read_tin(os.path.join(GOLDEN, "cv", "t_001a.in"))          # N_tau = 22
    return FreeProblem(td, n_omega, 0.0, omega_m, d_omega, par=par, kernel_type="bosonic")


def problem_synth(n_tau=64, n_omega=2000, omega_m=10.0, d_omega=1e-4, par=1, kernel_type="bosonic",
                  omega_0=0.0, symm=0, seed=12345, n_boot=2000):
    td = synthetic_timedata(n_tau=n_tau, seed=seed, n_boot=n_boot, kernel_type=kernel_type)
    return FreeProblem(td, n_omega, omega_0, omega_m, d_omega, par=par, symm=symm, kernel_type=kernel_type)


def oracle_chain(prob, seed, chain, hist_mode=0):
    o = ora.OracleFree(prob.K, prob.td.G, prob.td.sigma_inv, prob.n_omega, prob.wi_0, prob.wi_m,
                       par=prob.par, symm=prob.symm)
    o.set_state(prob.wi, prob.A, np.full(4, prob.window))
    o.init_gbar()
    o.use_philox(seed, chain)
    o.set_hist_mode(hist_mode)
    return o


def problem_peak(kind="golden", n_omega=60, n_p=2, A_0=0.6, omega_m=6.0, d_omega=2e-3, kernel_type="bosonic",
                 symm=0, approx_ratio=None, n_tau=64, fix_edge=0, omega_0=0.0):
    from sac_amd.peak import PeakProblem
    if kind == "golden":
        td = read_tin(os.path.join(GOLDEN, "cv", "t_001a.in"))
    else:
        td = synthetic_timedata(n_tau=n_tau, seed=4321, n_boot=2000,
                                kernel_type=kernel_type if kernel_type != "finiteT" else "finiteT")
    return PeakProblem(td, n_omega, n_p, A_0, omega_m, d_omega, omega_0=omega_0, fix_edge=fix_edge, symm=symm,
                       kernel_type=kernel_type, approx_ratio=approx_ratio)


def oracle_peak_chain(prob, seed, chain, indiv=False):
    Kn = prob.Kn if prob.Kn is not None else np.zeros_like(prob.Kp)
    o = ora.OraclePeak(prob.Kp, Kn, prob.td.G, prob.td.sigma_inv, prob.n_omega, prob.n_p, prob.wi_0,
                       prob.wi_m, symm=prob.symm, bosonic=int(prob.kernel_type == "bosonic"),
                       fix_edge=prob.fix_edge)
    o.set_state(prob.wi, prob.A, np.full(11, prob.window / 10))
    o.init()
    o.use_philox(seed, chain)
    o.set_indiv_update(indiv)
    return o
