"""Behavioural tests of the CPU oracle's sampler (structure, invariants, both draw sources)."""
import os

import numpy as np
import pytest

from oracle import oracle as ora

CV = os.path.join(os.path.dirname(__file__), "golden", "cv")


@pytest.fixture(scope="module")
def problem():
    t = ora.read_tin(f"{CV}/t_001a.in")
    init = ora.initialize(t, n_omega=60, omega_0=0.0, omega_m=4.0, d_omega=2e-3,
                          par=1, symm=0, kernel_type="bosonic")
    return t, init


def make(problem, par=1, n_omega=60):
    t, init = problem
    o = ora.OracleFree(init["K"], t["G_D"], t["sigma_inv"], n_omega, init["wi0"], init["wim"], par=par)
    o.set_state(init["wi"], init["A"], np.full(4, init["window"]))
    o.init_gbar()
    return o


def test_incremental_gbar_tracks_from_scratch(problem):
    t, init = problem
    o = make(problem)
    o.use_philox(11, 0)
    o.adjust_windows(50, 5.0)
    o.sample(40, 5.0)
    wi, A, win = o.get_state()
    g, c, cmin = o.get_gbar()
    ref = init["K"][wi - init["wi0"]].T @ A
    assert np.abs(g - ref).max() <= 1e-12 * np.abs(ref).max()
    assert c == ora.chi2(g, t["G_D"], t["sigma_inv"])     # chi2 is always the tree sum of the kept Gbar
    assert cmin <= c
    assert (wi >= init["wi0"]).all() and (wi <= init["wim"]).all()
    assert abs(A.sum() - 1.0) < 1e-13
    ar = o.accept_rates()
    assert 0 < ar[0] <= 1 and 0 < ar[1] <= 2 and 0 <= ar[2] <= 1 and ar[3] == ar[4] == 0


def test_reference_histogram_collapses_duplicates(problem):
    t, init = problem
    o = make(problem)
    o.use_philox(5, 1)
    o.sample(30, 2.0)
    s_ref = o.spec()
    o2 = make(problem)
    o2.use_philox(5, 1)
    o2.set_hist_mode(1)
    o2.sample(30, 2.0)
    s_acc = o2.spec()
    # same chain, same draws; accumulate mode conserves sum(A) per sweep, reference mode loses duplicates
    assert abs(s_acc.sum() - 1.0) < 1e-12
    assert s_ref.sum() <= s_acc.sum() + 1e-15
    assert (s_ref <= s_acc + 1e-15).all()


def test_first_moment_conserved_by_pair_and_triple_moves(problem):
    # par=1: equal amplitudes, so double/triple moves conserve sum(omega_i) up to the
    # truncation of the triple move's integer division (free:424); single moves do not.
    o = make(problem)
    o.use_philox(1, 0)
    o.enable_log(10000)
    o.run_updates(3.0)
    log = o.log()
    n = 60
    assert len(log) == n + n // 2 + n // 3
    assert (log["move"][:n] == 1).all() and (log["move"][n:n + n // 2] == 2).all()
    assert ((log["flags"] & 2) <= 2 * (log["flags"] & 1)).all()   # accepted implies evaluated


def test_trace_replay_reproduces_philox_chain(problem):
    """Record the Philox-driven chain's draws in the reference's rand() call order
    (SURVEY Appendix A), replay them through the explicit-trace source: identical chain."""
    t, init = problem
    n = 60
    o = make(problem)
    o.use_philox(42, 3)
    o.enable_log(1000)
    wi0s, _, win = o.get_state()
    o.run_updates(1.5)
    log = o.log()

    # rebuild the draw sequence by hand from the Philox spec
    def words(move, it, b, sweep=0, chain=3, seed=42):
        return ora.philox([it, sweep & 0xffffffff, ((sweep >> 32) & 0xffff) | (move << 16) | (b << 24), chain],
                          [seed & 0xffffffff, seed >> 32])

    def u01(a, b):
        return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0

    trace = []
    pos = 0
    for move, iters in ((1, n), (2, n // 2), (3, n // 3)):
        for it in range(iters):
            w = words(move, it, 0) + words(move, it, 1)
            rec = log[pos]
            pos += 1
            evaluated = bool(rec["flags"] & 1)
            if move == 1:
                trace += [float((w[0] * n >> 32) + 1), u01(w[1], w[2]), u01(w[3], w[4])]
                if evaluated:
                    trace.append(u01(w[5], w[6]))
            elif move == 2:
                retry = [2]
                k1 = w[0] * n >> 32
                ks = [w[1] * n >> 32]
                while ks[-1] == k1:                       # free:339-341
                    ks.append(words(move, it, retry[0])[0] * n >> 32)
                    retry[0] += 1
                trace += [float(k1 + 1)] + [float(k + 1) for k in ks]
                trace += [u01(w[2], w[3]), u01(w[4], w[5])]
                if evaluated:
                    trace.append(u01(w[6], w[7]))
            else:
                retry = 2
                k1 = w[0] * n >> 32
                k2s = [w[1] * n >> 32]
                while k2s[-1] == k1:                      # free:414-416
                    k2s.append(words(move, it, retry)[0] * n >> 32)
                    retry += 1
                k3s = [w[2] * n >> 32]
                while k3s[-1] in (k1, k2s[-1]):           # free:418-420
                    k3s.append(words(move, it, retry)[1] * n >> 32)
                    retry += 1
                trace += [float(k1 + 1)] + [float(k + 1) for k in k2s] + [float(k + 1) for k in k3s]
                if evaluated:
                    trace.append(u01(w[3], w[4]))
    o2 = make(problem)
    o2.use_trace(np.array(trace))
    o2.enable_log(1000)
    o2.run_updates(1.5)
    assert o2.trace_pos() == len(trace)
    log2 = o2.log()
    assert np.array_equal(log["flags"], log2["flags"])
    assert np.array_equal(log["chi2_new"], log2["chi2_new"])       # bit-for-bit
    assert np.array_equal(o.get_state()[0], o2.get_state()[0])


def test_par2_amplitude_moves(problem):
    o = make(problem, par=2)
    o.use_philox(9, 0)
    o.adjust_windows(30, 4.0)
    o.sample(20, 4.0)
    wi, A, win = o.get_state()
    assert (A >= 0).all() and abs(A.sum() - 1.0) < 1e-12
    ar = o.accept_rates()
    assert ar[3] > 0 and ar[4] > 0
    t, init = problem
    g, c, cmin = o.get_gbar()
    ref = init["K"][wi - init["wi0"]].T @ A
    # renormalisation after each amplitude sweep is not propagated into Gbar by the reference
    assert np.abs(g - ref).max() <= 1e-9 * np.abs(ref).max()
