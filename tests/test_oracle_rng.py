"""Known-answer and accuracy tests for the oracle's draw source and exp."""
import math

import numpy as np

from oracle import oracle as ora


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert ora.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ora.philox([0xffffffff] * 4, [0xffffffff] * 2) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ora.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                      [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_exp_neg_accuracy():
    rng = np.random.default_rng(7)
    xs = np.concatenate([-np.exp(rng.uniform(-30, math.log(707.0), 20000)),
                         -rng.uniform(0, 40, 20000)])
    worst = 0.0
    for x in xs:
        got = ora.exp_neg(x)
        ref = math.exp(x)
        ulp = math.ulp(ref)
        worst = max(worst, abs(got - ref) / ulp)
    assert worst <= 1.0, worst
    assert ora.exp_neg(-708.0) == 0.0 and ora.exp_neg(-1e9) == 0.0
    assert ora.exp_neg(float("nan")) == 0.0
    assert ora.exp_neg(-1e-300) == 1.0


def test_chi2_tree_matches_plain_sum():
    rng = np.random.default_rng(3)
    for n in (13, 22, 40, 64, 80, 102, 128):
        gbar, g = rng.normal(size=n), rng.normal(size=n)
        sinv = rng.uniform(1, 1e6, size=n)
        ref = math.fsum(((gbar - g) * sinv) ** 2)
        assert abs(ora.chi2(gbar, g, sinv) - ref) <= 1e-14 * ref
