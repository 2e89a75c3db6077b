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


def test_log_u01_accuracy():
    rng = np.random.default_rng(7)
    us = np.concatenate([rng.random(30000), np.exp(-rng.uniform(0, 36, 20000)),
                         1.0 - np.exp(-rng.uniform(0, 36, 5000)), [2.0 ** -53, 1 - 2.0 ** -53, 0.5, 0.25]])
    us = np.floor(us * 2.0 ** 53) / 2.0 ** 53
    us = us[(us > 0) & (us < 1)]
    worst = 0.0
    for u in us:
        got = ora.log_u01(u)
        ref = math.log(u)
        worst = max(worst, abs(got - ref) / math.ulp(ref))
    assert worst <= 1.0, worst
    assert ora.log_u01(0.0) == -math.inf
    assert ora.log_u01(0.5) == math.log(0.5)


def test_chi2_tree_matches_plain_sum():
    rng = np.random.default_rng(3)
    for n in (13, 22, 40, 64, 80, 102, 128):
        gbar, g = rng.normal(size=n), rng.normal(size=n)
        sinv = rng.uniform(1, 1e6, size=n)
        ref = math.fsum(((gbar - g) * sinv) ** 2)
        assert abs(ora.chi2(gbar, g, sinv) - ref) <= 1e-14 * ref
