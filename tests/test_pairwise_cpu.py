"""CPU: oracle/pairwise.py (the order in which NumPy sums a float32 array -- what decides
the reference's noise gains, utils/reader.py:325-331,350-353) against the installed NumPy,
bit for bit."""
import numpy as np
import pytest

from oracle import pairwise as pw


@pytest.mark.parametrize("seed,n", [(0, 480000), (1, 436363), (2, 129), (3, 1000), (4, 160000), (5, 7), (6, 8191),
                                    (7, 480001), (8, 333333), (9, 99999), (10, 8193), (11, 20000), (12, 1), (13, 128)])
def test_numpy_float32_sum_and_mean_order(seed, n):
    x = (0.1 * np.random.default_rng(seed).standard_normal(n)).astype(np.float32)
    sq = x ** 2
    assert pw.sum_f32(sq) == np.sum(sq)
    assert pw.mean_f32(sq) == np.mean(sq)
    # ... and therefore the reference's rms_db
    assert 10 * np.log10(pw.mean_f32(sq)) == 10 * np.log10(np.mean(sq))
