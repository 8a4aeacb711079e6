"""CPU: the numpy Philox4x32-10 restatement (oracle/regularizer_oracle.py) against the published
Random123 known-answer vectors, and basic sanity of the derived keep mask."""
import numpy as np

from oracle import regularizer_oracle as R


def test_philox_known_answers():
    for ctr, key, expect in R.KAT:
        out = R.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in out) == expect


def test_keep_mask_rate_and_determinism():
    m1 = R.keep_mask(200000, 0.25, seed=2 ** 50 + 7)
    m2 = R.keep_mask(200000, 0.25, seed=2 ** 50 + 7)
    m3 = R.keep_mask(200000, 0.25, seed=2 ** 50 + 8)
    assert np.array_equal(m1, m2) and not np.array_equal(m1, m3)
    assert abs(m1.mean() - 0.75) < 5e-3
    assert R.keep_mask(64, 0.0, 1).all()
    y = R.dropout(np.ones(1000, np.float32), 0.5, 3)
    assert set(np.unique(y)) <= {0.0, 2.0}
