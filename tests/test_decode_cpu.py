"""CPU: CTC prefix-score oracle vs the reference's CTCPrefixScore outputs (golden)."""
import numpy as np

from oracle import decode_oracle as DO
from helpers import load_golden, rel_err


def test_prefix_score_oracle_matches_reference():
    g = load_golden("decode")
    x = g["ps_x"][0]
    r0 = DO.init_state(x)
    assert np.allclose(r0, g["ps_r0"], rtol=1e-6, atol=1e-6)
    psi1, r1 = DO.prefix_scores(x, [], r0, [3, 1, 5, 8])
    assert rel_err(psi1, g["ps_psi1"]) < 1e-5 and rel_err(r1, g["ps_r1"]) < 1e-5
    psi2, r2 = DO.prefix_scores(x, [3], r1[0], [3, 4, 1, 2])
    assert rel_err(psi2, g["ps_psi2"]) < 1e-5 and rel_err(r2, g["ps_r2"]) < 1e-5
    psi3, r3 = DO.prefix_scores(x, [3, 3], r2[0], [1, 7, 3])
    assert rel_err(psi3, g["ps_psi3"]) < 1e-5 and rel_err(r3, g["ps_r3"]) < 1e-5


def test_prefix_score_oracle_full_compute_matches_reference():
    """full_compute: every token as the continuation, no <eos> override (src/ctc.py:37-74)"""
    g = load_golden("prefix_full")
    x = g["x"][0]
    V = x.shape[1]
    r = DO.init_state(x)
    assert np.allclose(r, g["r0"], rtol=1e-6, atol=1e-6)
    for k, (prefix, pick) in enumerate([([], 3), ([3], 3), ([3, 3], 7), ([3, 3, 7], None)], 1):
        psi, rn = DO.prefix_scores(x, prefix, r, list(range(V)), eos=-1)
        assert rel_err(psi, g["psi%d" % k]) < 1e-5 and rel_err(rn, g["r%d" % k]) < 1e-5, k
        if pick is not None:
            r = rn[pick]
