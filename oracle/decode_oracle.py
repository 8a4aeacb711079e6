"""CPU ORACLE (test infrastructure only): numpy restatement of CTC prefix scoring
(Watanabe et al. 2017, Algo. 2) as used by the reference at src/ctc.py:27-35 (init_state) and
src/ctc.py:76-116 (cheap_compute).  Pinned by tests/golden/decode.npz (outputs of the reference's
own CTCPrefixScore)."""
import numpy as np

LOGZERO = -100000000.0


def init_state(x, blank=0):
    """x [T,V] log-probs -> r [T,2]: r[t,1] = sum_{tau<=t} x[tau,blank], r[t,0] = logzero"""
    T = x.shape[0]
    r = np.full((T, 2), LOGZERO, dtype=np.float32)
    r[:, 1] = np.cumsum(x[:, blank], dtype=np.float32)
    return r


def prefix_scores(x, g, r_prev, candidates, blank=0, eos=1):
    """-> (psi [C], r [C,T,2]) for prefix g extended by each candidate"""
    T = x.shape[0]
    C = len(candidates)
    r = np.full((C, T, 2), LOGZERO, dtype=np.float32)
    psi = np.zeros(C, dtype=np.float32)
    start = max(1, len(g))
    for ci, c in enumerate(candidates):
        if len(g) == 0:
            r[ci, 0, 0] = x[0, c]
        p = r[ci, start - 1, 0]
        for t in range(start, T):
            if len(g) > 0 and c == g[-1]:
                phi = r_prev[t - 1, 1]
            else:
                phi = np.logaddexp(r_prev[t - 1, 0], r_prev[t - 1, 1])
            r[ci, t, 0] = np.logaddexp(r[ci, t - 1, 0], phi) + x[t, c]
            r[ci, t, 1] = np.logaddexp(r[ci, t - 1, 1], r[ci, t - 1, 0]) + x[t, blank]
            p = np.logaddexp(p, phi + x[t, c])
        if c == eos:
            p = np.logaddexp(r_prev[-1, 0], r_prev[-1, 1])
        psi[ci] = p
    return psi, r
