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


def test_vectorised_beam_bookkeeping_equals_the_record_loop():
    """BeamDecoder._select_survivors (ONE numpy pass over the rows of all utterances, used by forward_batch per decode
    position) against _expand_beam per utterance (the record loop pinned on the reference's hypotheses through
    forward(), src/decode.py:150-167, 209-239): same survivors in the same order, same parents, candidate columns,
    scores and CTC prefix probabilities over 300 random positions with tied scores, duplicate labels, <eos> among the
    top-k, labels missing from the CTC candidates, several utterances and beam 1"""
    import importlib
    import random
    from conftest import PKG_NAME
    D = importlib.import_module(PKG_NAME + ".src.decode")
    for trial in range(300):
        rng = random.Random(trial)
        nr = np.random.RandomState(trial)
        d = object.__new__(D.BeamDecoder)
        d.beam_size = rng.choice([1, 2, 3, 5])
        d.apply_ctc = rng.random() < 0.7
        B = d.beam_size
        C = int(1.5 * B) if d.apply_ctc else 0
        t = rng.randint(0, 4)
        n_utt = rng.randint(1, 4)
        utt, hyps, rows = [], [], []
        for u in range(n_utt):
            for _ in range(rng.randint(1, B)):
                sc = [float(np.float32(rng.uniform(-3, 0))) for _ in range(t)]
                hyps.append(D.Hypothesis(None, [rng.randint(3, 9) for _ in range(t)], sc, None, None, 0.0, None))
                utt.append(u * 3 + 1)
                toks = nr.choice(np.arange(1, 9), B, replace=False) if rng.random() < 0.8 else nr.randint(1, 9, B)
                scs = np.sort(nr.uniform(-3, 0, B).astype(np.float32))[::-1]
                if rng.random() < 0.3:
                    scs[1:] = scs[0]                                        # ties
                r = list(map(float, scs)) + list(map(float, toks))
                if d.apply_ctc:
                    r += list(map(float, nr.uniform(-5, 0, C).astype(np.float32))) + list(map(float, nr.randint(1, 9, C)))
                rows.append(r)
        packed = np.asarray(rows, dtype=np.float64)
        utt = np.asarray(utt)
        ssum = np.array([h.score_sum for h in hyps], dtype=np.float64)
        sel_rows, sel_k, sel_col, sel_ctc, is_eos = d._select_survivors(
            packed[:, :B], packed[:, B:2 * B].astype(np.int64), packed, utt, ssum, t, C)
        got = {}
        for j in range(len(sel_rows)):
            i, k = int(sel_rows[j]), int(sel_k[j])
            got.setdefault(int(utt[i]), []).append(
                (i, int(packed[i, B + k]), float(packed[i, k]), int(sel_col[j]),
                 float(sel_ctc[j]) if sel_ctc is not None else None))
        for u in sorted(set(utt.tolist())):
            idx = np.flatnonzero(utt == u)
            import copy
            prev = [copy.deepcopy(hyps[i]) for i in idx]
            nxt, _ = d._expand_beam(prev, [rows[i] for i in idx], t, 10 ** 9, [], C)
            ref = [(int(idx[h.parent]), h.output_seq[-1], h.output_scores[-1], h.cand, h.ctc_prob) for h in nxt]
            assert got.get(u, []) == ref, (trial, u)
            for loc, i in enumerate(idx):                                   # the <eos> mask = "this hypothesis finishes"
                ended = any(int(v) == 1 for v in rows[i][B:2 * B])
                assert bool(is_eos[i].any()) == ended
