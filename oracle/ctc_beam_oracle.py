"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's pure-CTC prefix beam search
(`/root/reference/src/ctc.py:118-352`: CTCHypothesis + the frame loop of CTCBeamDecoder.forward), taking the
CTC log-probabilities and an LM-step callback instead of a model, so that the SEARCH can be checked on its own.
Plain Python objects, deep copies and string sorts exactly like the reference (same numpy promotion flow:
float32 inputs, python-float / float64 accumulation).  Pinned by tests/test_ctc_beam_oracle_cpu.py on golden
hypotheses produced by RUNNING the real reference class (oracle/gen_golden.py --ctc-beam-big).
Never imported by the product path."""
import copy

import numpy as np

LOG_ZERO = -10000000.0          # src/ctc.py:9


class Hyp:
    """src/ctc.py:118-208"""

    def __init__(self):
        self.y = []
        self.Pr_y_t_blank = 0.0
        self.Pr_y_t_nblank = LOG_ZERO
        self.Pr_y_t_blank_bkup = 0.0
        self.Pr_y_t_nblank_bkup = LOG_ZERO
        self.lm_output = None
        self.lm_hidden = None
        self.updated_lm = False

    def update_lm(self, output, hidden):
        self.lm_output, self.lm_hidden, self.updated_lm = output, hidden, True

    def get_string(self):                                           # src/ctc.py:144-146
        return ''.join([str(s) for s in self.y])

    def get_score(self):
        return np.logaddexp(self.Pr_y_t_blank, self.Pr_y_t_nblank)

    def get_final_score(self):                                      # src/ctc.py:151-155
        if len(self.y) > 0:
            return np.logaddexp(self.Pr_y_t_blank, self.Pr_y_t_nblank) / len(self.y)
        return np.logaddexp(self.Pr_y_t_blank, self.Pr_y_t_nblank)

    def check_same(self, y_2):
        return len(self.y) == len(y_2) and all(a == b for a, b in zip(self.y, y_2))

    def update_Pr_nblank(self, ctc_y_t):                            # src/ctc.py:165-168
        self.Pr_y_t_nblank += ctc_y_t

    def update_Pr_nblank_prefix(self, ctc_y_t, pb_prefix, pnb_prefix, Pr_ye_y=None):   # src/ctc.py:170-182
        lm_prob = Pr_ye_y if Pr_ye_y is not None else 0.0
        if len(self.y) == 0:
            return
        if len(self.y) == 1:
            v = ctc_y_t + lm_prob + np.logaddexp(pb_prefix, pnb_prefix)
        else:
            v = ctc_y_t + lm_prob + (pb_prefix if self.y[-1] == self.y[-2]
                                     else np.logaddexp(pb_prefix, pnb_prefix))
        self.Pr_y_t_nblank = np.logaddexp(self.Pr_y_t_nblank, v)

    def update_Pr_blank(self, ctc_blank_t):                         # src/ctc.py:184-186
        self.Pr_y_t_blank = np.logaddexp(self.Pr_y_t_nblank_bkup, self.Pr_y_t_blank_bkup) + ctc_blank_t

    def add_token(self, token, ctc_token_t, Pr_k_y=None):           # src/ctc.py:188-204
        lm_prob = Pr_k_y if Pr_k_y is not None else 0.0
        if len(self.y) == 0:
            new = ctc_token_t + lm_prob + np.logaddexp(self.Pr_y_t_blank_bkup, self.Pr_y_t_nblank_bkup)
        else:
            new = ctc_token_t + lm_prob + (self.Pr_y_t_blank_bkup if self.y[-1] == token else
                                           np.logaddexp(self.Pr_y_t_blank_bkup, self.Pr_y_t_nblank_bkup))
        self.Pr_y_t_blank = LOG_ZERO
        self.Pr_y_t_nblank = new
        self.Pr_y_t_blank_bkup = self.Pr_y_t_blank
        self.Pr_y_t_nblank_bkup = self.Pr_y_t_nblank
        self.y.append(token)

    def orig_backup(self):
        self.Pr_y_t_blank_bkup = self.Pr_y_t_blank
        self.Pr_y_t_nblank_bkup = self.Pr_y_t_nblank


def prefix_beam_search(ctc_output, vocab_range, beam_size, vocab_cand, lm_step=None, lm_w=0.0):
    """src/ctc.py:250-352.  ctc_output [T, V] float32 numpy (what the reference has after its log_softmax);
    lm_step(token, hidden) -> (log-probs [V] float32 numpy, hidden) or None.  Returns [b.y for b in B]."""
    ctc_output = np.asarray(ctc_output, dtype=np.float32)
    T = len(ctc_output)
    apply_lm = lm_step is not None and lm_w > 0
    B = [Hyp()]
    if apply_lm:
        B[0].update_lm(*lm_step(0, None))                           # 0 == <sos> for RNNLM
    start = True
    for t in range(T):
        if np.argmax(ctc_output[t]) == 0 and start:                 # src/ctc.py:265-268
            continue
        start = False
        B_new = []
        for i in range(len(B)):
            B_i_new = copy.deepcopy(B[i])
            if len(B_i_new.y) > 0:
                if B_i_new.y[-1] == 1:                              # <eos>
                    B_new.append(B_i_new)
                    continue
                B_i_new.update_Pr_nblank(ctc_output[t, B_i_new.y[-1]])
                for j in range(len(B)):
                    if i != j and B[j].check_same(B_i_new.y[:-1]):
                        lm_prob = 0.0
                        if apply_lm:
                            lm_prob = lm_w * B[j].lm_output[B_i_new.y[-1]]
                        B_i_new.update_Pr_nblank_prefix(ctc_output[t, B_i_new.y[-1]], B[j].Pr_y_t_blank,
                                                        B[j].Pr_y_t_nblank, lm_prob)
                        break
            B_i_new.update_Pr_blank(ctc_output[t, 0])
            lm_probs = B_i_new.lm_output if apply_lm else None
            if apply_lm:
                cand = sorted(zip(vocab_range, ctc_output[t, vocab_range] + lm_w * lm_probs[vocab_range]),
                              reverse=True, key=lambda x: x[1])
            else:
                cand = sorted(zip(vocab_range, ctc_output[t, vocab_range]), reverse=True, key=lambda x: x[1])
            for j in range(vocab_cand):
                k = cand[j][0]
                hyp_yk = copy.deepcopy(B_i_new)
                lm_prob = 0.0 if not apply_lm else lm_w * lm_probs[k]
                hyp_yk.add_token(k, ctc_output[t, k], lm_prob)
                hyp_yk.updated_lm = False
                B_new.append(hyp_yk)
            B_i_new.orig_backup()
            B_new.append(B_i_new)
        B_new = sorted(B_new, key=lambda x: x.get_string())          # src/ctc.py:320-329
        B = [B_new[0]]
        for i in range(1, len(B_new)):
            if B_new[i].check_same(B[-1].y):
                if B_new[i].get_score() > B[-1].get_score():
                    B[-1] = B_new[i]
                continue
            B.append(B_new[i])
        if t == T - 1:
            B = sorted(B, reverse=True, key=lambda x: x.get_final_score())
        else:
            B = sorted(B, reverse=True, key=lambda x: x.get_score())
        if len(B) > beam_size:
            B = B[:beam_size]
        if apply_lm and t < T - 1:                                  # src/ctc.py:342-350
            for h in B:
                if len(h.y) > 0 and not h.updated_lm:
                    h.update_lm(*lm_step(h.y[-1], h.lm_hidden))
    return [b.y for b in B]
