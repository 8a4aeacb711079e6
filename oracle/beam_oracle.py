"""CPU ORACLE (test infrastructure only — never imported by the product path).

Restates the reference's joint CTC-attention (+RNN-LM) beam search, src/decode.py:64-173 (loop) and
src/decode.py:176-257 (Hypothesis), one hypothesis at a time exactly like the reference, on top of
the oracle's own decoder step (oracle/asr_oracle.py:DecodeMemory.step) and a candidate-vectorised
restatement of CTCPrefixScore.cheap_compute (src/ctc.py:76-116; checked against the scalar
oracle/decode_oracle.py in tests/test_oracle_cpu.py).

Parity pin: tests/test_beam_oracle_cpu.py runs this search on the golden models and compares the
hypotheses and per-token scores with what the REAL reference BeamDecoder produced
(tests/golden/decode.npz, decode_more.npz, decode_cfg5.npz - oracle/gen_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import asr_oracle as O

LOG_ZERO = -10000000.0          # src/ctc.py:10 (mask value of the joint score)
PREFIX_LOGZERO = -100000000.0   # src/ctc.py:20 (log-zero inside the prefix scorer)
CTC_BEAM_RATIO = 1.5            # src/decode.py:10


def prefix_init_state(x, blank=0):
    """src/ctc.py:27-35: r[t,1] = running sum of the blank log-prob (sequential float32 adds)."""
    T = x.shape[0]
    r = np.full((T, 2), PREFIX_LOGZERO, dtype=np.float32)
    r[0, 1] = x[0, blank]
    for i in range(1, T):
        r[i, 1] = r[i - 1, 1] + x[i, blank]
    return r


def prefix_cheap_compute(x, g, r_prev, candidates, blank=0, eos=1):
    """src/ctc.py:76-116 (vectorised over the candidates like the reference; float32 numpy).
    x [T,V] log-probs, g prefix (list of int), r_prev [T,2] -> (psi [C], r [C,T,2])."""
    T = x.shape[0]
    C = len(candidates)
    plen = len(g)
    last = g[-1] if plen > 0 else 0
    r = np.full((T, 2, C), PREFIX_LOGZERO, dtype=np.float32)
    start = max(1, plen)
    if plen == 0:
        r[0, 0, :] = x[0, candidates]
    psi = r[start - 1, 0, :]
    sum_prev = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
    phi = np.repeat(sum_prev[..., None], C, axis=-1)
    if plen > 0 and last in candidates:
        phi[:, candidates.index(last)] = r_prev[:, 1]
    for t in range(start, T):
        r[t, 0, :] = np.logaddexp(r[t - 1, 0, :], phi[t - 1]) + x[t, candidates]
        r[t, 1, :] = np.logaddexp(r[t - 1, 1, :], r[t - 1, 0, :]) + x[t, blank]
        psi = np.logaddexp(psi, phi[t - 1] + x[t, candidates])
    if eos in candidates:
        psi[candidates.index(eos)] = sum_prev[-1]
    return psi, np.rollaxis(r, 2)


def make_lm_state_dict(vocab_size, lm_cfg, seed=0):
    """Random reference-layout RNNLM state_dict (src/lm.py:9-24: emb, rnn.*_l{k}, trans)."""
    g = torch.Generator().manual_seed(seed)
    E, D, nl = lm_cfg['emb_dim'], lm_cfg['dim'], lm_cfg['n_layers']
    gate = 4 if lm_cfg['module'].upper() == 'LSTM' else 3
    sd = {'emb.weight': torch.randn(vocab_size, E, generator=g)}
    for l in range(nl):
        din = E if l == 0 else D
        sd['rnn.weight_ih_l%d' % l] = torch.randn(gate * D, din, generator=g) / math.sqrt(din)
        sd['rnn.weight_hh_l%d' % l] = torch.randn(gate * D, D, generator=g) / math.sqrt(D)
        sd['rnn.bias_ih_l%d' % l] = torch.randn(gate * D, generator=g) * 0.1
        sd['rnn.bias_hh_l%d' % l] = torch.randn(gate * D, generator=g) * 0.1
    if not lm_cfg['emb_tying']:
        sd['trans.weight'] = torch.randn(vocab_size, D, generator=g) / math.sqrt(D)
        sd['trans.bias'] = torch.randn(vocab_size, generator=g) * 0.1
    return sd


def lm_step(sd, lm_cfg, token, hidden):
    """RNNLM.forward on one token from `hidden` (src/lm.py:31-45 as called at src/decode.py:143-146).
    token [1] long; hidden = None | (h [nl,1,D], c [nl,1,D]) | h -> (logits [1,V], hidden)."""
    nl, D = lm_cfg['n_layers'], lm_cfg['dim']
    lstm = lm_cfg['module'].upper() == 'LSTM'
    x = sd['emb.weight'][token]                                  # [1,E]
    if hidden is None:
        h = [x.new_zeros(1, D) for _ in range(nl)]
        c = [x.new_zeros(1, D) for _ in range(nl)]
    elif lstm:
        h, c = [hidden[0][l] for l in range(nl)], [hidden[1][l] for l in range(nl)]
    else:
        h, c = [hidden[l] for l in range(nl)], None
    for l in range(nl):
        w_ih, w_hh = sd['rnn.weight_ih_l%d' % l], sd['rnn.weight_hh_l%d' % l]
        b_ih, b_hh = sd['rnn.bias_ih_l%d' % l], sd['rnn.bias_hh_l%d' % l]
        if lstm:
            g = F.linear(x, w_ih, b_ih) + F.linear(h[l], w_hh, b_hh)
            i, f, gg, o = g.chunk(4, dim=1)
            c[l] = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
            h[l] = torch.sigmoid(o) * torch.tanh(c[l])
        else:
            h[l] = O.gru_cell(F.linear(x, w_ih, b_ih), F.linear(h[l], w_hh, b_hh), h[l])
        x = h[l]
    logits = F.linear(x, sd['emb.weight']) if lm_cfg['emb_tying'] else \
        F.linear(x, sd['trans.weight'], sd['trans.bias'])
    hid = (torch.stack(h, 0), torch.stack(c, 0)) if lstm else torch.stack(h, 0)
    return logits, hid


class Hyp:
    """src/decode.py:176-257"""

    def __init__(self, dec_state, seq, scores, lm_state, ctc_state, ctc_prob, att_map):
        self.dec_state, self.seq, self.scores = dec_state, seq, scores
        self.lm_state, self.ctc_state, self.ctc_prob, self.att_map = lm_state, ctc_state, ctc_prob, att_map

    def avg(self):
        return sum(self.scores) / len(self.scores)

    def add_topk(self, topi, topv, dec_state, att_map, lm_state, ctc_state, ctc_prob, cands):
        new, term = [], None
        for i in range(len(topi)):
            if topi[i] == 1:
                term = topv[i]
                continue
            cs = cp = None
            if ctc_state is not None:
                idx = cands.index(topi[i])       # the reference raises ValueError here too
                cs, cp = ctc_state[idx], ctc_prob[idx]
            new.append(Hyp(dec_state, self.seq + [topi[i]], self.scores + [topv[i]], lm_state, cs, cp,
                           att_map))
        if term is not None:
            self.seq.append(1)
            self.scores.append(term)
            return self, new
        return None, new


def beam_search(sd, model_cfg, feat, feat_len, beam_size, min_len_ratio, max_len_ratio,
                ctc_weight=0.0, lm_weight=0.0, lm_sd=None, lm_cfg=None, lstm_impl='loop'):
    """BeamDecoder.forward (src/decode.py:64-173) -> list of (outIndex, output_scores)."""
    assert feat.shape[0] == 1
    with torch.no_grad():
        max_len = int(np.ceil(int(feat_len[0]) * max_len_ratio))
        min_len = int(np.ceil(int(feat_len[0]) * min_len_ratio))
        enc, enc_len = O.encoder_forward(sd, model_cfg['encoder'], feat, feat_len, lstm_impl=lstm_impl)
        mem = O.DecodeMemory(sd, model_cfg['attention'], model_cfg['decoder'], enc, enc_len)
        store_att = mem.mode == 'loc'
        apply_ctc, apply_lm = ctc_weight > 0, lm_weight > 0
        ctc_state = x = None
        n_cand = int(CTC_BEAM_RATIO * beam_size)
        if apply_ctc:
            x = O.ctc_head(sd, enc)[0].numpy()
            ctc_state = prefix_init_state(x)
        emb = sd['pre_embed.weight']
        h0, c0 = mem.zero_state()
        prev_top = [Hyp((h0, c0), [], [], None, ctc_state, 0, None)]
        final, nxt = [], []
        for t in range(max_len):
            for hyp in prev_top:
                tok = hyp.seq[-1] if hyp.seq else 0
                h, c = hyp.dec_state
                logits, _, attn, h, c = mem.step(emb[torch.tensor([tok])], h, c, hyp.att_map)
                cur = F.log_softmax(logits, dim=-1)
                cands = cprob = cstate = None
                if apply_ctc:
                    cands = cur[0].topk(n_cand)[1].tolist()
                    cprob, cstate = prefix_cheap_compute(x, hyp.seq, hyp.ctc_state, cands)
                    ctc_char = torch.from_numpy((cprob - hyp.ctc_prob).astype(np.float32))
                    hack = torch.full_like(cur, LOG_ZERO)
                    hack[0, cands] = ctc_char
                    cur = (1 - ctc_weight) * cur + ctc_weight * hack
                    cur[0, 0] = LOG_ZERO
                lm_state = None
                if apply_lm:
                    lm_out, lm_state = lm_step(lm_sd, lm_cfg, torch.tensor([tok]), hyp.lm_state)
                    cur = cur + lm_weight * lm_out.log_softmax(dim=-1)
                topv, topi = cur[0].topk(beam_size)
                fin, top = hyp.add_topk(topi.tolist(), topv.tolist(), (h, c), attn if store_att else None,
                                        lm_state, cstate, cprob, cands)
                if fin is not None and t >= min_len:
                    final.append(fin)
                    if beam_size == 1:
                        return [(f.seq, f.scores) for f in final]
                nxt.extend(top)
            nxt.sort(key=lambda o: o.avg(), reverse=True)
            prev_top, nxt = nxt[:beam_size], []
        final += prev_top
        final.sort(key=lambda o: o.avg(), reverse=True)
        return [(f.seq, f.scores) for f in final[:beam_size]]
