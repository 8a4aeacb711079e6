"""Joint CTC-attention (+RNN-LM) beam search — MI355X mirror of the reference's src/decode.py
(BeamDecoder / Hypothesis, same constructor arguments, same scoring rules, returns Hypothesis
objects with `.outIndex` / `.output_scores`).

The reference advances ONE hypothesis at a time (batch = 1, state ping-pong through the CPU,
numpy prefix scoring per hypothesis: src/decode.py:103-162).  Here every live hypothesis of the
utterance is a row of one device batch: one attention step, one decoder-cell step, one vocabulary
projection, ONE CTC prefix-score launch for all (hypothesis, candidate) pairs and one LM step per
decode position; only the final top-k bookkeeping (<= beam^2 scalars) is host logic.
"""
import numpy as np
import torch
import yaml
from torch import nn

from .. import ops
from .. import decoder_ops as dops
from .. import gru_ops
from .lm import RNNLM
from .ctc import CTCPrefixScore, LOG_ZERO

CTC_BEAM_RATIO = 1.5   # (reference: src/decode.py:10)


class BeamDecoder(nn.Module):
    ''' Beam decoder for ASR (reference: src/decode.py:13-173) '''

    def __init__(self, asr, emb_decoder, beam_size, min_len_ratio, max_len_ratio,
                 lm_path='', lm_config='', lm_weight=0.0, ctc_weight=0.0):
        super().__init__()
        self.beam_size = beam_size
        self.min_len_ratio = min_len_ratio
        self.max_len_ratio = max_len_ratio
        self.asr = asr
        assert self.asr.enable_att

        self.apply_ctc = ctc_weight > 0
        if self.apply_ctc:
            assert self.asr.ctc_weight > 0, 'ASR was not trained with CTC decoder'
            self.ctc_w = ctc_weight
            self.ctc_beam_size = int(CTC_BEAM_RATIO * self.beam_size)

        self.apply_lm = lm_weight > 0
        if self.apply_lm:
            self.lm_w = lm_weight
            self.lm_path = lm_path
            lm_config = yaml.load(open(lm_config, 'r'), Loader=yaml.FullLoader)
            self.lm = RNNLM(self.asr.vocab_size, **lm_config['model'])
            self.lm.load_state_dict(torch.load(self.lm_path, map_location='cpu')['model'])
            self.lm.eval()

        self.apply_emb = emb_decoder is not None
        if self.apply_emb:
            raise NotImplementedError('embedding-fusion decoding (src/plugin.py) is out of scope')

    def create_msg(self):
        msg = ['Decode spec| Beam size = {}\t| Min/Max len ratio = {}/{}'.format(
            self.beam_size, self.min_len_ratio, self.max_len_ratio)]
        if self.apply_ctc:
            msg.append('           |Joint CTC decoding enabled \t| weight = {:.2f}\t'.format(self.ctc_w))
        if self.apply_lm:
            msg.append('           |Joint LM decoding enabled \t| weight = {:.2f}\t| src = {}'.format(
                self.lm_w, self.lm_path))
        return msg

    @torch.no_grad()
    def forward(self, audio_feature, feature_len):
        assert audio_feature.shape[0] == 1, "Batchsize == 1 is required for beam search"
        asr = self.asr
        device = audio_feature.device
        dec, att = asr.decoder, asr.attention
        N = att.num_head
        max_output_len = int(np.ceil(feature_len.cpu().item() * self.max_len_ratio))
        min_output_len = int(np.ceil(feature_len.cpu().item() * self.min_len_ratio))
        store_att = att.mode == 'loc'

        encode_feature, encode_len = asr.encoder(audio_feature, feature_len)
        T = encode_feature.shape[1]
        att.reset_mem()
        base_tape, _, _, _ = att.build_memory(encode_feature, encode_len)
        tapes = {1: base_tape}

        ctc_prefix, ctc_state0 = None, None
        if self.apply_ctc:
            ctc_output = ops.log_softmax(ops.linear(encode_feature, asr.ctc_layer.weight,
                                                    asr.ctc_layer.bias))
            ctc_prefix = CTCPrefixScore(ctc_output)
            ctc_state0 = ctc_prefix.init_state_device()

        zeros = lambda: torch.zeros((dec.layer, 1, dec.dim), device=device)
        lstm_dec = dec.enable_cell                      # GRU decoders carry h only (state = (h, h))
        prev_top = [Hypothesis(decoder_state=(zeros(), zeros()), output_seq=[], output_scores=[],
                               lm_state=None, ctc_prob=0.0, ctc_state=ctc_state0, att_map=None)]
        final_hypothesis, next_top = [], []
        if self.apply_lm:
            self.lm.to(device)

        for t in range(max_output_len):
            n = len(prev_top)
            if n not in tapes:
                tapes[n] = dops.expand_tape(base_tape, n)
            tape = tapes[n]
            # ---- gather the live hypotheses into one batch
            prev_token = torch.tensor([h.last_token for h in prev_top], dtype=torch.long, device=device)
            h_dec = torch.cat([h.decoder_state[0] for h in prev_top], dim=1)      # [layers,n,dim]
            c_dec = torch.cat([h.decoder_state[1] for h in prev_top], dim=1)
            prev_att = None
            if store_att:
                maps = [h.att_map if h.att_map is not None
                        else att.att_layer.uniform_init(1, T, device) for h in prev_top]
                prev_att = torch.cat(maps, dim=0)                                  # [n,N,T]
            # ---- attention + decoder step for all hypotheses (src/decode.py:110-121)
            query = h_dec[0] if dec.layer == 1 else h_dec.transpose(0, 1).reshape(n, -1)
            q = ops.tanh(ops.linear(query, att.proj_q.weight, att.proj_q.bias)).view(n * N, att.dim)
            attn, context = dops.attn_step_infer(tape, q, prev_att)
            if N > 1:
                context = ops.linear(context.view(n, N * att.v_dim), att.merge_head.weight,
                                     att.merge_head.bias)
            x = dops.concat_last(dops.embedding(prev_token, asr.pre_embed.weight), context)
            hs, cs = [], []
            for l in range(dec.layer):
                if lstm_dec:
                    hl, cl = dops.lstm_cell_infer(x, h_dec[l], c_dec[l], *dec.layers.layer_params(l))
                else:
                    hl = gru_ops.gru_cell_infer(x, h_dec[l], *dec.layers.layer_params(l))
                    cl = hl
                hs.append(hl)
                cs.append(cl)
                x = hl
            h_new, c_new = torch.stack(hs, 0), torch.stack(cs, 0)
            cur_prob = ops.log_softmax(ops.linear(x, dec.char_trans.weight, dec.char_trans.bias))

            # ---- CTC prefix scoring on limited candidates (src/decode.py:123-138)
            cand_host, psi, r_new = None, None, None
            if self.apply_ctc:
                _, cand = ops.topk(cur_prob, self.ctc_beam_size)                   # [n,C]
                r_prev = torch.stack([h.ctc_state for h in prev_top], 0)           # [n,T,2]
                plen = [len(h.output_seq) for h in prev_top]
                psi, r_new = ctc_prefix.cheap_compute_batch(plen, [h.last_token for h in prev_top],
                                                            r_prev, cand)
                prev_ctc = torch.tensor([h.ctc_prob for h in prev_top], dtype=torch.float32,
                                        device=device).unsqueeze(1)
                hack = torch.full_like(cur_prob, LOG_ZERO)
                hack.scatter_(1, cand, psi - prev_ctc)
                cur_prob = (1 - self.ctc_w) * cur_prob + self.ctc_w * hack
                cur_prob[:, 0] = LOG_ZERO                                          # ignore <sos>
                cand_host = cand.cpu().tolist()

            # ---- joint RNN-LM decoding (src/decode.py:140-148)
            lm_h = lm_c = None
            if self.apply_lm:
                hidden = None
                lm_lstm = self.lm.rnn_type == 'LSTM'
                if prev_top[0].lm_state is not None:
                    hidden = (torch.cat([h.lm_state[0] for h in prev_top], dim=1),
                              torch.cat([h.lm_state[1] for h in prev_top], dim=1))
                    if not lm_lstm:
                        hidden = hidden[0]
                lm_out, lm_hid = self.lm(prev_token.unsqueeze(1), torch.ones([n]), hidden=hidden)
                lm_h, lm_c = lm_hid if lm_lstm else (lm_hid, lm_hid)
                cur_prob = cur_prob + self.lm_w * ops.log_softmax(lm_out[:, 0, :])

            # ---- beam bookkeeping on the host (src/decode.py:150-167)
            topv, topi = ops.topk(cur_prob, self.beam_size)
            topv_h, topi_h = topv.cpu().tolist(), topi.cpu().tolist()
            psi_h = psi.cpu().tolist() if psi is not None else None
            for i, hyp in enumerate(prev_top):
                state_i = (h_new[:, i:i + 1], c_new[:, i:i + 1])
                att_i = attn[i:i + 1] if store_att else None
                lm_i = (lm_h[:, i:i + 1], lm_c[:, i:i + 1]) if self.apply_lm else None
                final, top = hyp.addTopk(topi_h[i], topv_h[i], state_i, att_map=att_i, lm_state=lm_i,
                                         ctc_state=r_new[i] if r_new is not None else None,
                                         ctc_prob=psi_h[i] if psi_h is not None else 0.0,
                                         ctc_candidates=cand_host[i] if cand_host is not None else [])
                if final is not None and (t >= min_output_len):
                    final_hypothesis.append(final)
                    if self.beam_size == 1:
                        return final_hypothesis
                next_top.extend(top)

            next_top.sort(key=lambda o: o.avgScore(), reverse=True)
            prev_top = next_top[:self.beam_size]
            next_top = []

        final_hypothesis += prev_top
        final_hypothesis.sort(key=lambda o: o.avgScore(), reverse=True)
        return final_hypothesis[:self.beam_size]


class Hypothesis:
    ''' Hypothesis for beam search decoding (reference: src/decode.py:176-257): history of labels
        and scores plus the decoder / LM / CTC / attention state needed to extend it.  States stay
        on the device (the reference ping-pongs them through the CPU). '''

    def __init__(self, decoder_state, output_seq, output_scores, lm_state, ctc_state, ctc_prob, att_map):
        assert len(output_seq) == len(output_scores)
        self.decoder_state = decoder_state
        self.att_map = att_map
        self.lm_state = lm_state
        self.output_seq = output_seq
        self.output_scores = output_scores
        self.ctc_state = ctc_state
        self.ctc_prob = ctc_prob

    @property
    def last_token(self):
        return self.output_seq[-1] if len(self.output_seq) != 0 else 0

    def avgScore(self):
        ''' averaged log probability of the hypothesis '''
        assert len(self.output_scores) != 0
        return sum(self.output_scores) / len(self.output_scores)

    def addTopk(self, topi, topv, decoder_state, att_map=None, lm_state=None, ctc_state=None,
                ctc_prob=0.0, ctc_candidates=[]):
        ''' Expand the hypothesis with its top-k continuations; <eos>=1 finalises it
            (src/decode.py:209-239) '''
        new_hypothesis = []
        term_score = None
        for i in range(len(topi)):
            if topi[i] == 1:
                term_score = topv[i]
                continue
            idxes = self.output_seq[:] + [topi[i]]
            scores = self.output_scores[:] + [topv[i]]
            ctc_s, ctc_p = None, None
            if ctc_state is not None:
                idx = ctc_candidates.index(topi[i])
                ctc_s = ctc_state[idx]
                ctc_p = ctc_prob[idx]
            new_hypothesis.append(Hypothesis(decoder_state, output_seq=idxes, output_scores=scores,
                                             lm_state=lm_state, ctc_state=ctc_s, ctc_prob=ctc_p,
                                             att_map=att_map))
        if term_score is not None:
            self.output_seq.append(1)
            self.output_scores.append(term_score)
            return self, new_hypothesis
        return None, new_hypothesis

    @property
    def outIndex(self):
        return [int(i) for i in self.output_seq]
