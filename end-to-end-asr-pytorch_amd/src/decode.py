"""Joint CTC-attention (+RNN-LM) beam search — MI355X mirror of the reference's src/decode.py
(BeamDecoder / Hypothesis, same constructor arguments, same scoring rules, returns Hypothesis
objects with `.outIndex` / `.output_scores`).

The reference advances ONE hypothesis at a time (batch = 1, state ping-pong through the CPU,
numpy prefix scoring per hypothesis: src/decode.py:103-162).  Here every live hypothesis of the
utterance is a row of one device batch: one attention step, one decoder-cell step, one vocabulary
projection, ONE CTC prefix-score launch for all (hypothesis, candidate) pairs and one LM step per
decode position; only the final top-k bookkeeping (<= beam^2 scalars) is host logic.
"""
import ctypes
import os

import numpy as np
import torch
import yaml
from torch import nn

from .. import _lib
from .. import ops
from .. import decoder_ops as dops
from .. import gru_ops
from .. import speller_ops as sops
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
        if self.batchable() and os.environ.get("ASRK_DECODE_HOST_BEAM", "0") != "1":
            # single-head location-aware attention + one-layer decoder: the device-resident loop of forward_batch
            # (no read-back per position) with one utterance
            return self.forward_batch(audio_feature, feature_len)[0]
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
        # single-head location-aware attention + one-layer LSTM / GRU decoder: one fused C call per step
        # (csrc/speller.hip) over ONE copy of the utterance's key / value for all hypotheses
        fused = sops.supported_loop(att, dec)
        steppers = {}
        if fused:
            enc_len_dev = encode_len.to(device)
            s_key = ops.tanh(ops.linear(encode_feature, att.proj_k.weight, att.proj_k.bias))
            s_value = ops.tanh(ops.linear(encode_feature, att.proj_v.weight, att.proj_v.bias)) \
                if att.v_proj else encode_feature

            def stepper(n):
                if n not in steppers:
                    steppers[n] = sops.SpellerStepper(att, dec, s_key, s_value, enc_len_dev, n, shared=True)
                return steppers[n]
        else:
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
        # Hypotheses are host-side bookkeeping only (labels, scores, parent row, candidate column);
        # the states of the live beam are ROWS of a few device tensors that are re-gathered once per
        # step with the surviving (parent, candidate) indices -- no per-hypothesis tensor slicing.
        prev_top = [Hypothesis(decoder_state=None, output_seq=[], output_scores=[],
                               lm_state=None, ctc_prob=0.0, ctc_state=None, att_map=None)]
        h_dec, c_dec = zeros(), zeros()                                           # [layers,n,dim]
        if store_att:                                                             # [n,N,T]
            if att.att_layer.k_len is None:
                att.att_layer.compute_mask(encode_feature, encode_len.to(device))
            prev_att = att.att_layer.uniform_init(1, T, device)
        else:
            prev_att = None
        if fused:
            stepper(1).h[0].zero_()
            stepper(1).c[0].zero_()
        lm_hidden = None
        r_prev = ctc_state0.unsqueeze(0) if self.apply_ctc else None              # [n,T,2]
        final_hypothesis, next_top = [], []
        if self.apply_lm:
            self.lm.to(device)

        lm_lstm = self.apply_lm and self.lm.rnn_type == 'LSTM'
        V = asr.vocab_size
        C = self.ctc_beam_size if self.apply_ctc else 0
        # per-step host -> device traffic is ONE small tensor (rows: last token, prefix length, parent row,
        # candidate column of every live hypothesis, + their CTC prefix scores), device -> host ONE
        # packed tensor (top-k values / labels, candidate labels / prefix scores): a single sync per step
        meta = torch.zeros((4, 1), dtype=torch.int64)
        prev_ctc_h = torch.zeros((1,), dtype=torch.float32)
        for t in range(max_output_len):
            n = len(prev_top)
            meta_d = meta.to(device, non_blocking=True)
            prev_token, plen_d, pi, ci = meta_d[0], meta_d[1], meta_d[2], meta_d[3]
            if t > 0:
                # ---- the survivors' states: one gather per state tensor (parents = rows of step t-1)
                if fused:                       # straight into this step's state slots
                    st = stepper(n)
                    torch.index_select(h_new[0], 0, pi, out=st.h[0])
                    torch.index_select(c_new[0], 0, pi, out=st.c[0])
                else:
                    h_dec, c_dec = h_new.index_select(1, pi), c_new.index_select(1, pi)
                if store_att:
                    prev_att = attn.index_select(0, pi)
                if self.apply_lm:
                    lm_hidden = (lm_h.index_select(1, pi), lm_c.index_select(1, pi)) if lm_lstm \
                        else lm_h.index_select(1, pi)
                if self.apply_ctc:
                    r_prev = r_new[pi, ci]                                         # [n,T,2]
            # ---- attention + decoder step for all hypotheses (src/decode.py:110-121)
            if fused:
                st = stepper(n)         # the entering state sits in st.h[0] / st.c[0]
                attn, context, x, c_top = st.step(dops.embedding(prev_token, asr.pre_embed.weight), prev_att)
                h_new, c_new = x.unsqueeze(0), c_top.unsqueeze(0)
            else:
                if n not in tapes:
                    tapes[n] = dops.expand_tape(base_tape, n)
                tape = tapes[n]
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
            att_logp = ops.log_softmax(ops.linear(x, dec.char_trans.weight, dec.char_trans.bias))

            # ---- CTC prefix scoring on limited candidates (src/decode.py:123-138)
            cand, psi, r_new, prev_ctc = None, None, None, None
            if self.apply_ctc:
                _, cand = ops.topk(att_logp, C)                                    # [n,C]
                psi, r_new = ctc_prefix.cheap_compute_batch(plen_d, prev_token, r_prev, cand)
                prev_ctc = prev_ctc_h.to(device, non_blocking=True)
            # ---- joint RNN-LM decoding (src/decode.py:140-148)
            lm_h = lm_c = lm_logp = None
            if self.apply_lm:
                lm_out, lm_hid = self.lm(prev_token.unsqueeze(1), None, hidden=lm_hidden)
                lm_h, lm_c = lm_hid if lm_lstm else (lm_hid, lm_hid)
                lm_logp = ops.log_softmax(lm_out[:, 0, :])
            if self.apply_ctc or self.apply_lm:
                cur_prob = dops.joint_score(att_logp, cand, psi, prev_ctc, lm_logp,
                                            self.ctc_w if self.apply_ctc else 0.0,
                                            self.lm_w if self.apply_lm else 0.0, LOG_ZERO)
            else:
                cur_prob = att_logp

            # ---- beam bookkeeping on the host (src/decode.py:150-167)
            topv, topi = ops.topk(cur_prob, self.beam_size)
            B_ = self.beam_size
            parts = [topv, topi.to(torch.float32)]                  # labels < 2^24: exact in f32
            if self.apply_ctc:
                parts += [psi, cand.to(torch.float32)]
            packed = torch.cat(parts, dim=1).cpu().tolist()         # the step's only sync
            next_top, done = self._expand_beam(prev_top, packed, t, min_output_len, final_hypothesis, C)
            if done:
                return final_hypothesis
            prev_top = next_top
            next_top = []
            if not prev_top:
                break
            meta = torch.tensor([[h.last_token for h in prev_top], [len(h.output_seq) for h in prev_top],
                                 [h.parent for h in prev_top], [h.cand for h in prev_top]], dtype=torch.int64)
            if self.apply_ctc:
                prev_ctc_h = torch.tensor([h.ctc_prob for h in prev_top], dtype=torch.float32)

        final_hypothesis += prev_top
        final_hypothesis.sort(key=lambda o: o.avgScore(), reverse=True)
        return final_hypothesis[:self.beam_size]


    # ------------------------------------------------------------------------------------------------------------
    def batchable(self):
        ''' forward_batch decodes several utterances per device step (else it falls back to one forward() each):
            single-head location-aware attention + one-layer LSTM decoder (the fused step kernels) behind an encoder
            that can encode a padded batch utterance-exactly (Encoder.supports_packed) '''
        asr = self.asr
        return sops.supported_loop(asr.attention, asr.decoder) and asr.encoder.supports_packed() \
            and asr.vocab_size < (1 << 24) and self.beam_size <= 32 \
            and (not self.apply_ctc or self.ctc_beam_size <= 48)        # asrk_beam_select_f32's limits

    @torch.no_grad()
    def forward_batch(self, audio_feature, feature_len):
        ''' Beam search over U utterances AT ONCE: audio_feature [U,Tmax,D] (zero-padded), feature_len [U] ->
            list of U hypothesis lists, each what forward() returns for that utterance alone.

            The reference parallelises decoding over utterances with CPU worker processes (bin/test_asr.py:163-167,
            joblib.Parallel around a batch-1 decoder, src/decode.py:64); on one GPU a single utterance's <= beam_size
            hypotheses leave the chip idle (a decode step is ~0.7 ms of launch-bound kernels).  Here the live
            hypotheses of ALL utterances are rows of one device batch: one packed encoder pass (every utterance
            encoded exactly as if alone and unpadded), then per decode position ONE attention + decoder step, one
            vocabulary projection, one CTC prefix-score launch, one LM step and one read-back for all of them; rows
            carry the index of the utterance whose encoder memory / CTC posteriors they use (row_mem).  Per-utterance
            bookkeeping (_expand_beam), length limits and termination are forward()'s.

            "What forward() returns" holds up to f32 SUMMATION ORDER: the key / value / CTC / vocabulary projections run
            as GEMMs over [U * Te] or [rows] rows instead of [T] or [beam] rows, and the kernel a shape selects (tile
            form, split-K) fixes the order in which a dot product is added up, so scores agree with batch-1 runs to
            ~1e-6 relative, not bit for bit; two hypotheses whose scores tie closer than that may swap.  A zero-frame
            utterance (shorter than the encoder's time reduction) scores logzero on every CTC path.
            ASRK_DETERMINISTIC=1 pins the GEMM forms that do not depend on the row count (no split-K). '''
        U = audio_feature.shape[0]
        lens_h = [int(v) for v in torch.as_tensor(feature_len).cpu().tolist()]
        if not self.batchable():
            return [self.forward(audio_feature[u:u + 1, :lens_h[u]].contiguous(),
                                 torch.as_tensor(feature_len)[u:u + 1]) for u in range(U)]
        asr = self.asr
        device = audio_feature.device
        dec, att = asr.decoder, asr.attention
        max_len = [int(np.ceil(l * self.max_len_ratio)) for l in lens_h]
        min_len = [int(np.ceil(l * self.min_len_ratio)) for l in lens_h]
        flen_dev = torch.as_tensor(feature_len).to(device)
        encode_feature, encode_len = asr.encoder(audio_feature, flen_dev, packed=True)       # [U,Te,Dv]
        Te = encode_feature.shape[1]
        enc_len_dev = encode_len.to(device=device, dtype=torch.int64)
        frames_dev = asr.encoder.packed_frames.to(device=device, dtype=torch.int64)   # tensor lengths of batch-1 runs
        enc_len_h = [int(v) for v in frames_dev.cpu().tolist()]     # the CTC prefix scorer walks the whole tensor
        att.reset_mem()
        s_key = ops.tanh(ops.linear(encode_feature, att.proj_k.weight, att.proj_k.bias))
        s_value = ops.tanh(ops.linear(encode_feature, att.proj_v.weight, att.proj_v.bias)) \
            if att.v_proj else encode_feature
        V = asr.vocab_size
        C = self.ctc_beam_size if self.apply_ctc else 0
        ctc_output, r0, mem_len32 = None, None, None
        if self.apply_ctc:
            ctc_output = ops.log_softmax(ops.linear(encode_feature, asr.ctc_layer.weight, asr.ctc_layer.bias))
            # CTCPrefixScore.init_state per utterance (src/ctc.py:27-35): r[t,1] = running f32 sum of the blank
            # log-probabilities over the utterance's own frames (host, sequential, as forward() does), r[t,0] = logzero
            blank_h = ctc_output[:, :, 0].cpu().numpy()
            r0_h = np.full((U, Te, 2), LOG_ZERO, dtype=np.float32)
            for u in range(U):
                r0_h[u, :enc_len_h[u], 1] = np.cumsum(blank_h[u, :enc_len_h[u]], dtype=np.float32)
            r0 = torch.from_numpy(r0_h).to(device)
            mem_len32 = frames_dev.to(torch.int32)
        if self.apply_lm:
            self.lm.to(device)
        lm_lstm = self.apply_lm and self.lm.rnn_type == 'LSTM'

        shared = dict(s_key=s_key, s_value=s_value, enc_len_dev=enc_len_dev, Te=Te, ctc_output=ctc_output, r0=r0,
                      mem_len32=mem_len32, lm_lstm=lm_lstm, C=C, max_len=max_len, min_len=min_len, device=device)
        # (Two interleaved halves on two host threads / streams - one half's kernels under the other half's host
        # bookkeeping - were measured and LOSE: 94.7 vs 163.9 utt/s at 32 utterances; the two Python threads fight over
        # the interpreter lock inside the launch path.  profiles/r04_decode_cfg5.json.)
        return self._search_rows(shared, list(range(U)), [None] * U)

    def _search_rows(self, sh, utts, result):
        ''' the decode loop of forward_batch over all utterances of the shared encoder memories `sh`; fills result[u]
            and returns `result`.

            Utterance u owns the device rows [u*B, (u+1)*B) (alive or not); at decode position t every live row is a
            hypothesis of t labels.  One position = one set of device launches over ALL rows and NO read-back: the beam
            bookkeeping of src/decode.py:150-167 / 209-239 (records in the reference's order, average scores in float64,
            stable ranking, <eos> and length handling) is asrk_beam_select_f32, which leaves the next position's gather
            indices, labels and CTC prefix probabilities on the device and appends to a back-pointer history and a log
            of finished hypotheses.  The host enqueues positions back to back (looking at the number of unfinished
            utterances every 8th position) and reads history + log once at the end; Hypothesis objects exist only for
            the results.  (Until round 6 every position ended in a read-back and a numpy pass: 2.6 ms per position at 32
            utterances, of which 1.4 were kernels.) '''
        asr = self.asr
        dec, att = asr.decoder, asr.attention
        device, Te, C = sh['device'], sh['Te'], sh['C']
        enc_len_dev, ctc_output, r0, mem_len32 = sh['enc_len_dev'], sh['ctc_output'], sh['r0'], sh['mem_len32']
        lm_lstm = sh['lm_lstm']
        B_ = self.beam_size
        U = len(utts)
        assert utts == list(range(U))
        R = U * B_
        lmax = max(1, max(sh['max_len']))
        fcap = B_ * (lmax + 2)
        L = _lib.load()
        i32 = dict(dtype=torch.int32, device=device)
        i64 = dict(dtype=torch.int64, device=device)
        row_mem = torch.arange(R, **i64) // B_                     # utterance of every row slot: fixed
        row_mem32 = row_mem.to(torch.int32)
        alive = torch.zeros(R, **i32)
        alive[::B_] = 1                                            # one empty hypothesis per utterance
        ssum = torch.zeros(R, dtype=torch.float64, device=device)
        utt_done = torch.zeros(U, **i32)
        prev_token, col = torch.zeros(R, **i64), torch.zeros(R, **i64)
        parent = torch.arange(R, **i64)
        pctc = torch.zeros(R, dtype=torch.float32, device=device)
        hist_tok, hist_par = torch.zeros((lmax, R), **i32), torch.zeros((lmax, R), **i32)
        hist_sc = torch.zeros((lmax, R), dtype=torch.float32, device=device)
        fin_count = torch.zeros(U, **i32)
        fin_kind, fin_t, fin_row = (torch.zeros((U, fcap), **i32) for _ in range(3))
        fin_term = torch.zeros((U, fcap), dtype=torch.float32, device=device)
        fin_ssum = torch.zeros((U, fcap), dtype=torch.float64, device=device)
        live = torch.full((1,), U, **i32)
        min_len_d = torch.tensor(sh['min_len'], dtype=torch.int32).to(device)
        max_len_d = torch.tensor(sh['max_len'], dtype=torch.int32).to(device)
        plen_all = torch.arange(lmax, **i32).view(lmax, 1).expand(lmax, R).contiguous()   # int32: what the scorer takes
        stepper = sops.MultiSpellerStepper(att, dec, sh['s_key'], sh['s_value'], enc_len_dev, R, row_group=B_)
        dops.drop_weight_panels()                 # weights may have been updated since the last search
        p_ = lambda t_: ctypes.c_void_p(t_.data_ptr()) if t_ is not None else ctypes.c_void_p(0)
        h_new = c_new = attn = lm_h = lm_c = r_new = None
        lm_hidden = None
        # Many rows: the language model's position (embedding, cells, vocabulary projection, log-softmax) depends on
        # nothing the acoustic side computes until joint_score, and its M = 512 GEMMs are 128 tiles each - half the chip.
        # It runs on a side stream beside the attention / decoder / prefix-score chain: it starts when the previous
        # position's bookkeeping has written prev_token / parent (ev_pos) and joint_score waits for it (ev_lm).  Every
        # tensor that crosses streams is consumed before the event that lets its producer stream run on is reached.
        lm_side = None
        if self.apply_lm and R >= dops.LSTM_CELL_GEMM_ROWS and os.environ.get("ASRK_DECODE_LM_STREAM", "1") != "0":
            lm_side = torch.cuda.Stream(device)
            ev_pos, ev_lm = torch.cuda.Event(), torch.cuda.Event()
            main_stream = torch.cuda.current_stream(device)

        f32 = dict(dtype=torch.float32, device=device)

        def lm_segments():
            """(gather segments, hidden) of the LM states entering a position: one segment per layer and state tensor"""
            hs = torch.empty_like(lm_h)
            segs = [(lm_h[l], hs[l], lm_h.shape[2], 1, 0) for l in range(lm_h.shape[0])]
            if not lm_lstm:
                return segs, hs
            cs = torch.empty_like(lm_c)
            return segs + [(lm_c[l], cs[l], lm_c.shape[2], 1, 0) for l in range(lm_c.shape[0])], (hs, cs)

        def lm_position(t, hid=None):
            if t > 0 and hid is None:
                segs, hid = lm_segments()
                if len(segs) <= 8 and os.environ.get("ASRK_DECODE_MULTI_GATHER", "1") != "0":
                    dops.gather_rows_multi(segs, parent)
                else:
                    hid = (lm_h.index_select(1, parent), lm_c.index_select(1, parent)) if lm_lstm \
                        else lm_h.index_select(1, parent)
            lm_out, lm_hid = self.lm(prev_token.unsqueeze(1), None, hidden=hid)
            return ops.log_softmax(lm_out[:, 0, :]), lm_hid

        for t in range(lmax):
            plen_d = plen_all[t]
            lm_logp = None
            if lm_side is not None:
                ev_pos.record(main_stream)
                with torch.cuda.stream(lm_side):
                    lm_side.wait_event(ev_pos)
                    lm_logp, lm_hid = lm_position(t)
                    lm_h, lm_c = lm_hid if lm_lstm else (lm_hid, lm_hid)
                    ev_lm.record(lm_side)
            state, lm_hid_in = None, None
            if t == 0:
                h_in = ops.zeros((R, dec.dim), device)
                c_in = ops.zeros((R, dec.dim), device)
                prev_att = sops.uniform_attention(enc_len_dev.index_select(0, row_mem), Te).unsqueeze(1)
                r_prev = r0.index_select(0, row_mem) if self.apply_ctc else None
            elif os.environ.get("ASRK_DECODE_MULTI_GATHER", "1") == "0":     # A/B: one ATen gather per state tensor
                h_in, c_in = h_new.index_select(0, parent), c_new.index_select(0, parent)
                prev_att = attn.index_select(0, parent)
                if self.apply_ctc:
                    r_prev = r_new[parent, col]
            else:
                # the survivors' states (rows `parent` of position t - 1; the prefix state also by candidate column) in ONE
                # launch: decoder h / c straight into the step's state slots, previous alignment, CTC prefix state and -
                # when the language model runs in line - its layers' states
                h_in = c_in = None
                state = (torch.empty((2, R, dec.dim), **f32), torch.empty((2, R, dec.dim), **f32))
                prev_att = torch.empty((R, 1, Te), **f32)
                segs = [(h_new, state[0][0], dec.dim, 1, 0), (c_new, state[1][0], dec.dim, 1, 0), (attn, prev_att, Te, 1, 0)]
                if self.apply_ctc:
                    r_prev = torch.empty((R, Te, 2), **f32)
                    segs.append((r_new, r_prev, 2 * Te, C, 1))
                if self.apply_lm and lm_side is None:
                    lsegs, hid = lm_segments()
                    if len(segs) + len(lsegs) <= 8:
                        segs += lsegs
                        lm_hid_in = hid
                dops.gather_rows_multi(segs, parent, col)
            attn, context, x, c_top = stepper.step(row_mem32, dops.embedding(prev_token, asr.pre_embed.weight),
                                                   prev_att, h_in, c_in, state=state)
            h_new, c_new = x, c_top
            att_logp = ops.log_softmax(dops.linear_infer(x, dec.char_trans.weight, dec.char_trans.bias))
            cand, psi, r_new = None, None, None
            if self.apply_ctc:
                _, cand = ops.topk(att_logp, C)
                psi, r_new = dops.ctc_prefix_scores(ctc_output, r_prev, plen_d, prev_token, cand, 0, 1, LOG_ZERO,
                                                    row_mem=row_mem32, mem_len=mem_len32)
            if self.apply_lm and lm_side is None:
                lm_logp, lm_hid = lm_position(t, lm_hid_in)
                lm_h, lm_c = lm_hid if lm_lstm else (lm_hid, lm_hid)
            elif lm_side is not None:
                main_stream.wait_event(ev_lm)
            if self.apply_ctc or self.apply_lm:
                cur_prob = dops.joint_score(att_logp, cand, psi, pctc if self.apply_ctc else None, lm_logp,
                                            self.ctc_w if self.apply_ctc else 0.0,
                                            self.lm_w if self.apply_lm else 0.0, LOG_ZERO)
            else:
                cur_prob = att_logp
            topv, topi = ops.topk(cur_prob, B_)
            # ---- bookkeeping for ALL utterances on the device; overwrites prev_token / parent / col / pctc / ssum /
            # alive for position t + 1 (stream-ordered after this position's kernels, which read them)
            _lib.check(L.asrk_beam_select_f32(
                p_(topv), p_(topi), p_(psi), p_(cand), U, B_, C, t, lmax, fcap, p_(min_len_d), p_(max_len_d),
                p_(alive), p_(ssum), p_(utt_done), p_(prev_token), p_(parent), p_(col), p_(pctc), p_(hist_tok),
                p_(hist_sc), p_(hist_par), p_(fin_count), p_(fin_kind), p_(fin_t), p_(fin_row), p_(fin_term),
                p_(fin_ssum), p_(live), ops._stream()), "beam_select")
            if (t & 7) == 7 and t + 1 < lmax and int(live.item()) == 0:
                break
        # ---- one read-back: the log of finished hypotheses and the back-pointer history they point into
        n_fin = fin_count.cpu().numpy()
        kind_h, t_h, row_h = fin_kind.cpu().numpy(), fin_t.cpu().numpy(), fin_row.cpu().numpy()
        term_h, fsum_h = fin_term.cpu().numpy(), fin_ssum.cpu().numpy()
        tok_h, par_h, sc_h = hist_tok.cpu().numpy(), hist_par.cpu().numpy(), hist_sc.cpu().numpy().astype(np.float64)

        # back-pointer walk of ALL logged hypotheses at once (one numpy gather per position instead of a Python loop per
        # hypothesis and position: 10 ms of a 70-ms batch decode)
        ent = [(u, j) for u in utts for j in range(int(n_fin[u]))]
        if ent:
            eu = np.asarray([e[0] for e in ent])
            ej = np.asarray([e[1] for e in ent])
            kind = kind_h[eu, ej]
            t_last = t_h[eu, ej] - (kind == 0)             # kind 0: row of position t (t labels) followed by <eos>
            slot = row_h[eu, ej].astype(np.int64)
            T_ = int(t_last.max()) + 1 if len(t_last) else 0
            toks = np.zeros((len(ent), max(T_, 1)), dtype=np.int64)
            scs = np.zeros((len(ent), max(T_, 1)), dtype=np.float64)
            for tt in range(T_ - 1, -1, -1):
                on = t_last >= tt
                sl = slot[on]
                toks[on, tt] = tok_h[tt, sl]
                scs[on, tt] = sc_h[tt, sl]
                slot[on] = par_h[tt, sl]
        fins = {u: [] for u in utts}
        for i, (u, j) in enumerate(ent):
            n_lab = int(t_last[i]) + 1
            seq, sc = toks[i, :n_lab].tolist(), scs[i, :n_lab].tolist()
            if kind[i] == 0:
                seq.append(1)
                sc.append(float(term_h[u, j]))
            fins[u].append(Hypothesis(None, output_seq=seq, output_scores=sc, lm_state=None, ctc_state=None,
                                      ctc_prob=None, att_map=None, score_sum=float(fsum_h[u, j])))
        for u in utts:
            fin = fins[u]
            if B_ > 1 or len(fin) > 1:
                fin.sort(key=lambda o: o.score_sum / len(o.output_scores), reverse=True)
            result[u] = fin[:B_]
        return result

    def _expand_beam(self, prev_top, packed, t, min_output_len, final_hypothesis, C):
        ''' Beam bookkeeping of ONE utterance for one decode position (src/decode.py:150-167): `packed[i]` is row i of
            the step's read-back (top-k values, top-k labels[, candidate prefix scores, candidate labels]) for
            hypothesis prev_top[i].  Returns (next_top, done): the surviving continuations, and whether the search
            ended (beam 1 stops at its first finished hypothesis, src/decode.py:160-161).
            Hypothesis.addTopk for every live hypothesis (src/decode.py:209-239), without materialising the beam^2
            continuations: only (average score, parent, label, ...) records are sorted (stable, same order as the
            reference's list) and the beam_size survivors become objects.  A hypothesis' average is
            sum(scores) / len in the reference; the running sum adds the same floats in the same order, so the
            values (and every tie) are identical. '''
        B_ = self.beam_size
        records, ended = [], []
        for i, hyp in enumerate(prev_top):
            row = packed[i]
            ssum, slen = hyp.score_sum, len(hyp.output_scores)
            if self.apply_ctc:
                psi_i, cand_i = row[2 * B_:2 * B_ + C], [int(v) for v in row[2 * B_ + C:]]
            term = None
            for k in range(B_):
                tok, sc = int(row[B_ + k]), row[k]
                if tok == 1:
                    term = sc
                    continue
                col, ctc_p = 0, None
                if self.apply_ctc:
                    if tok not in cand_i:
                        # only reachable when every CTC candidate is infeasible (fewer encoder frames
                        # than labels): the reference dies here with a ValueError from list.index
                        # (src/decode.py:225); drop the continuation instead
                        continue
                    col = cand_i.index(tok)
                    ctc_p = psi_i[col]
                records.append(((ssum + sc) / (slen + 1), i, tok, sc, col, ctc_p))
            if term is not None:
                ended.append((hyp, term))
        records.sort(key=lambda r: r[0], reverse=True)
        next_top = []
        for _, i, tok, sc, col, ctc_p in records[:self.beam_size]:
            par = prev_top[i]
            next_top.append(Hypothesis(None, output_seq=par.output_seq + [tok],
                                       output_scores=par.output_scores + [sc], lm_state=None,
                                       ctc_state=None, ctc_prob=ctc_p, att_map=None, parent=i, cand=col,
                                       score_sum=par.score_sum + sc))
        for hyp, term in ended:        # <eos> finalises the parent itself (src/decode.py:236-239)
            hyp.output_seq.append(1)
            hyp.output_scores.append(term)
            hyp.score_sum += term
            if t >= min_output_len:
                final_hypothesis.append(hyp)
                if self.beam_size == 1:
                    return next_top, True
        return next_top, False


class Hypothesis:
    ''' Hypothesis for beam search decoding (reference: src/decode.py:176-257): history of labels
        and scores.  The decoder / LM / CTC / attention state needed to extend it lives in the batched
        device tensors of BeamDecoder.forward; a hypothesis only remembers which row (`parent`) and
        which CTC candidate column (`cand`) of the previous step it came from.  The reference's
        constructor / addTopk signatures are kept (state arguments may be given and are stored). '''
    __slots__ = ('decoder_state', 'att_map', 'lm_state', 'output_seq', 'output_scores', 'ctc_state',
                 'ctc_prob', 'parent', 'cand', 'score_sum')

    def __init__(self, decoder_state, output_seq, output_scores, lm_state, ctc_state, ctc_prob, att_map,
                 parent=0, cand=0, score_sum=None):
        assert len(output_seq) == len(output_scores)
        self.decoder_state = decoder_state
        self.att_map = att_map
        self.lm_state = lm_state
        self.output_seq = output_seq
        self.output_scores = output_scores
        self.ctc_state = ctc_state
        self.ctc_prob = ctc_prob
        self.parent = parent
        self.cand = cand
        # sum(output_scores), accumulated left to right exactly like the built-in sum
        self.score_sum = sum(output_scores) if score_sum is None else score_sum

    @property
    def last_token(self):
        return self.output_seq[-1] if len(self.output_seq) != 0 else 0

    def avgScore(self):
        ''' averaged log probability of the hypothesis '''
        assert len(self.output_scores) != 0
        return sum(self.output_scores) / len(self.output_scores)

    def addTopk(self, topi, topv, decoder_state, att_map=None, lm_state=None, ctc_state=None,
                ctc_prob=0.0, ctc_candidates=[], parent=0):
        ''' Expand the hypothesis with its top-k continuations; <eos>=1 finalises it
            (src/decode.py:209-239).  `ctc_state` truthy = CTC scoring is on: the continuation keeps
            the candidate column of its label and that column's prefix probability. '''
        new_hypothesis = []
        term_score = None
        for i in range(len(topi)):
            if topi[i] == 1:
                term_score = topv[i]
                continue
            idxes = self.output_seq[:] + [topi[i]]
            scores = self.output_scores[:] + [topv[i]]
            cand, ctc_p = 0, None
            if ctc_state is not None and ctc_state is not False:
                if topi[i] not in ctc_candidates:
                    # only reachable when every CTC candidate is infeasible (fewer encoder frames than
                    # labels): an un-scored label outranks them.  The reference dies here with a
                    # ValueError from list.index (src/decode.py:225); drop the continuation instead.
                    continue
                cand = ctc_candidates.index(topi[i])
                ctc_p = ctc_prob[cand]
            new_hypothesis.append(Hypothesis(decoder_state, output_seq=idxes, output_scores=scores,
                                             lm_state=lm_state, ctc_state=None, ctc_prob=ctc_p,
                                             att_map=att_map, parent=parent, cand=cand))
        if term_score is not None:
            self.output_seq.append(1)
            self.output_scores.append(term_score)
            self.score_sum += term_score
            return self, new_hypothesis
        return None, new_hypothesis

    @property
    def outIndex(self):
        return [int(i) for i in self.output_seq]
