"""ASR model (Listener / Attention / Speller + CTC head) — MI355X mirror of the reference's
src/asr.py: same class names, constructor kwargs, attributes, state_dict keys and return tuple.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions.categorical import Categorical

from .. import ops
from .. import decoder_ops as dops
from .. import speller_ops as sops
from .util import init_weights, init_gate
from .module import (VGGExtractor, CNNExtractor, RNNLayer, RNNParams, ScaleDotAttention,
                     LocationAwareAttention)


class ASR(nn.Module):
    ''' ASR model, including Encoder/Decoder(s) (reference: src/asr.py:12-155) '''

    def __init__(self, input_size, vocab_size, init_adadelta, ctc_weight, encoder, attention,
                 decoder, emb_drop=0.0):
        super(ASR, self).__init__()

        assert 0 <= ctc_weight <= 1
        self.vocab_size = vocab_size
        self.ctc_weight = ctc_weight
        self.enable_ctc = ctc_weight > 0
        self.enable_att = ctc_weight != 1
        self.lm = None

        self.encoder = Encoder(input_size, **encoder)
        if self.enable_ctc:
            self.ctc_layer = nn.Linear(self.encoder.out_dim, vocab_size)
        if self.enable_att:
            self.dec_dim = decoder['dim']
            self.pre_embed = nn.Embedding(vocab_size, self.dec_dim)
            self.embed_drop = nn.Dropout(emb_drop)
            self.decoder = Decoder(self.encoder.out_dim + self.dec_dim, vocab_size, **decoder)
            query_dim = self.dec_dim * self.decoder.layer
            self.attention = Attention(self.encoder.out_dim, query_dim, **attention)

        if init_adadelta:
            self.apply(init_weights)
            if self.enable_att:
                for l in range(self.decoder.layer):
                    bias = getattr(self.decoder.layers, 'bias_ih_l{}'.format(l))
                    bias = init_gate(bias)

    def set_state(self, prev_state, prev_attn):
        ''' Setting up all memory states for beam decoding'''
        self.decoder.set_state(prev_state)
        self.attention.set_mem(prev_attn)

    def create_msg(self):
        msg = []
        msg.append('Model spec.| Encoder\'s downsampling rate of time axis is {}.'.format(
            self.encoder.sample_rate))
        if self.encoder.vgg:
            msg.append('           | VGG Extractor w/ time downsampling rate = 4 in encoder enabled.')
        if self.encoder.cnn:
            msg.append('           | CNN Extractor w/ time downsampling rate = 4 in encoder enabled.')
        if self.enable_ctc:
            msg.append('           | CTC training on encoder enabled ( lambda = {}).'.format(
                self.ctc_weight))
        if self.enable_att:
            msg.append('           | {} attention decoder enabled ( lambda = {}).'.format(
                self.attention.mode, 1 - self.ctc_weight))
        return msg

    def forward(self, audio_feature, feature_len, decode_step, tf_rate=0.0, teacher=None,
                emb_decoder=None, get_dec_state=False):
        ''' Same contract as the reference (src/asr.py:72-155): returns
            (ctc_output [B,T',V] log-probs | None, encode_len [B], att_output [B,L,V] logits | None,
             att_seq [B,N,L,T'] | None, dec_state [B,L,D] | None) '''
        bs = audio_feature.shape[0]
        ctc_output, att_output, att_seq = None, None, None
        dec_state = [] if get_dec_state else None

        encode_feature, encode_len = self.encoder(audio_feature, feature_len)

        if self.enable_ctc:
            # (The head depends on the encoder only, like the decoder loop below, so it was tried on a second stream
            # beside the loop - round 3 at equal priority, round 4 with the loop on a high-priority stream: the loop's
            # kernels lose what the head gains, 109.0 vs 109.1 ms/step; profiles/r04_head_beside_loop_priorities.log.)
            ctc_output = ops.log_softmax(
                ops.linear(encode_feature, self.ctc_layer.weight, self.ctc_layer.bias))

        if self.enable_att:
            # Init (init char = <SOS>, reset all rnn state and cell) - src/asr.py:101-105
            W = self.pre_embed.weight
            last_char = dops.embedding(
                torch.zeros((bs), dtype=torch.long, device=encode_feature.device), W)
            teacher_ids = teacher
            if (teacher is not None) and (0 < tf_rate < 1) and (emb_decoder is None) \
                    and teacher.shape[1] >= decode_step and sops.supported_loop(self.attention, self.decoder) \
                    and not (self.training and (self.decoder.dropout > 0 or self.embed_drop.p > 0)):
                # scheduled sampling through the fused loop, in two passes (see _scheduled_sampling_inputs)
                self.attention.reset_mem()
                key, value = self._speller_memory(encode_feature, encode_len)
                mixed_ids = self._scheduled_sampling_inputs(key.detach(), value.detach(), encode_len, last_char.detach(),
                                                            teacher_ids, decode_step, tf_rate)
                att_output, att_seq, states = self._teacher_forced_loop(
                    encode_feature, encode_len, last_char, self._embed_drop(dops.embedding(mixed_ids, W)),
                    decode_step, memory=(key, value))
                return ctc_output, encode_len, att_output, att_seq, (states if get_dec_state else None)
            if teacher is not None:
                teacher = self._embed_drop(dops.embedding(teacher, W))
            if (teacher is not None) and (tf_rate == 1) and (emb_decoder is None) \
                    and teacher.shape[1] >= decode_step - 1 \
                    and sops.supported_train_loop(self.attention, self.decoder, self.training):
                # the whole teacher-forced loop as one autograd node (csrc/speller.hip)
                self.attention.reset_mem()
                att_output, att_seq, states = self._teacher_forced_loop(
                    encode_feature, encode_len, last_char, teacher, decode_step)
                return ctc_output, encode_len, att_output, att_seq, (states if get_dec_state else None)
            if (teacher is None) and (emb_decoder is None) and (not torch.is_grad_enabled()) \
                    and sops.supported_loop(self.attention, self.decoder):
                # greedy inference (validation / greedy test decoding, src/asr.py:136-142): one fused C
                # call per step for attention + decoder cell
                self.attention.reset_mem()
                att_output, att_seq, states = self._greedy_loop(encode_feature, encode_len, last_char,
                                                                decode_step)
                return ctc_output, encode_len, att_output, att_seq, (states if get_dec_state else None)
            self.decoder.init_state(bs, max_steps=decode_step)
            self.attention.reset_mem()
            att_seq, output_seq, state_seq = [], [], []

            # full teacher forcing: the vocabulary projection of all L steps is ONE GEMM after the
            # loop instead of L small ones (same math; src/asr.py:220 applies it per step)
            defer_char = (teacher is not None) and (tf_rate == 1) and (emb_decoder is None)

            for t in range(decode_step):
                attn, context = self.attention(self.decoder.get_query(), encode_feature, encode_len)
                decoder_input = dops.concat_last(last_char, context)
                cur_char, d_state = self.decoder(decoder_input, project=not defer_char)
                if teacher is not None:
                    if (tf_rate == 1) or (torch.rand(1).item() <= tf_rate):
                        last_char = teacher[:, t, :]
                    else:
                        with torch.no_grad():
                            if (emb_decoder is not None) and emb_decoder.apply_fuse:
                                _, cur_prob = emb_decoder(d_state, cur_char, return_loss=False)
                            else:
                                cur_prob = ops.log_softmax(cur_char).exp()
                            sampled_char = Categorical(cur_prob).sample()
                        last_char = self._embed_drop(dops.embedding(sampled_char, W))
                else:
                    if (emb_decoder is not None) and emb_decoder.apply_fuse:
                        _, cur_char = emb_decoder(d_state, cur_char, return_loss=False)
                    last_char = dops.embedding(ops.argmax(cur_char), W)

                output_seq.append(cur_char)
                state_seq.append(d_state)
                att_seq.append(attn)

            if defer_char:
                states = dops.stack_steps(state_seq)                                  # [B,L,D]
                # Decoder.forward applies final_dropout to the state that feeds char_trans
                # (src/asr.py:220); one mask over [B,L,D] = L independent per-step masks
                states = ops.dropout(states, self.decoder.dropout, self.training)
                att_output = ops.linear(states, self.decoder.char_trans.weight,
                                        self.decoder.char_trans.bias)                 # [B,L,V]
            else:
                att_output = dops.stack_steps(output_seq)                             # [B,L,V]
            N, T = att_seq[0].shape[1], att_seq[0].shape[2]
            att_seq = dops.stack_steps([a.reshape(bs * N, T) for a in att_seq]) \
                .view(bs, N, decode_step, T)                                          # [B,N,L,T]
            if get_dec_state:
                dec_state = dops.stack_steps(state_seq)

        return ctc_output, encode_len, att_output, att_seq, dec_state

    def _speller_memory(self, encode_feature, encode_len):
        ''' key / value projections of the encoder memory (src/asr.py:277-288) + the attention's length mask '''
        att = self.attention
        att.att_layer.compute_mask(encode_feature, encode_len.to(encode_feature.device))
        key = ops.tanh(ops.linear(encode_feature, att.proj_k.weight, att.proj_k.bias))
        value = ops.tanh(ops.linear(encode_feature, att.proj_v.weight, att.proj_v.bias)) \
            if att.v_proj else encode_feature
        return key, value

    def _scheduled_sampling_inputs(self, key, value, encode_len, sos_emb, teacher_ids, decode_step, tf_rate):
        ''' 0 < tf_rate < 1 (src/asr.py:119-135): the token fed to step t + 1 is the teacher's with probability
            tf_rate, else one SAMPLED from the model's own softmax at step t - no gradient flows through the draw, so
            the training step is the teacher-forced step on the MIXED token sequence.  Pass 1 (here, no autograd): the
            loop one fused step at a time (asrk_speller_step_f32), taking the reference's decisions and draws in the
            reference's order (one torch.rand(1) per step, Categorical.sample on the steps that sample); pass 2 (the
            caller): the fused teacher-forced loop on the mixed sequence, forward and backward.  -> ids [B, L] '''
        att, dec = self.attention, self.decoder
        bs, ts, _ = key.shape
        W = self.pre_embed.weight.detach()
        mixed = teacher_ids[:, :decode_step].clone()
        with torch.no_grad():
            st = sops.SpellerStepper(att, dec, key, value, encode_len.to(key.device), bs, shared=False)
            st.h[0].zero_()
            st.c[0].zero_()
            prev_att = att.att_layer.uniform_init(bs, ts, key.device)
            last_char = sos_emb
            for t in range(decode_step):
                attn, _, x, c = st.step(last_char, prev_att)
                if torch.rand(1).item() > tf_rate:
                    cur_char = ops.linear(x, dec.char_trans.weight, dec.char_trans.bias)
                    mixed[:, t] = Categorical(ops.log_softmax(cur_char).exp()).sample()
                last_char = dops.embedding(mixed[:, t].contiguous(), W)
                prev_att = attn.clone()
                st.h[0].copy_(x)
                st.c[0].copy_(c)
        return mixed

    def _teacher_forced_loop(self, encode_feature, encode_len, sos_emb, teacher_emb, decode_step, memory=None):
        ''' src/asr.py:112-148 with tf_rate == 1 -> (att_output [B,L,V], att_seq [B,1,L,T], states [B,L,D]) '''
        att, dec = self.attention, self.decoder
        enc_len = encode_len.to(encode_feature.device)
        key, value = memory if memory is not None else self._speller_memory(encode_feature, encode_len)
        al = att.att_layer
        N = att.num_head
        if N > 1:                               # the heads' rows b * N + n, built as src/asr.py:294-304 builds them
            key = HeadSplitFn.apply(key, N)
            value = HeadSplitFn.apply(value, N) if att.v_proj else RepeatBatchFn.apply(value, N)
        w_ih, w_hh, b_ih, b_hh = dec.layers.layer_params(0)
        if not dec.enable_cell:                 # GRU decoder: the loop's four-rows-per-unit layout (speller_ops)
            w_ih, w_hh, b_ih, b_hh = sops.stack_gru_params(w_ih, w_hh, b_ih, b_hh)
        upper = [p_ for l in range(1, dec.layer) for p_ in dec.layers.layer_params(l)]     # stacked LSTM decoder
        loc_w = (al.loc_conv.weight, al.loc_proj.weight, al.gen_energy.weight, al.gen_energy.bias) \
            if att.mode == 'loc' else (None, None, None, None)                             # None: dot-product energies
        merge = (att.merge_head.weight, att.merge_head.bias) if N > 1 else (None, None)
        states, att_seq = sops.SpellerLoopFn.apply(
            key, value, enc_len, sos_emb, teacher_emb, att.proj_q.weight, att.proj_q.bias, *loc_w,
            w_ih, w_hh, b_ih, b_hh, decode_step, al.temperature, 0 if dec.enable_cell else 1, N, *merge, *upper)
        # module state as the step-by-step loop leaves it (decode-time callers read these)
        att.key, att.value = key, value
        al.prev_att = att_seq.detach()[:, :, -1, :]
        x = ops.dropout(states, dec.dropout, self.training)       # Decoder.final_dropout (src/asr.py:220)
        att_output = ops.linear(x, dec.char_trans.weight, dec.char_trans.bias)
        return att_output, att_seq, states

    def _greedy_loop(self, encode_feature, encode_len, sos_emb, decode_step):
        ''' argmax-feedback decoding without autograd -> (att_output [B,L,V], att_seq [B,1,L,T], states) '''
        att, dec = self.attention, self.decoder
        bs, ts, _ = encode_feature.shape
        dev = encode_feature.device
        enc_len = encode_len.to(dev)
        att.att_layer.compute_mask(encode_feature, enc_len)
        key = ops.tanh(ops.linear(encode_feature, att.proj_k.weight, att.proj_k.bias))
        value = ops.tanh(ops.linear(encode_feature, att.proj_v.weight, att.proj_v.bias)) \
            if att.v_proj else encode_feature
        st = sops.SpellerStepper(att, dec, key, value, enc_len, bs, shared=False)
        st.h[0].zero_()
        st.c[0].zero_()
        prev_att = att.att_layer.uniform_init(bs, ts, dev)
        W = self.pre_embed.weight
        f = dict(dtype=torch.float32, device=dev)
        att_output = torch.empty((bs, decode_step, self.vocab_size), **f)
        att_seq = torch.empty((bs, 1, decode_step, ts), **f)
        states = torch.empty((bs, decode_step, dec.dim), **f)
        last_char = sos_emb
        for t in range(decode_step):
            attn, _, x, c = st.step(last_char, prev_att)
            cur_char = ops.linear(ops.dropout(x, dec.dropout, self.training), dec.char_trans.weight,
                                  dec.char_trans.bias)
            last_char = dops.embedding(ops.argmax(cur_char), W)
            att_output[:, t].copy_(cur_char)
            att_seq[:, :, t].copy_(attn)
            states[:, t].copy_(x)
            prev_att = att_seq[:, :, t].contiguous()
            st.h[0].copy_(x)
            st.c[0].copy_(c)
        att.key, att.value = key, value
        att.att_layer.prev_att = prev_att
        dec.hidden_state = (st.h[0].clone().unsqueeze(0), st.c[0].clone().unsqueeze(0)) if dec.enable_cell \
            else st.h[0].clone().unsqueeze(0)
        return att_output, att_seq, states

    def _embed_drop(self, x):
        return ops.dropout(x, self.embed_drop.p, self.training)


class Decoder(nn.Module):
    ''' Decoder (a.k.a. Speller in LAS) (reference: src/asr.py:158-221) '''

    def __init__(self, input_dim, vocab_size, module, dim, layer, dropout):
        super(Decoder, self).__init__()
        self.in_dim = input_dim
        self.layer = layer
        self.dim = dim
        self.dropout = dropout

        assert module in ['LSTM', 'GRU'], NotImplementedError
        self.hidden_state = None
        self.enable_cell = module == 'LSTM'

        # parameters named exactly like nn.LSTM(input_dim, dim, num_layers=layer) (decode.py reads
        # decoder.layers.bias_ih_l{l}); the math is dops.LSTMCellStepFn
        self.layers = RNNParams(module, input_dim, dim, num_layers=layer, dropout=dropout,
                                batch_first=True)
        self.char_trans = nn.Linear(dim, vocab_size)
        self.final_dropout = nn.Dropout(dropout)
        self._tapes = None

    def init_state(self, bs, max_steps=None):
        ''' Set all hidden states to zeros '''
        device = next(self.parameters()).device
        z = lambda: torch.zeros((self.layer, bs, self.dim), device=device)
        self.hidden_state = (z(), z()) if self.enable_cell else z()
        self._tapes = None
        if self.enable_cell and max_steps is not None and torch.is_grad_enabled():
            self._tapes = []
            for l in range(self.layer):
                w = self.layers.layer_params(l)
                tape = dops.CellTape(*[ops._f32c(p) for p in w], bs, max_steps)
                token = dops.CellHubFn.apply(tape, *w)
                self._tapes.append((tape, token))
        return self.get_state()

    def set_state(self, hidden_state):
        ''' Set all hidden states/cells, for decoding purpose'''
        device = next(self.parameters()).device
        if self.enable_cell:
            self.hidden_state = (hidden_state[0].to(device), hidden_state[1].to(device))
        else:
            self.hidden_state = hidden_state.to(device)

    def get_state(self):
        ''' Return all hidden states/cells, for decoding purpose'''
        if self.enable_cell:
            return (self.hidden_state[0].cpu(), self.hidden_state[1].cpu())
        else:
            return self.hidden_state.cpu()

    def get_query(self):
        ''' Return state of all layers as query for attention '''
        h = self.hidden_state[0] if self.enable_cell else self.hidden_state
        if self.layer == 1:
            return h[0]
        return h.transpose(0, 1).reshape(-1, self.dim * self.layer)

    def forward(self, x, project=True):
        ''' One decode step through the stacked cells, then transform into vocab '''
        if not self.enable_cell:
            return self._forward_gru(x, project)
        hs, cs = [], []
        h_all, c_all = self.hidden_state
        bs = x.shape[0]
        for l in range(self.layer):
            if self._tapes is not None and self._tapes[l][0].used < self._tapes[l][0].cap:
                tape, token = self._tapes[l]
            else:  # stepping outside ASR.forward (beam search) or beyond the planned length
                w = self.layers.layer_params(l)
                tape = dops.CellTape(*[ops._f32c(p) for p in w], bs, 1)
                token = dops.CellHubFn.apply(tape, *w)
            h, c = dops.LSTMCellStepFn.apply(tape, token, x, h_all[l], c_all[l])
            hs.append(h)
            cs.append(c)
            x = h
            if l + 1 < self.layer:
                x = ops.dropout(x, self.dropout, self.training)   # nn.LSTM inter-layer dropout
        if self.layer == 1:
            self.hidden_state = (hs[0].unsqueeze(0), cs[0].unsqueeze(0))
        else:
            self.hidden_state = (torch.stack(hs, 0), torch.stack(cs, 0))
        char = None
        if project:
            char = ops.linear(ops.dropout(x, self.dropout, self.training), self.char_trans.weight,
                              self.char_trans.bias)
        return char, x

    def _forward_gru(self, x, project):
        ''' stacked nn.GRU cells, one step (state = h only) '''
        from .. import gru_ops
        hs = []
        for l in range(self.layer):
            h = gru_ops.gru_cell(x, self.hidden_state[l], *self.layers.layer_params(l))
            hs.append(h)
            x = h
            if l + 1 < self.layer:
                x = ops.dropout(x, self.dropout, self.training)
        self.hidden_state = hs[0].unsqueeze(0) if self.layer == 1 else torch.stack(hs, 0)
        char = None
        if project:
            char = ops.linear(ops.dropout(x, self.dropout, self.training), self.char_trans.weight,
                              self.char_trans.bias)
        return char, x


class Attention(nn.Module):
    ''' Attention mechanism (reference: src/asr.py:224-313).
        Input : decoder state [B, q_dim], encoder memory [B, T, v_dim], lengths [B]
        Output: attention score [B, num_head, T], context vector [B, v_dim] '''

    def __init__(self, v_dim, q_dim, mode, dim, num_head, temperature, v_proj,
                 loc_kernel_size, loc_kernel_num):
        super(Attention, self).__init__()
        self.v_dim = v_dim
        self.dim = dim
        self.mode = mode.lower()
        self.num_head = num_head

        self.proj_q = nn.Linear(q_dim, dim * num_head)
        self.proj_k = nn.Linear(v_dim, dim * num_head)
        self.v_proj = v_proj
        if v_proj:
            self.proj_v = nn.Linear(v_dim, v_dim * num_head)

        if self.mode == 'dot':
            self.att_layer = ScaleDotAttention(temperature, self.num_head)
        elif self.mode == 'loc':
            self.att_layer = LocationAwareAttention(
                loc_kernel_size, loc_kernel_num, dim, num_head, temperature)
        else:
            raise NotImplementedError

        if self.num_head > 1:
            self.merge_head = nn.Linear(v_dim * num_head, v_dim)

        self.key = None
        self.value = None
        self.mask = None
        self._tape = None
        self._token = None

    def reset_mem(self):
        self.key = None
        self.value = None
        self.mask = None
        self._tape = None
        self._token = None
        self.att_layer.reset_mem()

    def set_mem(self, prev_attn):
        self.att_layer.set_mem(prev_attn)

    def _split_heads(self, x, bs, ts, d):
        ''' [B,T,N*d] -> [B*N,T,d] (src/asr.py:294-301) '''
        N = self.num_head
        out = torch.empty((bs * N, ts, d), dtype=torch.float32, device=x.device)
        xc = ops._f32c(x)
        for b in range(bs):  # out[b*N+n, t, :] = x[b, t, n*d:(n+1)*d]
            ops.copy3d(xc[b], out[b * N:], N, ts, d, d, N * d, ts * d, d)
        return out

    def build_memory(self, enc_feat, enc_len):
        ''' key/value cache + padded-frame mask of src/asr.py:285-304 -> (AttnTape, key, value, loc_w) '''
        N = self.num_head
        enc_len = enc_len.to(enc_feat.device)
        self.att_layer.compute_mask(enc_feat, enc_len)
        key = ops.tanh(ops.linear(enc_feat, self.proj_k.weight, self.proj_k.bias))
        value = ops.tanh(ops.linear(enc_feat, self.proj_v.weight, self.proj_v.bias)) \
            if self.v_proj else enc_feat
        if N > 1:
            key = HeadSplitFn.apply(key, N)
            if self.v_proj:
                value = HeadSplitFn.apply(value, N)
            else:
                value = RepeatBatchFn.apply(value, N)   # reference: value.repeat(N,1,1)
        loc_w = ()
        if self.mode == 'loc':
            al = self.att_layer
            loc_w = (al.loc_conv.weight, al.loc_proj.weight, al.gen_energy.weight.view(-1),
                     al.gen_energy.bias)
        tape = dops.AttnTape(self.mode, ops._f32c(key), ops._f32c(value), enc_len, N,
                             self.att_layer.temperature,
                             tuple(ops._f32c(w) for w in loc_w) if loc_w else None)
        return tape, key, value, loc_w

    def forward(self, dec_state, enc_feat, enc_len):
        bs, ts, _ = enc_feat.shape
        N = self.num_head
        query = ops.tanh(ops.linear(dec_state, self.proj_q.weight, self.proj_q.bias))
        query = query.view(bs * N, self.dim)  # BNxD

        if self.key is None:
            self._tape, key, value, loc_w = self.build_memory(enc_feat, enc_len)
            self.key, self.value = key, value
            self._token = dops.AttnHubFn.apply(self._tape, key, value, *loc_w)

        prev_att = None
        if self.mode == 'loc':
            if self.att_layer.prev_att is None:
                self.att_layer.prev_att = self.att_layer.uniform_init(bs, ts, enc_feat.device)
            prev_att = self.att_layer.prev_att
        attn, context = dops.AttnStepFn.apply(self._tape, self._token, query, prev_att)
        if self.mode == 'loc':
            self.att_layer.prev_att = attn
        if N > 1:
            context = context.view(bs, N * self.v_dim)
            context = ops.linear(context, self.merge_head.weight, self.merge_head.bias)
        return attn, context


class HeadSplitFn(torch.autograd.Function):
    ''' [B,T,N*d] -> [B*N,T,d]: view(B,T,N,d).permute(0,2,1,3) of src/asr.py:294-301 '''

    @staticmethod
    def forward(ctx, x, N):
        xc = ops._f32c(x)
        B, T, ND = xc.shape
        d = ND // N
        out = torch.empty((B * N, T, d), dtype=torch.float32, device=x.device)
        for b in range(B):
            ops.copy3d(xc[b], out[b * N:], N, T, d, d, ND, T * d, d)
        ctx.dims = (B, T, N, d)
        return out

    @staticmethod
    def backward(ctx, g):
        B, T, N, d = ctx.dims
        gc = ops._f32c(g)
        out = torch.empty((B, T, N * d), dtype=torch.float32, device=g.device)
        for b in range(B):
            ops.copy3d(gc[b * N:], out[b], N, T, d, T * d, d, d, N * d)
        return out, None


class RepeatBatchFn(torch.autograd.Function):
    ''' x.repeat(N,1,1) for [B,T,D] (src/asr.py:304; rows ordered (n, b) exactly like the reference) '''

    @staticmethod
    def forward(ctx, x, N):
        xc = ops._f32c(x)
        B, T, D = xc.shape
        out = torch.empty((N * B, T, D), dtype=torch.float32, device=x.device)
        for n in range(N):
            ops.copy3d(xc, out[n * B:], 1, B * T, D, 0, D, 0, D)
        ctx.dims = (B, T, D, N)
        return out

    @staticmethod
    def backward(ctx, g):
        B, T, D, N = ctx.dims
        gc = ops._f32c(g)
        out = torch.zeros((B, T, D), dtype=torch.float32, device=g.device)
        for n in range(N):
            ops.copy3d(gc[n * B:], out, 1, B * T, D, 0, D, 0, D, accumulate=True)
        return out, None


class Encoder(nn.Module):
    ''' Encoder (a.k.a. Listener in LAS) (reference: src/asr.py:316-366).  Layers are chained in a
    time-major layout internally ([T,B,D], what the persistent recurrence kernels want); the
    module's own interface stays batch-major like the reference. '''

    def __init__(self, input_size, prenet, module, bidirection, dim, dropout, layer_norm, proj,
                 sample_rate, sample_style):
        super(Encoder, self).__init__()

        self.vgg = prenet == 'vgg'
        self.cnn = prenet == 'cnn'
        self.sample_rate = 1
        assert len(sample_rate) == len(dropout), 'Number of layer mismatch'
        assert len(dropout) == len(dim), 'Number of layer mismatch'
        num_layers = len(dim)
        assert num_layers >= 1, 'Encoder should have at least 1 layer'

        module_list = []
        input_dim = input_size

        if self.vgg:
            vgg_extractor = VGGExtractor(input_size)
            module_list.append(vgg_extractor)
            input_dim = vgg_extractor.out_dim
            self.sample_rate = self.sample_rate * 4
        if self.cnn:
            cnn_extractor = CNNExtractor(input_size, out_dim=dim[0])
            module_list.append(cnn_extractor)
            input_dim = cnn_extractor.out_dim
            self.sample_rate = self.sample_rate * 4

        if module in ['LSTM', 'GRU']:
            for l in range(num_layers):
                module_list.append(RNNLayer(input_dim, module, dim[l], bidirection, dropout[l],
                                            layer_norm[l], sample_rate[l], sample_style, proj[l]))
                input_dim = module_list[-1].out_dim
                self.sample_rate = self.sample_rate * sample_rate[l]
        else:
            raise NotImplementedError

        self.in_dim = input_size
        self.out_dim = input_dim
        self.layers = nn.ModuleList(module_list)

    def supports_packed(self):
        ''' forward(packed=True) is available: no prenet (its convolutions would see the neighbours' padding) and
            every recurrent layer is an LSTM with a hidden size the packed kernel takes '''
        return not (self.vgg or self.cnn) and all(l.supports_packed() for l in self.layers)

    def forward(self, input_x, enc_len, packed=False):
        ''' packed=True (inference): input_x [U,T,D] zero-padded, enc_len [U]; every utterance is encoded exactly as
            if it had been passed alone and unpadded (RNNLayer.forward_tm(packed=True)); output frames beyond an
            utterance's length are zero (LayerNorm / projection outputs there are NOT - consumers mask by length) '''
        if packed:
            if not self.supports_packed():
                raise RuntimeError('Encoder: packed encoding is not available for this configuration')
            x = ops.swap_bt(input_x)
            frames = enc_len
            for layer in self.layers:
                x, enc_len, frames = layer.forward_tm(x, enc_len, packed=True, frames=frames)
            # frames an utterance's own (batch-1) encoder output would have: what everything that walks the output
            # TENSOR of such a run sees (CTC prefix scorer, CTC beam search); == enc_len unless a 'drop' reduction
            # met an odd length
            self.packed_frames = frames
            # inference: nobody polls the recurrence kernels' error word every N steps as the training loop does, and an
            # aborted launch leaves its pooled exchange buffer dirty - ask right here, before anything decodes from x
            ops.check_errors()
            return ops.swap_bt(x), enc_len
        layers = list(self.layers)
        if self.vgg or self.cnn:
            # the prenet reads the batch-major features in place and emits time-major frames
            x, enc_len = layers[0].forward_bm2tm(input_x, enc_len)
            layers = layers[1:]
        else:
            x = ops.swap_bt(input_x)  # [B,T,D] -> [T,B,D] once
        for i, layer in enumerate(layers):
            # a wide LSTM layer whose output goes straight into the next LSTM layer's input projection (no LayerNorm /
            # dropout / projection in between) may write that output as a split panel itself (ops.set_panel_hint)
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            ops.set_panel_hint(isinstance(layer, RNNLayer) and isinstance(nxt, RNNLayer) and layer.rnn_type == 'LSTM'
                               and nxt.rnn_type == 'LSTM' and nxt.bidirection and nxt.layer.hidden_size >= 768
                               and not layer.layer_norm and not layer.proj
                               and not (layer.dropout > 0 and self.training)
                               and (layer.sample_rate == 1 or layer.sample_style == 'concat'))
            x, enc_len = layer.forward_tm(x, enc_len)
        ops.set_panel_hint(False)
        if not torch.is_grad_enabled():
            ops.check_errors()          # decoding / validation: see the packed branch
        return ops.swap_bt(x), enc_len
