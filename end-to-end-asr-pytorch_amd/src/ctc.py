"""CTC prefix scoring + pure-CTC beam search — MI355X mirror of the reference's src/ctc.py.

* CTCPrefixScore keeps the reference API (`init_state`, `cheap_compute`, `full_compute`, numpy in /
  numpy out) but the T'-frame recursion runs in the gfx950 kernel `asrk_ctc_prefix_score_f32`;
  `cheap_compute_batch` scores every (hypothesis, candidate) pair of a beam step in ONE launch.
* CTCBeamDecoder is Graves-2014 prefix beam search (reference: src/ctc.py:210-352): the encoder, CTC head,
  log-softmax, the optional RNN-LM AND the search itself (expansion, the reference's string-order
  de-duplication, pruning: csrc/prefix_beam.hip) run on the device; hypotheses are read back once at the end.
"""
import numpy as np
import torch
import yaml
from torch import nn

from .. import ops
from .. import decoder_ops as dops
from .lm import RNNLM

LOG_ZERO = -10000000.0  # Log-zero for CTC


class CTCPrefixScore():
    ''' CTC prefix score calculator (Watanabe et al. Algo. 2; reference: src/ctc.py:12-116) '''

    def __init__(self, x):
        self.logzero = -100000000.0
        self.blank = 0
        self.eos = 1
        self.xd = ops._f32c(x.detach()[0])          # [T', V] log-probs, stays on the device
        self.odim = x.shape[-1]
        self.input_length = self.xd.shape[0]
        self._x_host = None

    @property
    def x(self):
        if self._x_host is None:
            self._x_host = self.xd.cpu().numpy()
        return self._x_host

    def init_state(self):
        ''' r[t,1] = cumulative blank log-prob, r[t,0] = logzero (src/ctc.py:27-35) '''
        r = np.full((self.input_length, 2), self.logzero, dtype=np.float32)
        r[:, 1] = np.cumsum(self.x[:, self.blank], dtype=np.float32)
        return r

    def init_state_device(self):
        return torch.from_numpy(self.init_state()).to(self.xd.device)

    def cheap_compute_batch(self, prefix_len, last_char, r_prev, candidates):
        ''' r_prev [n,T',2] (device), candidates [n,C] -> psi [n,C], r [n,C,T',2] (device) '''
        return dops.ctc_prefix_scores(self.xd, r_prev, prefix_len, last_char, candidates, self.blank,
                                      self.eos, self.logzero)

    def cheap_compute(self, g, r_prev, candidates):
        ''' reference signature (src/ctc.py:76-116): prefix g (list), r_prev [T',2] numpy,
            candidates (list) -> (psi [C], r [C,T',2]) numpy '''
        rp = torch.from_numpy(np.ascontiguousarray(r_prev, dtype=np.float32)).to(self.xd.device)
        psi, r = self.cheap_compute_batch([len(g)], [g[-1] if len(g) > 0 else 0], rp.unsqueeze(0),
                                          torch.tensor([list(candidates)], dtype=torch.int32))
        return psi[0].cpu().numpy(), r[0].cpu().numpy()

    def full_compute(self, g, r_prev):
        ''' all tokens as candidates (src/ctc.py:37-74; the <eos> override is commented out there,
            so it is undone here) '''
        rp = torch.from_numpy(np.ascontiguousarray(r_prev, dtype=np.float32)).to(self.xd.device)
        cands = torch.arange(self.odim, dtype=torch.int32).unsqueeze(0)
        psi, r = dops.ctc_prefix_scores(self.xd, rp.unsqueeze(0), [len(g)], [g[-1] if len(g) > 0 else 0],
                                        cands, self.blank, -1, self.logzero)   # eos = -1: no override
        return psi[0].cpu().numpy(), r[0].cpu().numpy()


class _LMOut:
    """RNN-LM log-probabilities of one hypothesis: host copy for the scalar look-ups of the prefix
    bookkeeping (indexable like the reference's numpy vector) + the device tensor for batched ranking"""
    __slots__ = ('host', 'dev')

    def __init__(self, host, dev):
        self.host, self.dev = host, dev

    def __getitem__(self, k):
        return self.host[k]


class CTCHypothesis():
    ''' Hypothesis for pure CTC beam search decoding (Graves 2014 Algo. 1; reference:
        src/ctc.py:118-208).  Plain-float state; `clone()` replaces the reference's deepcopy. '''
    __slots__ = ('y', 'Pr_y_t_blank', 'Pr_y_t_nblank', 'Pr_y_t_blank_bkup', 'Pr_y_t_nblank_bkup',
                 'lm_output', 'lm_hidden', 'updated_lm')

    def __init__(self):
        self.y = []
        self.Pr_y_t_blank = 0.0
        self.Pr_y_t_nblank = LOG_ZERO
        self.Pr_y_t_blank_bkup = 0.0
        self.Pr_y_t_nblank_bkup = LOG_ZERO
        self.lm_output = None
        self.lm_hidden = None
        self.updated_lm = False

    def clone(self):
        h = CTCHypothesis()
        h.y = self.y[:]
        h.Pr_y_t_blank, h.Pr_y_t_nblank = self.Pr_y_t_blank, self.Pr_y_t_nblank
        h.Pr_y_t_blank_bkup, h.Pr_y_t_nblank_bkup = self.Pr_y_t_blank_bkup, self.Pr_y_t_nblank_bkup
        h.lm_output, h.lm_hidden, h.updated_lm = self.lm_output, self.lm_hidden, self.updated_lm
        return h

    def update_lm(self, output, hidden):
        self.lm_output = output
        self.lm_hidden = hidden
        self.updated_lm = True

    def get_len(self):
        return len(self.y)

    def get_string(self):
        return ''.join([str(s) for s in self.y])

    def get_score(self):
        return np.logaddexp(self.Pr_y_t_blank, self.Pr_y_t_nblank)

    def get_final_score(self):
        s = np.logaddexp(self.Pr_y_t_blank, self.Pr_y_t_nblank)
        return s / len(self.y) if len(self.y) > 0 else s

    def check_same(self, y_2):
        return self.y == list(y_2)

    def update_Pr_nblank(self, ctc_y_t):
        self.Pr_y_t_nblank += ctc_y_t

    def update_Pr_nblank_prefix(self, ctc_y_t, Pr_y_t_blank_prefix, Pr_y_t_nblank_prefix, Pr_ye_y=None):
        lm_prob = Pr_ye_y if Pr_ye_y is not None else 0.0
        if len(self.y) == 0:
            return
        if len(self.y) == 1 or self.y[-1] != self.y[-2]:
            base = np.logaddexp(Pr_y_t_blank_prefix, Pr_y_t_nblank_prefix)
        else:
            base = Pr_y_t_blank_prefix
        self.Pr_y_t_nblank = np.logaddexp(self.Pr_y_t_nblank, ctc_y_t + lm_prob + base)

    def update_Pr_blank(self, ctc_blank_t):
        self.Pr_y_t_blank = np.logaddexp(self.Pr_y_t_nblank_bkup, self.Pr_y_t_blank_bkup) + ctc_blank_t

    def add_token(self, token, ctc_token_t, Pr_k_y=None):
        lm_prob = Pr_k_y if Pr_k_y is not None else 0.0
        if len(self.y) == 0 or self.y[-1] != token:
            base = np.logaddexp(self.Pr_y_t_blank_bkup, self.Pr_y_t_nblank_bkup)
        else:
            base = self.Pr_y_t_blank_bkup
        self.Pr_y_t_blank = LOG_ZERO
        self.Pr_y_t_nblank = ctc_token_t + lm_prob + base
        self.Pr_y_t_blank_bkup = self.Pr_y_t_blank
        self.Pr_y_t_nblank_bkup = self.Pr_y_t_nblank
        self.y.append(token)

    def orig_backup(self):
        self.Pr_y_t_blank_bkup = self.Pr_y_t_blank
        self.Pr_y_t_nblank_bkup = self.Pr_y_t_nblank


class CTCBeamDecoder(nn.Module):
    ''' Beam decoder for ASR (CTC only) (reference: src/ctc.py:210-352) '''

    def __init__(self, asr, vocab_range, beam_size, vocab_candidate,
                 lm_path='', lm_config='', lm_weight=0.0, device=None):
        super().__init__()
        self.asr = asr
        self.vocab_range = list(vocab_range)
        self.beam_size = beam_size
        self.vocab_cand = vocab_candidate
        assert self.vocab_cand <= len(self.vocab_range)
        assert self.asr.enable_ctc

        self.apply_lm = lm_weight > 0
        self.lm_w = 0
        if self.apply_lm:
            self.device = device
            self.lm_w = lm_weight
            self.lm_path = lm_path
            lm_config = yaml.load(open(lm_config, 'r'), Loader=yaml.FullLoader)
            self.lm = RNNLM(self.asr.vocab_size, **lm_config['model']).to(self.device)
            self.lm.load_state_dict(torch.load(self.lm_path, map_location='cpu')['model'])
            self.lm.eval()

    def create_msg(self):
        return ['Decode spec| CTC decoding \t| Beam size = {} \t| LM weight = {}'.format(
            self.beam_size, self.lm_w)]

    def _lm_step(self, token, hidden):
        dev = self.device
        out, hid = self.lm(torch.full((1, 1), int(token), dtype=torch.long, device=dev),
                           torch.ones(1, dtype=torch.long), hidden)
        logp = ops.log_softmax(out).reshape(-1)
        return _LMOut(logp.cpu().numpy(), logp), hid

    def _device_search_ok(self, V):
        """the gfx950 prefix-beam kernel (csrc/prefix_beam.hip) covers the reference's configurations:
        ascending vocab_range (the tie order of its candidate sort), beam <= 32, beam * (cand + 1) <= 1024,
        V <= 16384 (one row of masked scores is staged in LDS)"""
        import os
        vr = self.vocab_range
        return (os.environ.get('ASRK_CTC_BEAM_DEVICE', '1') != '0' and self.beam_size <= 32 and V <= 16384 and
                self.beam_size * (self.vocab_cand + 1) <= 1024 and all(a < b for a, b in zip(vr, vr[1:])))

    @torch.no_grad()
    def search_device(self, ctc_dev, t_start=None, defer=False):
        """The whole search of src/ctc.py:262-352 on the device for log-probs ctc_dev [T, V]: ONE launch for
        all frames without an LM; with LM fusion one launch per frame followed by the batched LM step for the
        rows the kernel marks, nothing read back until the final hypotheses (2 small D2H in total: the per-frame
        arg-max that decides the leading blank frames to skip, and the surviving token rows).
        t_start: first frame whose arg-max is not blank, when the caller already knows it (forward_batch);
        defer=True (no LM only): enqueue the launch on the current stream and return a function that reads the
        hypotheses back - so that several utterances' searches run side by side on different streams."""
        import ctypes
        from .. import _lib
        L = _lib.load()
        dev = ctc_dev.device
        T, V = ctc_dev.shape
        W, C = self.beam_size, self.vocab_cand
        if t_start is None:
            amax = ops.argmax(ctc_dev).cpu().numpy()
            nz = np.nonzero(amax != 0)[0]
            t_start = int(nz[0]) if len(nz) else -1
        if t_start < 0:
            return (lambda: [[]]) if defer else [[]]
        allowed = torch.zeros((V,), dtype=torch.uint8)
        allowed[torch.as_tensor(self.vocab_range)] = 1
        allowed = allowed.to(dev)
        nws = int(L.asrk_ctc_prefix_beam_ws_bytes(W, T))
        ws = torch.zeros((nws,), dtype=torch.uint8, device=dev)
        stream = ops._stream

        def launch(t0, t1, cur, init, lm_rows, follows):
            _lib.check(L.asrk_ctc_prefix_beam_f32(ops._p(ctc_dev), T, V, ops._p(allowed), W, C, ops._p(lm_rows),
                                                  float(self.lm_w), t0, t1, cur, int(init), int(follows),
                                                  ops._p(ws), nws, stream()), "ctc_prefix_beam")

        def offsets(buf):
            o = [ctypes.c_int64(0) for _ in range(6)]
            _lib.check(L.asrk_ctc_prefix_beam_ws_offsets(W, T, buf, *[ctypes.byref(v) for v in o]), "beam offsets")
            return [int(v.value) for v in o]

        def i32(off, n):
            return ws[off:off + 4 * n].view(torch.int32)

        def read_back(cur):
            nb_off, len_off, tok_off, _, _, _ = offsets(cur)
            nb = int(i32(nb_off, 1).cpu()[0])
            lens = i32(len_off, W).cpu().tolist()
            toks = i32(tok_off, W * (T + 1)).view(W, T + 1).cpu()
            return [toks[r, :lens[r]].tolist() for r in range(nb)]

        if defer:
            assert not self.apply_lm
            launch(t_start, T, 0, True, None, False)
            keep = (ctc_dev, allowed)                  # alive until the kernel has run
            return lambda: (keep, read_back((T - t_start) & 1))[1]

        if not self.apply_lm:
            launch(t_start, T, 0, True, None, False)
            cur = (T - t_start) & 1
        else:
            out, hid = self.lm(torch.zeros((1, 1), dtype=torch.long, device=dev), torch.ones(1, dtype=torch.long),
                               None)                                   # 0 == <sos> for RNNLM
            lstm = isinstance(hid, tuple)
            states = list(hid) if lstm else [hid]                      # each [n_layers, rows, dim]
            pad = lambda x, dim: torch.cat([x, x.new_zeros(*x.shape[:dim], W - x.shape[dim], *x.shape[dim + 1:])],
                                           dim)
            lm_rows = pad(ops.log_softmax(out).reshape(1, V), 0).contiguous()          # [W, V]
            states = [pad(x, 1).contiguous() for x in states]
            _, _, _, p_off, l_off, g_off = offsets(0)
            cur = 0
            for t in range(t_start, T):
                follows = t < T - 1
                launch(t, t + 1, cur, t == t_start, lm_rows, follows)
                cur ^= 1
                if follows:
                    parent, last, gidx = (i32(o, W).long() for o in (p_off, l_off, g_off))
                    h_in = [x.index_select(1, parent) for x in states]
                    out, hid = self.lm(last.view(W, 1), torch.ones(W, dtype=torch.long),
                                       (h_in[0], h_in[1]) if lstm else h_in[0])
                    stepped = list(hid) if lstm else [hid]
                    lm_rows = torch.cat([lm_rows, ops.log_softmax(out).reshape(W, V)], 0).index_select(0, gidx)
                    states = [torch.cat([o_, n_], 1).index_select(1, gidx) for o_, n_ in zip(states, stepped)]
        return read_back(cur)

    @torch.no_grad()
    def forward_batch(self, feat, feat_len, n_streams=16):
        ''' CTC prefix beam search over U utterances at once: feat [U,Tmax,D] zero-padded, feat_len [U] -> U result
            lists, each what forward() returns for that utterance alone.  One packed encoder pass (every utterance
            encoded as if alone, Encoder.forward(packed=True)), then every utterance's search - ONE launch of the
            one-workgroup prefix-beam kernel (csrc/prefix_beam.hip) - goes to its own stream: the searches run side
            by side on different CUs (the reference's parallel axis, bin/test_asr.py:163-167, on one GPU).  With LM
            fusion (one launch + one LM step per frame) or outside the kernel's limits: one forward() each. '''
        U = feat.shape[0]
        lens_h = [int(v) for v in torch.as_tensor(feat_len).cpu().tolist()]
        asr = self.asr
        one_by_one = lambda: [self.forward(feat[u:u + 1, :lens_h[u]].contiguous(),
                                           torch.as_tensor(feat_len)[u:u + 1]) for u in range(U)]
        if U == 1 or self.apply_lm or not hasattr(asr, 'encoder') or not asr.encoder.supports_packed() \
                or not self._device_search_ok(asr.vocab_size):
            return one_by_one()
        dev = feat.device
        enc, enc_len = asr.encoder(feat, torch.as_tensor(feat_len).to(dev), packed=True)
        ctc = ops.log_softmax(ops.linear(enc, asr.ctc_layer.weight, asr.ctc_layer.bias))
        ctc = ops.log_softmax(ctc)               # the reference re-applies log_softmax (forward() below)
        enc_len_h = [int(v) for v in asr.encoder.packed_frames.cpu().tolist()]   # forward() searches the whole tensor
        amax = ops.argmax(ctc).cpu().numpy()                                     # [U, T']
        main = torch.cuda.current_stream(dev)
        streams = [torch.cuda.Stream(device=dev) for _ in range(min(U, n_streams))]
        pending = []
        for u in range(U):
            nz = np.nonzero(amax[u, :enc_len_h[u]] != 0)[0]
            st = streams[u % len(streams)]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                x = ctc[u, :enc_len_h[u]].contiguous()
                pending.append((st, self.search_device(x, t_start=int(nz[0]) if len(nz) else -1, defer=True)))
        out = []
        for st, fetch in pending:
            with torch.cuda.stream(st):
                out.append(fetch())
        for st in streams:
            main.wait_stream(st)
        ctc.record_stream(main)
        return out

    @torch.no_grad()
    def forward(self, feat, feat_len):
        assert feat.shape[0] == 1, "Batchsize == 1 is required for beam search"
        ctc_output, _, _, _, _ = self.asr(feat, feat_len, 10)
        # the reference re-applies log_softmax to the (already normalised) log-probs
        ctc_dev = ops.log_softmax(ctc_output[0])
        if self._device_search_ok(ctc_dev.shape[1]):
            return self.search_device(ctc_dev.contiguous())
        return self._search_host(ctc_dev)

    def _search_host(self, ctc_dev):
        """configurations outside the kernel's limits (unordered vocab_range, beam > 32): the reference's
        bookkeeping as host objects, candidate ranking on the device (the round-2 path)"""
        ctc_output = ctc_dev.cpu().numpy()
        T = len(ctc_output)
        vr = np.asarray(self.vocab_range)
        # The per-hypothesis "sort the vocabulary by CTC (+LM) score, take the best vocab_candidate"
        # (src/ctc.py:296-303: a 5000-element Python sort per hypothesis and frame) is a device top-k:
        # same ranking (descending score, ties in vocab_range order = ascending id), same f32 arithmetic.
        allowed = torch.full((ctc_dev.shape[1],), float('-inf'), device=ctc_dev.device)
        allowed[torch.as_tensor(self.vocab_range, device=ctc_dev.device)] = 0.0
        ascending = all(a < b for a, b in zip(self.vocab_range, self.vocab_range[1:]))
        cand_all = None
        if ascending and not self.apply_lm:        # candidates do not depend on the hypothesis: all frames at once
            cand_all = ops.topk(ctc_dev + allowed, self.vocab_cand)[1].cpu().numpy()

        B = [CTCHypothesis()]
        if self.apply_lm:
            B[0].update_lm(*self._lm_step(0, None))           # 0 == <sos> for RNNLM

        start = True
        for t in range(T):
            # greedily ignoring pads at the beginning of the sequence
            if np.argmax(ctc_output[t]) == 0 and start:
                continue
            start = False
            B_new = []
            cand_t = None
            if ascending and self.apply_lm:        # one batched top-k for the live hypotheses of this frame
                live = [i for i in range(len(B)) if not (B[i].get_len() > 0 and B[i].y[-1] == 1)]
                if live:
                    lm_stack = torch.stack([B[i].lm_output.dev for i in live])
                    sc = (ctc_dev[t].unsqueeze(0) + self.lm_w * lm_stack) + allowed
                    top = ops.topk(sc, self.vocab_cand)[1].cpu().numpy()
                    cand_t = {i: top[q] for q, i in enumerate(live)}
            for i in range(len(B)):
                B_i_new = B[i].clone()
                if B_i_new.get_len() > 0:
                    if B_i_new.y[-1] == 1:                     # <eos>: finished
                        B_new.append(B_i_new)
                        continue
                    B_i_new.update_Pr_nblank(ctc_output[t, B_i_new.y[-1]])
                    for j in range(len(B)):                    # extension of another live prefix?
                        if i != j and B[j].check_same(B_i_new.y[:-1]):
                            lm_prob = self.lm_w * B[j].lm_output[B_i_new.y[-1]] if self.apply_lm else 0.0
                            B_i_new.update_Pr_nblank_prefix(ctc_output[t, B_i_new.y[-1]],
                                                            B[j].Pr_y_t_blank, B[j].Pr_y_t_nblank, lm_prob)
                            break
                B_i_new.update_Pr_blank(ctc_output[t, 0])      # 0 == <pad>/blank
                lm_probs = B_i_new.lm_output if self.apply_lm else None

                if cand_all is not None:
                    ks = cand_all[t]
                elif cand_t is not None:
                    ks = cand_t[i]
                else:                                          # unordered vocab_range: the reference's own sort
                    scores = ctc_output[t, vr] + (self.lm_w * lm_probs[vr] if self.apply_lm else 0.0)
                    # python's sorted(reverse=True) is stable: ties keep vocab_range order
                    order = sorted(range(len(vr)), key=lambda q: scores[q], reverse=True)
                    ks = [vr[order[j]] for j in range(self.vocab_cand)]
                for j in range(self.vocab_cand):
                    k = int(ks[j])
                    hyp_yk = B_i_new.clone()
                    lm_prob = 0.0 if not self.apply_lm else self.lm_w * lm_probs[k]
                    hyp_yk.add_token(k, ctc_output[t, k], lm_prob)
                    hyp_yk.updated_lm = False
                    B_new.append(hyp_yk)
                B_i_new.orig_backup()
                B_new.append(B_i_new)

            # Remove duplicated sequences by sorting first
            B_new = sorted(B_new, key=lambda h: h.get_string())
            B = [B_new[0]]
            for i in range(1, len(B_new)):
                if B_new[i].check_same(B[-1].y):
                    if B_new[i].get_score() > B[-1].get_score():
                        B[-1] = B_new[i]
                else:
                    B.append(B_new[i])

            if t == T - 1:
                B = sorted(B, reverse=True, key=lambda h: h.get_final_score())
            else:
                B = sorted(B, reverse=True, key=lambda h: h.get_score())
            B = B[:self.beam_size]

            if self.apply_lm and t < T - 1:
                for h in B:
                    if h.get_len() > 0 and not h.updated_lm:
                        h.update_lm(*self._lm_step(h.y[-1], h.lm_hidden))

        return [b.y for b in B]
