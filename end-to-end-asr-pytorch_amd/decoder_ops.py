"""Autograd Functions for the attention-decoder loop (reference: src/asr.py:101-153, 214-221,
277-313; src/module.py:179-258) on top of the libasrk attention / cell / embedding kernels.

A decode loop runs L dependent steps.  Per-step autograd nodes would normally hand back one
gradient tensor per step for every tensor that is shared by all steps (encoder memory `value`
52 MB, attention key 7.7 MB, decoder LSTM weights 67 MB at cfg3) and let autograd sum them.
Instead every shared tensor goes through a *hub*: the hub's forward returns a scalar `token` that
each step takes as an input; a step's backward accumulates / stashes its contribution on a Python
side `tape` and only returns a zero gradient for the token; because the hub sits upstream of every
step, autograd runs its backward last, where the deferred gradients are produced with ONE batched
GEMM / column sum per tensor (K = L*B instead of L GEMMs with K = B).
"""
import torch
from torch.autograd import Function

from . import _lib
from . import ops as _ops
from .ops import _L, _p, _stream, _f32c, _require_gpu, gemm, colsum, copy3d


def _zero_token_grad(dev):
    return torch.zeros((), dtype=torch.float32, device=dev)


# ------------------------------------------------------------------------------ embedding
class EmbeddingFn(Function):
    """nn.Embedding lookup (src/asr.py:103-109,134,142)."""

    @staticmethod
    def forward(ctx, idx, weight):
        _require_gpu(weight)
        w = _f32c(weight)
        ix = idx.to(device=w.device, dtype=torch.int64).contiguous()
        V, D = w.shape
        out = torch.empty(ix.shape + (D,), dtype=torch.float32, device=w.device)
        _lib.check(_L().asrk_embedding_fwd_f32(_p(ix), _p(w), _p(out), ix.numel(), D, V, _stream()),
                   "embedding")
        ctx.save_for_backward(ix)
        ctx.shape = (V, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ix,) = ctx.saved_tensors
        V, D = ctx.shape
        g = _f32c(dout)
        dW = _ops.zeros((V, D), g.device)
        _lib.check(_L().asrk_embedding_bwd_f32(_p(ix), _p(g), _p(dW), ix.numel(), D, V, _stream()),
                   "embedding_bwd")
        return None, dW


def embedding(idx, weight):
    if not torch.is_grad_enabled():              # inference: no autograd node
        _require_gpu(weight)
        w = _f32c(weight)
        ix = idx.to(device=w.device, dtype=torch.int64).contiguous()
        V, D = w.shape
        out = torch.empty(ix.shape + (D,), dtype=torch.float32, device=w.device)
        _lib.check(_L().asrk_embedding_fwd_f32(_p(ix), _p(w), _p(out), ix.numel(), D, V, _stream()), "embedding")
        return out
    return EmbeddingFn.apply(idx, weight)


# ------------------------------------------------------------------------------ attention
class AttnTape:
    """Per-utterance-batch attention memory + deferred-gradient state."""

    def __init__(self, mode, key, value, lens, num_head, temperature, loc_w=None):
        self.mode = mode                      # 'loc' | 'dot'
        self.key = key                        # [BN,T,A]
        self.value = value                    # [BN,T,Dv]
        self.lens = lens.to(device=key.device, dtype=torch.int64).contiguous()
        self.N = num_head
        self.temperature = float(temperature)
        self.BN, self.T, self.A = key.shape
        self.B = self.BN // num_head
        self.Dv = value.shape[-1]
        if mode == 'loc':
            self.Wc, self.Wp, self.we, self.be = loc_w   # [K,N,KW], [A,K], [A], [1]
            self.K = self.Wc.shape[0]
            self.ks = (self.Wc.shape[2] - 1) // 2
        self.attn_steps, self.dctx_steps = [], []
        self.acc = None

    def accumulators(self):
        if self.acc is None:
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.key.device)
            self.acc = dict(dkey=z(self.BN, self.T, self.A))
            if self.mode == 'loc':
                self.acc.update(dWc=z(*self.Wc.shape), dWp=z(*self.Wp.shape), dwe=z(self.A), dbe=z(1))
        return self.acc


class AttnHubFn(Function):
    """Identity hub in front of the decode loop for (key, value, loc weights); see module doc."""

    @staticmethod
    def forward(ctx, tape, key, value, *loc_w):
        ctx.tape = tape
        ctx.n_loc = len(loc_w)
        return torch.zeros((), dtype=torch.float32, device=key.device)

    @staticmethod
    def backward(ctx, g_token):
        t = ctx.tape
        dev = t.key.device
        acc = t.accumulators()
        dkey = acc['dkey']
        # dValue[bn] = sum_l attn_l[bn]^T (x) dctx_l[bn]  ==  A_bn^T [T,L] @ G_bn [L,Dv]
        dvalue = torch.zeros_like(t.value)
        L = len(t.attn_steps)
        if L > 0:
            A_all = torch.empty((t.BN, L, t.T), dtype=torch.float32, device=dev)
            G_all = torch.empty((t.BN, L, t.Dv), dtype=torch.float32, device=dev)
            for l in range(L):   # pack the stashed per-step rows (strided device copies)
                copy3d(t.attn_steps[l], A_all[:, l], 1, t.BN, t.T, 0, t.T, 0, L * t.T)
                copy3d(t.dctx_steps[l], G_all[:, l], 1, t.BN, t.Dv, 0, t.Dv, 0, L * t.Dv)
            for bn in range(t.BN):
                gemm(1, 0, t.T, t.Dv, L, A_all[bn], t.T, G_all[bn], t.Dv, dvalue[bn], t.Dv)
        t.attn_steps, t.dctx_steps = [], []
        if t.mode == 'loc':
            loc_grads = (acc['dWc'], acc['dWp'], acc['dwe'], acc['dbe'])
        else:
            loc_grads = ()
        return (None, dkey, dvalue) + loc_grads


class AttnStepFn(Function):
    """One attention step.  inputs: q [BN,A], prev_att [B,N,T] (loc) -> attn [B,N,T], ctx [BN,Dv]."""

    @staticmethod
    def forward(ctx, tape, token, q, prev_att):
        L = _L()
        t = tape
        dev = q.device
        qc = _f32c(q)
        attn = torch.empty((t.B, t.N, t.T), dtype=torch.float32, device=dev)
        e_scr = torch.empty((t.BN, t.T), dtype=torch.float32, device=dev)
        c = None
        if t.mode == 'loc':
            pc = _f32c(prev_att)
            c = torch.empty((t.B, t.T, t.K), dtype=torch.float32, device=dev)
            _lib.check(L.asrk_loc_conv_fwd_f32(_p(pc), _p(t.Wc), _p(c), t.B, t.N, t.T, t.K, t.ks,
                                               _stream()), "loc_conv")
            _lib.check(L.asrk_attn_energy_fwd_f32(1, _p(t.key), _p(qc), _p(c), _p(t.Wp), _p(t.we),
                                                  _p(t.be), _p(t.lens), _p(attn), _p(e_scr), t.B, t.N,
                                                  t.T, t.A, t.K, t.temperature, _stream()),
                       "attn_energy")
        else:
            pc = None
            _lib.check(L.asrk_attn_energy_fwd_f32(0, _p(t.key), _p(qc), None, None, None, None,
                                                  _p(t.lens), _p(attn), _p(e_scr), t.B, t.N, t.T, t.A,
                                                  0, t.temperature, _stream()), "attn_energy")
        context = torch.empty((t.BN, t.Dv), dtype=torch.float32, device=dev)
        _lib.check(L.asrk_attn_context_fwd_f32(_p(attn), _p(t.value), _p(context), t.BN, t.T, t.Dv,
                                               t.Dv, _stream()), "attn_context")
        ctx.tape = t
        ctx.save_for_backward(qc, pc, c, attn)
        return attn, context

    @staticmethod
    def backward(ctx, dattn_out, dctx):
        L = _L()
        t = ctx.tape
        qc, pc, c, attn = ctx.saved_tensors
        dev = qc.device
        acc = t.accumulators()
        dattn = torch.zeros((t.BN, t.T), dtype=torch.float32, device=dev)
        if dctx is not None:
            g = _f32c(dctx)
            _lib.check(L.asrk_attn_context_bwd_f32(_p(g), _p(t.value), _p(dattn), t.BN, t.T, t.Dv,
                                                   t.Dv, _stream()), "attn_context_bwd")
            t.attn_steps.append(attn.view(t.BN, t.T))
            t.dctx_steps.append(g)
        if dattn_out is not None:
            copy3d(_f32c(dattn_out), dattn, 1, t.BN, t.T, 0, t.T, 0, t.T, accumulate=True)
        dq = torch.empty((t.BN, t.A), dtype=torch.float32, device=dev)
        dprev = None
        if t.mode == 'loc':
            dc = torch.empty((t.B, t.T, t.K), dtype=torch.float32, device=dev)
            _lib.check(L.asrk_attn_energy_bwd_f32(1, _p(t.key), _p(qc), _p(c), _p(t.Wp), _p(t.we),
                                                  _p(t.lens), _p(attn), _p(dattn), _p(acc['dkey']),
                                                  _p(dq), _p(dc), _p(acc['dWp']), _p(acc['dwe']),
                                                  _p(acc['dbe']), t.B, t.N, t.T, t.A, t.K,
                                                  t.temperature, _stream()), "attn_energy_bwd")
            need_dprev = ctx.needs_input_grad[3]
            dprev = torch.empty_like(pc) if need_dprev else None
            _lib.check(L.asrk_loc_conv_bwd_f32(_p(dc), _p(pc), _p(t.Wc), _p(dprev), _p(acc['dWc']),
                                               t.B, t.N, t.T, t.K, t.ks, _stream()), "loc_conv_bwd")
        else:
            _lib.check(L.asrk_attn_energy_bwd_f32(0, _p(t.key), _p(qc), None, None, None, _p(t.lens),
                                                  _p(attn), _p(dattn), _p(acc['dkey']), _p(dq), None,
                                                  None, None, None, t.B, t.N, t.T, t.A, 0,
                                                  t.temperature, _stream()), "attn_energy_bwd")
        return None, _zero_token_grad(dev), dq, dprev


# ------------------------------------------------------------------------------ decoder LSTM cell
class CellTape:
    """Deferred weight gradients of one decoder LSTM layer over a whole decode loop."""

    def __init__(self, w_ih, w_hh, b_ih, b_hh, B, capacity):
        self.w_ih, self.w_hh, self.b_ih, self.b_hh = w_ih, w_hh, b_ih, b_hh
        self.B, self.H = B, w_hh.shape[1]
        self.In = w_ih.shape[1]
        self.cap = capacity
        dev = w_ih.device
        f = dict(dtype=torch.float32, device=dev)
        self.X = torch.empty((capacity * B, self.In), **f)     # step inputs
        self.Hp = torch.empty((capacity * B, self.H), **f)     # previous hidden states
        self.dG = None                                         # pre-activation gradients
        self.used = 0
        self.bwd_mask = None

    def slot(self):
        if self.used >= self.cap:
            raise RuntimeError("CellTape capacity exceeded")
        s = self.used
        self.used += 1
        return s


class CellHubFn(Function):
    @staticmethod
    def forward(ctx, tape, w_ih, w_hh, b_ih, b_hh):
        ctx.tape = tape
        return torch.zeros((), dtype=torch.float32, device=w_ih.device)

    @staticmethod
    def backward(ctx, g_token):
        t = ctx.tape
        dev = t.w_ih.device
        f = dict(dtype=torch.float32, device=dev)
        H4 = 4 * t.H
        dw_ih = torch.zeros((H4, t.In), **f)
        dw_hh = torch.zeros((H4, t.H), **f)
        db = torch.zeros((H4,), **f)
        if t.dG is not None and t.used > 0:
            M = t.used * t.B
            gemm(1, 0, H4, t.In, M, t.dG, H4, t.X, t.In, dw_ih, t.In)
            gemm(1, 0, H4, t.H, M, t.dG, H4, t.Hp, t.H, dw_hh, t.H)
            colsum(t.dG, M, H4, H4, db)
        return None, dw_ih, dw_hh, db, db.clone()


class LSTMCellStepFn(Function):
    """One step of nn.LSTM layer `l` on a length-1 sequence (src/asr.py:218)."""

    @staticmethod
    def forward(ctx, tape, token, x, h, c):
        L = _L()
        t = tape
        dev = x.device
        s = t.slot()
        B, H = t.B, t.H
        xs = t.X[s * B:(s + 1) * B]
        hs = t.Hp[s * B:(s + 1) * B]
        xs.copy_(x)      # device-to-device staging into the tape (plumbing)
        hs.copy_(h)
        cc = _f32c(c)
        gates = torch.empty((B, 4 * H), dtype=torch.float32, device=dev)
        gemm(0, 1, B, 4 * H, t.In, xs, t.In, t.w_ih, t.In, gates, 4 * H, bias=t.b_ih, bias2=t.b_hh)
        gemm(0, 1, B, 4 * H, H, hs, H, t.w_hh, H, gates, 4 * H, beta=1.0)
        c_new = torch.empty((B, H), dtype=torch.float32, device=dev)
        h_new = torch.empty((B, H), dtype=torch.float32, device=dev)
        _lib.check(L.asrk_lstm_cell_fwd_f32(_p(gates), _p(cc), _p(c_new), _p(h_new), B, H, _stream()),
                   "lstm_cell")
        ctx.tape, ctx.slot = t, s
        ctx.save_for_backward(gates, cc, c_new)
        return h_new, c_new

    @staticmethod
    def backward(ctx, dh, dc):
        L = _L()
        t, s = ctx.tape, ctx.slot
        gates, c_prev, c_new = ctx.saved_tensors
        dev = gates.device
        B, H = t.B, t.H
        if t.dG is None:
            t.dG = torch.zeros((t.cap * B, 4 * H), dtype=torch.float32, device=dev)
        dG = t.dG[s * B:(s + 1) * B]
        dG.copy_(gates)
        dhc = _f32c(dh) if dh is not None else None
        dcc = _f32c(dc) if dc is not None else None
        dc_prev = torch.empty((B, H), dtype=torch.float32, device=dev)
        _lib.check(L.asrk_lstm_cell_bwd_f32(_p(dG), _p(c_prev), _p(c_new), _p(dhc), _p(dcc), _p(dc_prev),
                                            B, H, _stream()), "lstm_cell_bwd")
        dx = torch.empty((B, t.In), dtype=torch.float32, device=dev)
        gemm(0, 0, B, t.In, 4 * H, dG, 4 * H, t.w_ih, t.In, dx, t.In)
        dh_prev = torch.empty((B, H), dtype=torch.float32, device=dev)
        gemm(0, 0, B, H, 4 * H, dG, 4 * H, t.w_hh, H, dh_prev, H)
        return None, _zero_token_grad(dev), dx, dh_prev, dc_prev


class ConcatLastFn(Function):
    """torch.cat([a, b], dim=-1) for 2-D tensors (decoder input [last_char ; context])."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a)
        ac, bc = _f32c(a), _f32c(b)
        R, Da = ac.shape
        Db = bc.shape[1]
        out = torch.empty((R, Da + Db), dtype=torch.float32, device=a.device)
        copy3d(ac, out, 1, R, Da, 0, Da, 0, Da + Db)
        copy3d(bc, out[:, Da:], 1, R, Db, 0, Db, 0, Da + Db)
        ctx.dims = (R, Da, Db)
        return out

    @staticmethod
    def backward(ctx, g):
        R, Da, Db = ctx.dims
        gc = _f32c(g)
        ga = torch.empty((R, Da), dtype=torch.float32, device=g.device)
        gb = torch.empty((R, Db), dtype=torch.float32, device=g.device)
        copy3d(gc, ga, 1, R, Da, 0, Da + Db, 0, Da)
        copy3d(gc[:, Da:], gb, 1, R, Db, 0, Da + Db, 0, Db)
        return ga, gb


def concat_last(a, b):
    return ConcatLastFn.apply(a, b)


class StackStepsFn(Function):
    """torch.stack(list_of [R, D] tensors, dim=1) -> [R, L, D] (att_output / dec_state / att_seq)."""

    @staticmethod
    def forward(ctx, *steps):
        _require_gpu(steps[0])
        L = len(steps)
        R, D = steps[0].shape
        out = torch.empty((R, L, D), dtype=torch.float32, device=steps[0].device)
        for l, s in enumerate(steps):
            copy3d(_f32c(s), out[:, l], 1, R, D, 0, D, 0, L * D)
        ctx.dims = (R, L, D)
        return out

    @staticmethod
    def backward(ctx, g):
        R, L, D = ctx.dims
        gc = _f32c(g)
        outs = []
        for l in range(L):
            o = torch.empty((R, D), dtype=torch.float32, device=g.device)
            copy3d(gc[:, l], o, 1, R, D, 0, L * D, 0, D)
            outs.append(o)
        return tuple(outs)


def stack_steps(steps):
    return StackStepsFn.apply(*steps)


# ------------------------------------------------------------------------------ inference helpers
# split panels of inference weights (beam search over many rows): built at a weight's first use in a decode call and reused
# by every later position of that call.  The key carries the tensor's version counter, but the fused optimiser writes
# parameters through raw pointers (no version bump), so a search never trusts panels of an earlier call:
# BeamDecoder._search_rows starts with drop_weight_panels() - ~0.15 ms of split passes per decode call at cfg5 widths.
_weight_panels = {}


def drop_weight_panels():
    _weight_panels.clear()


def weight_panel(w):
    """the bf16x3 split panel of a [rows, K] weight matrix, cached by (storage address, version, shape)"""
    from .ops import SplitPanel
    key = (w.data_ptr(), w._version, tuple(w.shape), w.device.index)
    p = _weight_panels.get(key)
    if p is None:
        if len(_weight_panels) >= 32:
            _weight_panels.clear()
        wc = _f32c(w.detach())
        p = _weight_panels[key] = (SplitPanel(wc, wc.shape[1], wc.shape[0], wc.shape[1], False), wc)
    return p[0]


LSTM_CELL_GEMM_ROWS = 128


def linear_infer(x, weight, bias=None):
    """x [B, K] W^T + b without autograd, for decode-time projections onto the vocabulary: with many rows (the beams of a
    batch of utterances) a bf16x6 panel GEMM against the cached split panel of the weight, else ops.linear."""
    B, K = x.shape
    if B < LSTM_CELL_GEMM_ROWS or K % 32 or x.dim() != 2:
        return _ops.linear(x, weight, bias)
    from .ops import SplitPanel, gemm_panels
    _require_gpu(x)
    xc = _f32c(x)
    N = weight.shape[0]
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    gemm_panels(B, N, K, SplitPanel(xc, K, B, K, False), 0, 0, weight_panel(weight), 0, 0, y, N,
                bias=_f32c(bias) if bias is not None else None)
    return y


def lstm_cell_infer(x, h, c, w_ih, w_hh, b_ih, b_hh, out=None):
    """One LSTM cell step without autograd bookkeeping (beam search / RNN-LM fusion).
    Few rows (one utterance's beam): input projection, recurrent projection and cell update in one weight-streaming
    kernel (csrc/speller.hip).  Many rows (the beams of a batch of utterances: 512 at 32 x beam 16): that kernel is a
    64-row tile that re-streams the whole weight per tile and multiplies on the f32-input matrix cores (8 launches of
    22 us per layer); here the two projections are bf16x6 panel GEMMs against weight panels split ONCE per decode
    (weights do not change), and the cell is the pointwise kernel."""
    B = x.shape[0]
    H = h.shape[1]
    In = x.shape[1]
    if B < LSTM_CELL_GEMM_ROWS or In % 32 or H % 32:
        from .speller_ops import lstm_cell_fused
        return lstm_cell_fused(x, h, c, w_ih, w_hh, b_ih, b_hh, out=out)
    from .ops import SplitPanel, gemm_panels
    _require_gpu(x)
    xc, hc, cc = _f32c(x), _f32c(h), _f32c(c)
    gates = torch.empty((B, 4 * H), dtype=torch.float32, device=x.device)
    gemm_panels(B, 4 * H, In, SplitPanel(xc, In, B, In, False), 0, 0, weight_panel(w_ih), 0, 0, gates, 4 * H,
                bias=_f32c(b_ih), bias2=_f32c(b_hh))
    gemm_panels(B, 4 * H, H, SplitPanel(hc, H, B, H, False), 0, 0, weight_panel(w_hh), 0, 0, gates, 4 * H, beta=1.0)
    if out is None:
        c_new = torch.empty((B, H), dtype=torch.float32, device=x.device)
        h_new = torch.empty((B, H), dtype=torch.float32, device=x.device)
    else:
        h_new, c_new = out
    _lib.check(_L().asrk_lstm_cell_fwd_f32(_p(gates), _p(cc), _p(c_new), _p(h_new), B, H, _stream()), "lstm_cell")
    return h_new, c_new


def expand_tape(tape, n):
    """AttnTape of a batch-1 utterance replicated to n hypotheses (rows ordered (hyp, head))."""
    def rep(x):   # [N, T, D] -> [n*N, T, D]
        N, T, D = x.shape
        out = torch.empty((n * N, T, D), dtype=torch.float32, device=x.device)
        for i in range(n):
            copy3d(x, out[i * N:], 1, N * T, D, 0, D, 0, D)
        return out
    loc_w = (tape.Wc, tape.Wp, tape.we, tape.be) if tape.mode == 'loc' else None
    return AttnTape(tape.mode, rep(tape.key), rep(tape.value), tape.lens.repeat(n), tape.N,
                    tape.temperature, loc_w)


def attn_step_infer(tape, q, prev_att):
    """AttnStepFn.forward without autograd (q [BN,A], prev_att [B,N,T] or None)."""
    with torch.no_grad():
        return AttnStepFn.forward(_NullCtx(), tape, None, q, prev_att)


class _NullCtx:
    def save_for_backward(self, *a):
        pass


def joint_score(att_logp, cand, psi, prev_ctc, lm_logp, ctc_w, lm_w, logzero):
    """(1 - ctc_w) * att + ctc_w * hack(cand, psi - prev_ctc), [:,0] = logzero, + lm_w * lm  in one
    kernel (src/decode.py:130-148); cand None = no CTC term, lm_logp None = no LM term."""
    _require_gpu(att_logp)
    a = _f32c(att_logp)
    n, V = a.shape
    out = torch.empty_like(a)
    C = 0
    if cand is not None:
        cand = cand.to(torch.int64).contiguous()
        psi, prev_ctc = _f32c(psi), _f32c(prev_ctc)
        C = cand.shape[1]
    lm = _f32c(lm_logp) if lm_logp is not None else None
    _lib.check(_L().asrk_joint_score_f32(_p(a), _p(cand), _p(psi), _p(prev_ctc), _p(lm), _p(out), n, V, C,
                                         float(ctc_w), float(lm_w), float(logzero), _stream()), "joint_score")
    return out


def ctc_prefix_scores(x, r_prev, prefix_len, last_char, candidates, blank=0, eos=1, logzero=-1e8,
                      row_mem=None, mem_len=None):
    """Batched CTCPrefixScore.cheap_compute on the device (src/ctc.py:76-116).
    x [T,V]; r_prev [n,T,2]; prefix_len/last_char [n] int; candidates [n,C] int
    -> psi [n,C], r [n,C,T,2].
    Several utterances at once: x [U,T,V] (padded to the longest), row_mem [n] int32 = utterance of every
    hypothesis, mem_len [U] int32 = frames of every utterance."""
    _require_gpu(x)
    xc = _f32c(x)
    multi = row_mem is not None
    U = xc.shape[0] if multi else 1
    T, V = xc.shape[-2], xc.shape[-1]
    rp = _f32c(r_prev)
    n, C = candidates.shape
    i32 = lambda t: torch.as_tensor(t).to(device=x.device, dtype=torch.int32).contiguous()
    pl, lc, cd = i32(prefix_len), i32(last_char), i32(candidates)
    psi = torch.empty((n, C), dtype=torch.float32, device=x.device)
    r = torch.empty((n, C, T, 2), dtype=torch.float32, device=x.device)
    if multi:
        rm, ml = i32(row_mem), i32(mem_len)
        _lib.check(_L().asrk_ctc_prefix_score_multi_f32(_p(xc), _p(rm), _p(ml), _p(rp), _p(pl), _p(lc), _p(cd),
                                                        _p(psi), _p(r), n, C, T, V, U, blank, eos, logzero,
                                                        _stream()), "ctc_prefix_score_multi")
    else:
        _lib.check(_L().asrk_ctc_prefix_score_f32(_p(xc), _p(rp), _p(pl), _p(lc), _p(cd), _p(psi), _p(r), n,
                                                  C, T, V, blank, eos, logzero, _stream()),
                   "ctc_prefix_score")
    return psi, r


def gather_rows_multi(segs, parent, col=None):
    """dst[i, :] = src[parent[i] * mul + (col[i] if use_col else 0), :] for up to 8 (src, dst, row_floats, mul, use_col)
    segments in one launch (asrk_gather_rows_multi_f32): the survivors' states of a beam-search position.  src / dst
    contiguous float32; parent / col int64 [n] on the device."""
    import ctypes
    n = int(parent.shape[0])
    k = len(segs)
    if k == 0 or n == 0:
        return
    _require_gpu(parent)
    for src, dst, rf, mul, uc in segs:
        assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.float32
        assert dst.numel() == n * rf and src.numel() % rf == 0
    vp, ci = ctypes.c_void_p * k, ctypes.c_int * k
    _lib.check(_L().asrk_gather_rows_multi_f32(
        k, vp(*[s[0].data_ptr() for s in segs]), vp(*[s[1].data_ptr() for s in segs]), ci(*[int(s[2]) for s in segs]),
        ci(*[int(s[3]) for s in segs]), ci(*[int(bool(s[4])) for s in segs]), _p(parent), _p(col), n, _stream()),
        "gather_rows_multi")
