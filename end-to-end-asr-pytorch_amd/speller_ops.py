"""The teacher-forced attention-decoder loop as ONE autograd node (reference: the loop of
src/asr.py:112-148 with tf_rate == 1; Attention.forward src/asr.py:277-313; LocationAwareAttention
src/module.py:234-258; Decoder.forward src/asr.py:214-221).

`asrk_speller_fwd_f32` / `asrk_speller_bwd_f32` (csrc/speller.hip) enqueue every step of the loop from
C++ (4 kernels per forward step, 5 per backward step, no per-step allocation); what is left here is
buffer ownership and the whole-sequence GEMMs on either side of the loop:
  before:  eproj = [<sos> ; teacher embeddings] W_ih[:, :E]^T + b_ih + b_hh   (inputs are known)
  after :  dW_ih, dW_hh, db, dWq, dbq from the tape with K = L*B instead of L GEMMs with K = B;
           d(embeddings) = dG W_ih[:, :E];  d(value) in one kernel;  partial-sum reductions.
Scope: single-head location-aware attention and a single-layer LSTM decoder (the reference's shipped
configs); every other variant runs the per-step kernels of decoder_ops.py.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib
from . import ops
from .ops import _L, _p, _stream, _f32c, _require_gpu, gemm, colsum, copy3d

c_int, c_i64, c_f32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class SpellerT(ctypes.Structure):
    """struct asrk_speller (include/asrk.h)"""
    _fields_ = ([(n, c_int) for n in ("B", "Te", "A", "Dv", "K", "ks", "H", "E", "L")]
                + [("temperature", c_f32), ("shared_kv", c_int)]
                + [(n, c_vp) for n in ("key", "value", "lens", "Wq", "bq", "Wc", "Wp", "we", "be", "W_ih",
                                       "W_hh", "b_ih", "b_hh", "eproj", "q", "conv", "attn")]
                + [("attn_ld", c_i64), ("attn_step", c_i64)]
                + [(n, c_vp) for n in ("ctx", "gates", "h", "c", "states", "e_scratch", "prev0", "row_mem")]
                + [("cell", c_int), ("nlayer", c_int)]
                + [(n, c_vp * 2) for n in ("Wu_ih", "Wu_hh", "bu_ih", "bu_hh", "hu", "cu", "gu")]
                + [("att_mode", c_int), ("nhead", c_int), ("Wm", c_vp), ("bm", c_vp), ("ctxh", c_vp)]
                + [("row_group", c_int)])


class SpellerBwdT(ctypes.Structure):
    """struct asrk_speller_bwd (include/asrk.h)"""
    _fields_ = ([(n, c_vp) for n in ("dstates", "dattn_seq", "WT", "WqT", "dkey", "dxh", "dq_pre", "dattn",
                                     "dprev", "dconv", "dq_part", "dwe_part", "dWp_part", "dbe_part",
                                     "dWc_part", "dc")]
                + [("tc", c_int)]
                + [(n, c_vp * 2) for n in ("WuT", "dxu", "dcu")]
                + [("WmT", c_vp), ("dctxh", c_vp)])


def _ptr(t):
    return t.data_ptr() if t is not None else None


MAX_LOOP_LAYERS = 3          # ASRK_SPELLER_MAX_LAYERS (include/asrk.h)


def supported_train_loop(attention, decoder, training):
    """what the one-node TEACHER-FORCED loop (SpellerLoopFn) covers beyond supported_loop: stacked LSTM decoders of up
    to MAX_LOOP_LAYERS layers (round 6) - unless inter-layer dropout is live (nn.LSTM applies it between layers in
    training, src/asr.py:175-176; the loop has no mask tape for it)"""
    if supported_loop(attention, decoder):
        return True
    if os.environ.get('ASRK_SPELLER', '1') == '0':
        return False
    # attention: single-head location-aware, or dot-product with any number of heads (a multi-head location-aware layer
    # convolves across the heads' previous alignments: per-step kernels); decoder: one LSTM / GRU layer, or a stack of
    # LSTM layers whose inter-layer dropout is not live
    att_ok = (attention.mode == 'loc' and attention.num_head == 1) or (attention.mode == 'dot' and attention.dim <= 512)
    dec_ok = decoder.layer == 1 or (decoder.enable_cell and decoder.layer <= MAX_LOOP_LAYERS
                                    and not (training and decoder.dropout > 0))
    return att_ok and dec_ok


def _set_slots(field, tensors):
    for i, t in enumerate(tensors):
        field[i] = _ptr(t)


def supported_loop(attention, decoder):
    """the configurations the one-node teacher-forced loop and the single fused step (SpellerStepper: greedy
    and beam decoding, pass 1 of scheduled sampling) cover: single-head location-aware attention over a one-layer LSTM or GRU decoder (the cell
    epilogues know both cells; a GRU's steppers carry an unused c slot)"""
    if os.environ.get('ASRK_SPELLER', '1') == '0':
        return False
    return attention.mode == 'loc' and attention.num_head == 1 and decoder.layer == 1


def stack_gru_params(w_ih, w_hh, b_ih, b_hh):
    """nn.GRU parameters (rows r, z, n) -> the loop's four-rows-per-unit layout (asrk_speller_t::cell = 1):
    W_ih4 = [W_ir; W_iz; W_in; 0], W_hh4 = [W_hr; W_hz; 0; W_hn], biases alike.  Plain torch ops: autograd hands the
    loop's weight gradients back to the right blocks and drops those of the zero blocks."""
    H = w_hh.shape[1]
    z = lambda *shape: torch.zeros(shape, dtype=w_ih.dtype, device=w_ih.device)
    return (torch.cat([w_ih, z(H, w_ih.shape[1])], 0), torch.cat([w_hh[:2 * H], z(H, H), w_hh[2 * H:]], 0),
            torch.cat([b_ih, z(H)], 0), torch.cat([b_hh[:2 * H], z(H), b_hh[2 * H:]], 0))


def uniform_attention(lens, Te):
    """prev_att of the first step: 1/len_b on the valid frames (src/module.py:239-242) -> [B,Te]"""
    idx = torch.arange(Te, device=lens.device).unsqueeze(0)
    valid = (idx < lens.unsqueeze(1)).to(torch.float32)
    return (valid / lens.clamp(min=1).to(torch.float32).unsqueeze(1)).contiguous()


def transpose_into(src, ldi, rows, cols, dst, ldo):
    _lib.check(_L().asrk_transpose_ld_f32(_p(src), ldi, _p(dst), ldo, rows, cols, _stream()), "transpose")


class SpellerLoopFn(Function):
    """(key [B,Te,A], value [B,Te,Dv], lens [B], sos_emb [B,E], teacher_emb [B,L,E], weights...) ->
    (states [B,L,H] = decoder outputs of every step, att_seq [B,1,L,Te])."""

    @staticmethod
    def forward(ctx, key, value, lens, sos_emb, teacher_emb, Wq, bq, Wc, Wp, we, be, W_ih, W_hh, b_ih,
                b_hh, L, temperature, cell=0, nhead=1, Wm=None, bm=None, *upper):
        """upper: (W_ih_l, W_hh_l, b_ih_l, b_hh_l) of the decoder's layers 1, 2, ... (stacked LSTM decoder); Wq is then
        [heads * A, layers * H] and `states` are the top layer's outputs.  Wc is None = dot-product attention (no Wc / Wp
        / we / be); nhead > 1 (dot only): key [B*N,Te,A] / value [B*N,Te,Dv] hold the heads' rows (b * N + n), Wm / bm
        are merge_head's, att_seq is [B,N,L,Te]"""
        _require_gpu(key)
        lib = _L()
        dev = key.device
        key, value = _f32c(key), _f32c(value)
        NH = int(nhead)
        dot = Wc is None
        BN, Te, A = key.shape
        B = BN // NH
        Dv = value.shape[2]
        Wq, bq = _f32c(Wq), _f32c(bq)
        Wc, Wp, we, be = (None,) * 4 if dot else tuple(_f32c(t) for t in (Wc, Wp, we, be))
        Wm, bm = (_f32c(Wm), _f32c(bm)) if NH > 1 else (None, None)
        W_ih, W_hh, b_ih, b_hh = (_f32c(t) for t in (W_ih, W_hh, b_ih, b_hh))
        H = W_hh.shape[1]
        E = W_ih.shape[1] - Dv
        K, ks = (0, 0) if dot else (Wc.shape[0], (Wc.shape[2] - 1) // 2)
        upper = [_f32c(t) for t in upper]
        NL = 1 + len(upper) // 4
        if ((not dot and (Wc.shape[1] != 1 or Wp.shape != (A, K) or we.numel() != A or NH != 1))
                or BN != B * NH or value.shape[0] != BN or Wq.shape != (NH * A, NL * H) or E <= 0
                or (NH > 1 and Wm.shape != (Dv, NH * Dv))
                or len(upper) % 4 or NL > MAX_LOOP_LAYERS or (NL > 1 and cell != 0)
                or any(upper[4 * i].shape != (4 * H, H) or upper[4 * i + 1].shape != (4 * H, H) for i in range(NL - 1))):
            raise _lib.AsrkError("speller loop: unsupported attention/decoder shapes")
        lens = lens.to(device=dev, dtype=torch.int64).contiguous()
        f = dict(dtype=torch.float32, device=dev)
        # decoder inputs of all steps, time-major: <sos>, then teacher tokens 0..L-2 (src/asr.py:103-104,122)
        emb_tm = torch.empty((L, B, E), **f)
        copy3d(_f32c(sos_emb), emb_tm, 1, B, E, 0, E, 0, E)
        if L > 1:
            te = _f32c(teacher_emb)
            copy3d(te, emb_tm[1:], L - 1, B, E, E, te.shape[1] * E, B * E, E)
        eproj = torch.empty((L * B, 4 * H), **f)
        gemm(0, 1, L * B, 4 * H, E, emb_tm, E, W_ih, E + Dv, eproj, 4 * H, bias=b_ih, bias2=b_hh)

        tape = dict(q=torch.empty((L, B, NH * A), **f), conv=torch.empty((L, B, Te, K) if not dot else (1,), **f),
                    ctxh=torch.empty((L, B, NH * Dv) if NH > 1 else (1,), **f),
                    ctx=torch.empty((L, B, Dv), **f), gates=torch.empty((L, B, 4 * H), **f),
                    h=torch.empty((L + 1, B, H), **f),
                    c=torch.empty((L + 1, B, H) if cell == 0 else (1,), **f))   # the GRU cell has no c
        _lib.check(_L().asrk_fill_f32(_p(tape['h'][0]), tape['h'][0].numel(), 0.0, _stream()), 'fill')
        if cell == 0:
            _lib.check(_L().asrk_fill_f32(_p(tape['c'][0]), tape['c'][0].numel(), 0.0, _stream()), 'fill')
        up_tape = [dict(h=torch.empty((L + 1, B, H), **f), c=torch.empty((L + 1, B, H), **f),
                        g=torch.empty((L, B, 4 * H), **f)) for _ in range(NL - 1)]
        for ut in up_tape:
            for n_ in ('h', 'c'):
                _lib.check(_L().asrk_fill_f32(_p(ut[n_][0]), ut[n_][0].numel(), 0.0, _stream()), 'fill')
        states = torch.empty((B, L, H), **f)
        att_seq = torch.empty((B, NH, L, Te), **f)
        e_scratch = torch.empty((BN, Te), **f)
        prev0 = uniform_attention(lens, Te) if not dot else torch.empty((1,), **f)
        d = SpellerT(B, Te, A, Dv, K, ks, H, E, L, float(temperature), 0,
                     _ptr(key), _ptr(value), _ptr(lens), _ptr(Wq), _ptr(bq), _ptr(Wc), _ptr(Wp), _ptr(we),
                     _ptr(be), _ptr(W_ih), _ptr(W_hh), _ptr(b_ih), _ptr(b_hh), _ptr(eproj), _ptr(tape['q']),
                     _ptr(tape['conv']), _ptr(att_seq), L * Te, Te, _ptr(tape['ctx']), _ptr(tape['gates']),
                     _ptr(tape['h']), _ptr(tape['c']) if cell == 0 else None, _ptr(states), _ptr(e_scratch),
                     _ptr(prev0))
        d.cell = int(cell)
        d.nlayer = NL
        d.att_mode, d.nhead = (1 if dot else 0), NH
        d.Wm, d.bm, d.ctxh = _ptr(Wm), _ptr(bm), (_ptr(tape['ctxh']) if NH > 1 else None)
        _set_slots(d.Wu_ih, upper[0::4]); _set_slots(d.Wu_hh, upper[1::4])
        _set_slots(d.bu_ih, upper[2::4]); _set_slots(d.bu_hh, upper[3::4])
        _set_slots(d.hu, [u['h'] for u in up_tape]); _set_slots(d.cu, [u['c'] for u in up_tape])
        _set_slots(d.gu, [u['g'] for u in up_tape])
        _lib.check(lib.asrk_speller_fwd_f32(ctypes.byref(d), _stream()), "speller_fwd")
        ctx.cell = int(cell)
        ctx.nlayer = NL
        ctx.nhead, ctx.dot = NH, dot
        ctx.merge = (Wm, tape['ctxh']) if NH > 1 else None
        ctx.dims = (B, Te, A, Dv, K, ks, H, E, L, float(temperature), teacher_emb.shape[1])
        ctx.save_for_backward(key, value, lens, emb_tm, Wq, bq, Wc, Wp, we, be, W_ih, W_hh, tape['q'],
                              tape['conv'], tape['ctx'], tape['gates'], tape['h'], tape['c'], att_seq, prev0,
                              *upper[0::4], *upper[1::4], *[u[n_] for n_ in ('h', 'c', 'g') for u in up_tape])
        ctx.weight_refs = tuple(t_ for t_ in (Wq, bq, Wc, Wp, we, be, W_ih, W_hh, b_ih, b_hh, Wm, bm) + tuple(upper)
                                if t_ is not None)
        ctx.consumed = False
        return states, att_seq

    @staticmethod
    def backward(ctx, dstates, datt_seq):
        lib = _L()
        saved = ctx.saved_tensors
        (key, value, lens, emb_tm, Wq, bq, Wc, Wp, we, be, W_ih, W_hh, q, conv, ctx_all, gates, h, c,
         att_seq, prev0) = saved[:20]
        NL = ctx.nlayer
        nu = NL - 1
        Wu_ih, Wu_hh = saved[20:20 + nu], saved[20 + nu:20 + 2 * nu]
        hu, cu, gu = (saved[20 + (2 + i) * nu:20 + (3 + i) * nu] for i in range(3))
        if ctx.consumed:
            raise RuntimeError("SpellerLoopFn: backward twice (the gate tape is reused in place)")
        ctx.consumed = True
        B, Te, A, Dv, K, ks, H, E, L, temperature, Lt = ctx.dims
        NH, dot = ctx.nhead, ctx.dot
        BN, NA = B * NH, NH * A
        dev = key.device
        f = dict(dtype=torch.float32, device=dev)
        In, XH, KW = E + Dv, Dv + H, 2 * ks + 1
        dstates = _f32c(dstates) if dstates is not None else ops.zeros((B, L, H), dev)
        datt = _f32c(datt_seq) if datt_seq is not None else None
        d = SpellerT(B, Te, A, Dv, K, ks, H, E, L, temperature, 0,
                     _ptr(key), _ptr(value), _ptr(lens), _ptr(Wq), _ptr(bq), _ptr(Wc), _ptr(Wp), _ptr(we),
                     _ptr(be), _ptr(W_ih), _ptr(W_hh), None, None, None, _ptr(q), _ptr(conv), _ptr(att_seq),
                     L * Te, Te, _ptr(ctx_all), _ptr(gates), _ptr(h), _ptr(c) if ctx.cell == 0 else None, None, None,
                     _ptr(prev0))
        d.cell = ctx.cell
        d.nlayer = NL
        d.att_mode, d.nhead = (1 if dot else 0), NH
        _set_slots(d.Wu_ih, Wu_ih); _set_slots(d.Wu_hh, Wu_hh)
        _set_slots(d.hu, hu); _set_slots(d.cu, cu); _set_slots(d.gu, gu)
        tc = c_int(0)
        _lib.check(lib.asrk_speller_plan(ctypes.byref(d), None, ctypes.byref(tc)), "speller_plan")
        tc = tc.value
        # [W_ih[:, E:] | W_hh]^T and Wq^T once per backward pass (the per-step GEMMs then stream rows)
        WT = torch.empty((XH, 4 * H), **f)
        transpose_into(W_ih[:, E:], In, 4 * H, Dv, WT, 4 * H)
        transpose_into(W_hh, H, 4 * H, H, WT[Dv:], 4 * H)
        WqT = torch.empty((NL * H, NA), **f)
        transpose_into(Wq, NL * H, NA, NL * H, WqT, NA)
        WmT = dctxh = None
        if NH > 1:
            Wm, ctxh_all = ctx.merge
            WmT = torch.empty((NH * Dv, Dv), **f)
            transpose_into(Wm, NH * Dv, Dv, NH * Dv, WmT, Dv)
            dctxh = torch.empty((L, B, NH * Dv), **f)
        WuT = []
        for l in range(nu):                  # [W_ih_l | W_hh_l]^T, rows 0..H-1 -> the layer below, H..2H-1 -> own past
            wt = torch.empty((2 * H, 4 * H), **f)
            transpose_into(Wu_ih[l], H, 4 * H, H, wt, 4 * H)
            transpose_into(Wu_hh[l], H, 4 * H, H, wt[H:], 4 * H)
            WuT.append(wt)
        dxu = [torch.empty((L, B, 2 * H), **f) for _ in range(nu)]
        dcu = [torch.empty((B, H), **f) for _ in range(nu)]
        dkey = ops.zeros((BN, Te, A), dev)
        dwe_part = dWp_part = dbe_part = dWc_part = None
        if not dot:
            dwe_part = ops.zeros((B * tc, A), dev)
            dWp_part = ops.zeros((B * tc, A * K), dev)
            dbe_part = ops.zeros((B * tc,), dev)
            dWc_part = ops.zeros((B, K * KW), dev)
        dxh = torch.empty((L, B, XH), **f)
        dq_pre = torch.empty((L, B, NA), **f)
        scratch = [torch.empty(s, **f) for s in ((BN, Te), (B, Te), (B, Te, max(K, 1)), (BN * tc, A), (B, H))]
        g = SpellerBwdT(_ptr(dstates), _ptr(datt), _ptr(WT), _ptr(WqT), _ptr(dkey), _ptr(dxh), _ptr(dq_pre),
                        _ptr(scratch[0]), _ptr(scratch[1]), _ptr(scratch[2]), _ptr(scratch[3]),
                        _ptr(dwe_part), _ptr(dWp_part), _ptr(dbe_part), _ptr(dWc_part), _ptr(scratch[4]), tc)
        _set_slots(g.WuT, WuT); _set_slots(g.dxu, dxu); _set_slots(g.dcu, dcu)
        g.WmT, g.dctxh = _ptr(WmT), _ptr(dctxh)
        _lib.check(lib.asrk_speller_bwd_f32(ctypes.byref(d), ctypes.byref(g), _stream()), "speller_bwd")
        dG = gates.view(L * B, 4 * H)            # now pre-activation gradients
        LB = L * B

        # ---- gradients that feed further back-propagation (encoder, embeddings): main stream
        dvalue = torch.empty((BN, Te, Dv), **f)
        # gradient of the per-row contexts: dxh[..., :Dv] (one head) or the merge_head backward's dctxh (rows b*N + n)
        dcr, dcr_step, dcr_ld = (dxh, B * XH, XH) if NH == 1 else (dctxh, BN * Dv, Dv)
        rc = lib.asrk_speller_dvalue_f32(_p(att_seq), L * Te, Te, _p(dcr), dcr_step, dcr_ld, _p(dvalue), BN, L, Te,
                                         Dv, _stream())
        if rc == -2:                              # very long encoder memories: one small GEMM per utterance
            att_rows = att_seq.view(BN, L, Te)
            for r in range(BN):
                src = dxh[:, r] if NH == 1 else dctxh.view(L, BN, Dv)[:, r]
                gemm(1, 0, Te, Dv, L, att_rows[r], Te, src, dcr_step, dvalue[r], Dv)
        else:
            _lib.check(rc, "speller_dvalue")
        demb = torch.empty((L, B, E), **f)
        gemm(0, 0, LB, E, 4 * H, dG, 4 * H, W_ih, In, demb, E)
        dsos = demb[0]
        dteacher = ops.zeros((B, Lt, E), dev)
        if L > 1:
            copy3d(demb[1:], dteacher, L - 1, B, E, B * E, E, E, Lt * E)

        # ---- weight gradients: nothing downstream reads them -> side stream when they are plain leaves
        def weight_grads():
            dW_ih = torch.empty((4 * H, In), **f)
            gemm(1, 0, 4 * H, E, LB, dG, 4 * H, emb_tm, E, dW_ih, In)
            gemm(1, 0, 4 * H, Dv, LB, dG, 4 * H, ctx_all, Dv, dW_ih[:, E:], In)
            dW_hh = torch.empty((4 * H, H), **f)
            gemm(1, 0, 4 * H, H, LB, dG, 4 * H, h, H, dW_hh, H)          # h[0:L] = states entering each step
            db = torch.empty((4 * H,), **f)
            colsum(dG, LB, 4 * H, 4 * H, db)
            dWq = torch.empty((NA, NL * H), **f)
            for l, hl in enumerate((h,) + tuple(hu)):           # column block l = the layer's states entering each step
                gemm(1, 0, NA, H, LB, dq_pre, NA, hl, H, dWq[:, l * H:], NL * H)
            dbq = torch.empty((NA,), **f)
            colsum(dq_pre, LB, NA, NA, dbq)
            if dot:
                dWc = dWp = dwe = dbe = None
            else:
                dWp = torch.empty((A * K,), **f)
                colsum(dWp_part, B * tc, A * K, A * K, dWp)
                dwe = torch.empty((A,), **f)
                colsum(dwe_part, B * tc, A, A, dwe)
                dbe = torch.empty((1,), **f)
                colsum(dbe_part, B * tc, 1, 1, dbe)
                dWc = torch.empty((K * KW,), **f)
                colsum(dWc_part, B, K * KW, K * KW, dWc)
                dWc, dWp, dwe = dWc.view(K, 1, KW), dWp.view(A, K), dwe.view(we.shape)
            dWm = dbm = None
            if NH > 1:                                           # merge_head: ctx = ctxh Wm^T + bm
                dWm = torch.empty((Dv, NH * Dv), **f)
                gemm(1, 0, Dv, NH * Dv, LB, dxh, XH, ctxh_all, NH * Dv, dWm, NH * Dv)
                dbm = torch.empty((Dv,), **f)
                colsum(dxh, LB, Dv, XH, dbm)
            out = [dWq, dbq, dWc, dWp, dwe, dbe, dW_ih, dW_hh, db, db.clone(), dWm, dbm]
            for l in range(nu):                                  # upper layers: input = the layer below's NEW state
                dGl = gu[l].view(LB, 4 * H)
                below = (h if l == 0 else hu[l - 1])[1:]
                dWi = torch.empty((4 * H, H), **f)
                gemm(1, 0, 4 * H, H, LB, dGl, 4 * H, below, H, dWi, H)
                dWh = torch.empty((4 * H, H), **f)
                gemm(1, 0, 4 * H, H, LB, dGl, 4 * H, hu[l], H, dWh, H)
                dbl = torch.empty((4 * H,), **f)
                colsum(dGl, LB, 4 * H, 4 * H, dbl)
                out += [dWi, dWh, dbl, dbl.clone()]
            return out

        # side stream only if what follows on the main stream (the top encoder layer's BPTT) leaves CUs free;
        # beside a plan that owns every CU the GEMMs would be parked, not overlapped (ops._defer_beside_bptt)
        if ops._can_defer(*ctx.weight_refs) and ops._defer_beside_bptt():
            used = tuple(t_ for t_ in (dG, emb_tm, ctx_all, h, dq_pre, dWp_part, dwe_part, dbe_part, dWc_part, dxh)
                         if t_ is not None) + tuple(hu) + tuple(gu) + ((ctx.merge[1],) if NH > 1 else ())
            with ops._SideStream(dev, used, background=False) as side:
                wg = weight_grads()
                side.keep(*[t_ for t_ in wg if t_ is not None])
        else:
            wg = weight_grads()
        # forward arguments: ..., W_ih, W_hh, b_ih, b_hh, L, temperature, cell, nhead, Wm, bm, *upper
        return (dkey, dvalue, None, dsos, dteacher, *wg[:10], None, None, None, None, wg[10], wg[11], *wg[12:])


class SpellerStepper:
    """One attention + decoder-cell step for `n` rows outside the training loop (greedy / beam decoding:
    src/decode.py:110-121, src/asr.py:136-142) through asrk_speller_step_f32: 4 kernels per step
    (query, location conv + energies, softmax + context, gate GEMM + LSTM cell).  `shared` = all rows
    attend over ONE utterance's key / value (the live hypotheses of a beam search) -> nothing is
    replicated per hypothesis."""

    def __init__(self, attention, decoder, key, value, lens, n, shared):
        _require_gpu(key)
        al = attention.att_layer
        self.key, self.value = _f32c(key), _f32c(value)
        self.lens = lens.to(device=key.device, dtype=torch.int64).contiguous()
        params = [p.detach() for p in decoder.layers.layer_params(0)]
        self.gru = not decoder.enable_cell
        if self.gru:                             # GRU decoder: the loop's four-rows-per-unit layout, c is not used
            params = stack_gru_params(*params)
        w_ih, w_hh, b_ih, b_hh = (_f32c(p) for p in params)
        self.w = [_f32c(t.detach()) for t in (attention.proj_q.weight, attention.proj_q.bias, al.loc_conv.weight,
                                              al.loc_proj.weight, al.gen_energy.weight, al.gen_energy.bias)]
        self.w += [w_ih, w_hh, b_ih, b_hh]
        _, Te, A = self.key.shape
        Dv, H = self.value.shape[2], w_hh.shape[1]
        E = w_ih.shape[1] - Dv
        K, ks = self.w[2].shape[0], (self.w[2].shape[2] - 1) // 2
        f = dict(dtype=torch.float32, device=key.device)
        self.n, self.Te, self.H, self.Dv = n, Te, H, Dv
        self.q = torch.empty((1, n, A), **f)
        self.conv = torch.empty((1, n, Te, K), **f)
        self.attn = torch.empty((n, 1, Te), **f)
        self.ctx = torch.empty((1, n, Dv), **f)
        self.h = torch.empty((2, n, H), **f)      # slot 0: state entering the step, slot 1: leaving it
        self.c = torch.empty((2, n, H), **f)
        self.e = torch.empty((n, Te), **f)
        self.d = SpellerT(n, Te, A, Dv, K, ks, H, E, 1, float(al.temperature), 1 if shared else 0,
                          _ptr(self.key), _ptr(self.value), _ptr(self.lens), *[_ptr(t) for t in self.w], None,
                          _ptr(self.q), _ptr(self.conv), _ptr(self.attn), Te, Te, _ptr(self.ctx), None,
                          _ptr(self.h), _ptr(self.c), None, _ptr(self.e), None)
        self.d.cell = 1 if self.gru else 0

    def step(self, emb, prev_att):
        """emb [n,E] embedded previous tokens, prev_att [n,1,Te] (contiguous); the entering state must be in
        self.h[0] / self.c[0].  Returns (attn [n,1,Te], ctx [n,Dv], h [n,H], c [n,H]) - views of the
        stepper's buffers, overwritten by the next call."""
        emb, prev = _f32c(emb), _f32c(prev_att)
        _lib.check(_L().asrk_speller_step_f32(ctypes.byref(self.d), 0, _p(prev), self.Te, _p(emb), _stream()),
                   "speller_step")
        return self.attn, self.ctx[0], self.h[1], self.c[1]


class MultiSpellerStepper:
    """SpellerStepper for the beams of SEVERAL utterances at once (BeamDecoder.forward_batch): key [U,Te,A] /
    value [U,Te,Dv] / lens [U] hold U utterances' encoder memories (zero-padded to the longest), and every batch row
    names the memory it attends over (`row_mem`, asrk_speller_t::row_mem).  The row count changes from step to step
    (beams grow, utterances finish), so buffers are sized for `capacity` rows and a step uses the first n."""

    def __init__(self, attention, decoder, key, value, lens, capacity, row_group=0):
        """row_group = RG > 1: rows [g*RG, (g+1)*RG) of every step share one memory (asrk_speller_t::row_group)"""
        _require_gpu(key)
        al = attention.att_layer
        self.key, self.value = _f32c(key), _f32c(value)
        self.lens = lens.to(device=key.device, dtype=torch.int64).contiguous()
        params = [p.detach() for p in decoder.layers.layer_params(0)]
        self.gru = not decoder.enable_cell
        if self.gru:                             # GRU decoder: four-rows-per-unit layout, the c slots are not used
            params = stack_gru_params(*params)
        w_ih, w_hh, b_ih, b_hh = (_f32c(p) for p in params)
        self.w = [_f32c(t.detach()) for t in (attention.proj_q.weight, attention.proj_q.bias, al.loc_conv.weight,
                                              al.loc_proj.weight, al.gen_energy.weight, al.gen_energy.bias)]
        self.w += [w_ih, w_hh, b_ih, b_hh]
        _, Te, A = self.key.shape
        Dv, H = self.value.shape[2], w_hh.shape[1]
        E = w_ih.shape[1] - Dv
        K, ks = self.w[2].shape[0], (self.w[2].shape[2] - 1) // 2
        f = dict(dtype=torch.float32, device=key.device)
        n = capacity
        self.cap, self.Te, self.H, self.Dv = n, Te, H, Dv
        self.q = torch.empty((n, A), **f)
        self.conv = torch.empty((n, Te, K), **f)
        self.ctx = torch.empty((n, Dv), **f)
        self.e = torch.empty((n, Te), **f)
        self.d = SpellerT(n, Te, A, Dv, K, ks, H, E, 1, float(al.temperature), 0,
                          _ptr(self.key), _ptr(self.value), _ptr(self.lens), *[_ptr(t) for t in self.w], None,
                          _ptr(self.q), _ptr(self.conv), None, Te, Te, _ptr(self.ctx), None,
                          None, None, None, _ptr(self.e), None, None)
        self.d.cell = 1 if self.gru else 0
        self.d.row_group = int(row_group)

    def step(self, row_mem, emb, prev_att, h_in, c_in, parent=None, state=None):
        """row_mem [n] int32 (device), emb [n,E], prev_att [n,1,Te], h_in / c_in [n,H] = the entering decoder state
        (parent [n] int64 given: rows parent[i] of h_in / c_in - the gather lands directly in the state slot;
        state = (hbuf, cbuf) [2,n,H] given: slot 0 already holds the entering state, h_in / c_in are not read).
        Returns (attn [n,1,Te], ctx [n,Dv], h [n,H], c [n,H]); attn / h / c are fresh tensors, ctx is a view of the
        stepper's buffer (consumed inside the step)."""
        n = int(row_mem.shape[0])
        if n > self.cap:
            raise _lib.AsrkError("MultiSpellerStepper: %d rows exceed the capacity %d" % (n, self.cap))
        f = dict(dtype=torch.float32, device=self.key.device)
        if state is not None:
            hbuf, cbuf = state
        else:
            hbuf = torch.empty((2, n, self.H), **f)  # slot 0: entering state, slot 1: leaving it (asrk_speller_t::h)
            cbuf = torch.empty((2, n, self.H), **f)
        if state is not None:
            pass
        elif parent is None:
            hbuf[0].copy_(h_in)
            cbuf[0].copy_(c_in)
        else:
            torch.index_select(h_in, 0, parent, out=hbuf[0])
            torch.index_select(c_in, 0, parent, out=cbuf[0])
        attn = torch.empty((n, 1, self.Te), **f)
        emb, prev = _f32c(emb), _f32c(prev_att)
        rm = row_mem.to(torch.int32).contiguous()
        d = self.d
        d.B, d.row_mem, d.attn, d.h, d.c = n, _ptr(rm), _ptr(attn), _ptr(hbuf), _ptr(cbuf)
        from . import decoder_ops as dops
        E = emb.shape[1]
        if not self.gru and n >= dops.LSTM_CELL_GEMM_ROWS and E % 32 == 0 and self.Dv % 32 == 0 and self.H % 32 == 0:
            # many rows: the attention half in the fused step, the cell as three bf16x6 panel GEMMs (embedding, context,
            # state) against weight panels split once per decode + the pointwise cell kernel - the fused cell is a
            # 64-row tile that re-streams W_ih | W_hh per tile (8 x 30 us at 512 rows)
            from .ops import SplitPanel, gemm_panels, tanh_
            H, Dv = self.H, self.Dv
            h_panel = SplitPanel(hbuf[0], H, n, H, False)             # the entering state: query AND recurrent projection
            A = self.q.shape[1]
            gemm_panels(n, A, H, h_panel, 0, 0, dops.weight_panel(self.w[0]), 0, 0, self.q, A, bias=self.w[1])
            tanh_(self.q[:n])
            wq, d.Wq = d.Wq, None                                    # the query is in place: attention half only
            try:
                _lib.check(_L().asrk_speller_step_f32(ctypes.byref(d), 0, _p(prev), self.Te, None, _stream()),
                           "speller_step(multi, attention)")
            finally:
                d.Wq = wq
            w_ih, w_hh, b_ih, b_hh = self.w[6:10]
            gates = torch.empty((n, 4 * H), **f)
            wp_ih, wp_hh = dops.weight_panel(w_ih), dops.weight_panel(w_hh)
            gemm_panels(n, 4 * H, E, SplitPanel(emb, E, n, E, False), 0, 0, wp_ih, 0, 0, gates, 4 * H, bias=b_ih,
                        bias2=b_hh)
            ctx = self.ctx[:n]
            gemm_panels(n, 4 * H, Dv, SplitPanel(ctx, Dv, n, Dv, False), 0, 0, wp_ih, 0, E, gates, 4 * H, beta=1.0)
            gemm_panels(n, 4 * H, H, h_panel, 0, 0, wp_hh, 0, 0, gates, 4 * H, beta=1.0)
            _lib.check(_L().asrk_lstm_cell_fwd_f32(_p(gates), _p(cbuf[0]), _p(cbuf[1]), _p(hbuf[1]), n, H, _stream()),
                       "lstm_cell")
            return attn, ctx, hbuf[1], cbuf[1]
        _lib.check(_L().asrk_speller_step_f32(ctypes.byref(d), 0, _p(prev), self.Te, _p(emb), _stream()),
                   "speller_step(multi)")
        return attn, self.ctx[:n], hbuf[1], cbuf[1]


def lstm_cell_fused(x, h, c, w_ih, w_hh, b_ih, b_hh, out=None):
    """nn.LSTM step on a length-1 sequence in one kernel -> (h', c'); out = (h', c') buffers to fill (contiguous [B,H])"""
    _require_gpu(x)
    xc, hc, cc = _f32c(x), _f32c(h), _f32c(c)
    B, In = xc.shape
    H = hc.shape[1]
    if out is None:
        h_new = torch.empty((B, H), dtype=torch.float32, device=x.device)
        c_new = torch.empty((B, H), dtype=torch.float32, device=x.device)
    else:
        h_new, c_new = out
        assert h_new.is_contiguous() and c_new.is_contiguous() and h_new.shape == (B, H) and c_new.shape == (B, H)
    _lib.check(_L().asrk_lstm_cell_fused_f32(_p(xc), In, In, _p(hc), _p(cc), _p(_f32c(w_ih)), _p(_f32c(w_hh)),
                                             _p(_f32c(b_ih)), _p(_f32c(b_hh)), _p(h_new), _p(c_new), B, H,
                                             _stream()), "lstm_cell_fused")
    return h_new, c_new
