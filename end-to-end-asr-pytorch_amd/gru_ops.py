"""GRU layers / cells (the `module: 'GRU'` option of src/module.py:112-113, src/asr.py:175-176 and
src/lm.py:20-21 — torch.nn.GRU, gate order r, z, n) over the gfx950 kernels: input projections are
one MFMA GEMM for the whole sequence, the recurrent projection of each step is a skinny
weight-streaming GEMM, the gate math is csrc/gru.hip.

A whole encoder / LM layer runs in the persistent recurrence kernels of csrc/lstm_rec.hip in their GRU
mode (GRURecLayerFn: one launch per direction pair and pass, W_hh resident in LDS, hand-off between
workgroups through the sentinel ring - the same machinery as the LSTM layers).  Hidden sizes that are not
a multiple of 4 fall back to GRULayerFn, a host loop over the time steps (2 launches per step)."""
import os as _os

import torch
from torch.autograd import Function

from . import _lib
from . import ops as _ops
from .ops import _L, _p, _stream, _f32c, _require_gpu, gemm, colsum, copy3d


def _cell_fwd(L, gi, gh, ldi, ldh, h_prev, ldp, h_new, ldn, B, H):
    _lib.check(L.asrk_gru_cell_fwd_f32(_p(gi), _p(gh), ldi, ldh, _p(h_prev), ldp, _p(h_new), ldn, B, H,
                                       _stream()), "gru_cell")


def _cell_bwd(L, gi, gh, ldi, ldh, h_prev, ldp, dh, ldd, dh2, dh_prev, ldo, B, H):
    _lib.check(L.asrk_gru_cell_bwd_f32(_p(gi), _p(gh), ldi, ldh, _p(h_prev), ldp, _p(dh), ldd, _p(dh2),
                                       _p(dh_prev), ldo, B, H, _stream()), "gru_cell_bwd")


class GRULayerFn(Function):
    """One (bi)directional GRU layer on a time-major sequence [T,B,Din] -> [T,B,ndir*H], zero initial
    state over the full padded length (what nn.GRU(batch_first=True) computes at src/module.py:131)."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        _require_gpu(x)
        L = _L()
        xc = _f32c(x)
        T, B, Din = xc.shape
        H = w_hh_f.shape[1]
        ndir = 2 if w_ih_r is not None else 1
        dev = x.device
        M = T * B
        ldg, ldy = ndir * 3 * H, ndir * H
        Gi = torch.empty((M, ldg), dtype=torch.float32, device=dev)
        Gh = torch.empty((M, ldg), dtype=torch.float32, device=dev)
        Y = torch.empty((M, ldy), dtype=torch.float32, device=dev)
        h0 = torch.zeros((B, H), dtype=torch.float32, device=dev)
        ws = [(_f32c(w_ih_f), _f32c(w_hh_f), b_ih_f, b_hh_f)]
        if ndir == 2:
            ws.append((_f32c(w_ih_r), _f32c(w_hh_r), b_ih_r, b_hh_r))
        for d, (w_ih, w_hh, b_ih, b_hh) in enumerate(ws):
            gemm(0, 1, M, 3 * H, Din, xc, Din, w_ih, Din, Gi[:, d * 3 * H:], ldg, bias=b_ih)
            for step in range(T):
                t = step if d == 0 else T - 1 - step
                tp = t - 1 if d == 0 else t + 1
                hp = h0 if step == 0 else Y[tp * B:, d * H:]
                ldp = H if step == 0 else ldy
                gh_t = Gh[t * B:, d * 3 * H:]
                gemm(0, 1, B, 3 * H, H, hp, ldp, w_hh, H, gh_t, ldg, bias=b_hh)
                _cell_fwd(L, Gi[t * B:, d * 3 * H:], gh_t, ldg, ldg, hp, ldp, Y[t * B:, d * H:], ldy, B, H)
        ctx.dims = (T, B, Din, H, ndir)
        ctx.has_bias = b_ih_f is not None
        ctx.save_for_backward(xc, ws[0][0], ws[0][1], ws[1][0] if ndir == 2 else None,
                              ws[1][1] if ndir == 2 else None, Gi, Gh, Y)
        ctx.consumed = False
        return Y.view(T, B, ldy)

    @staticmethod
    def backward(ctx, dY):
        L = _L()
        xc, w_ih_f, w_hh_f, w_ih_r, w_hh_r, Gi, Gh, Y = ctx.saved_tensors
        if ctx.consumed:
            raise RuntimeError("GRULayerFn: backward twice (the gate buffers are reused in place)")
        ctx.consumed = True
        T, B, Din, H, ndir = ctx.dims
        dev = dY.device
        M = T * B
        ldg, ldy = ndir * 3 * H, ndir * H
        dYc = _f32c(dY).reshape(M, ldy)
        f32 = dict(dtype=torch.float32, device=dev)
        ws = [(w_ih_f, w_hh_f)] + ([(w_ih_r, w_hh_r)] if ndir == 2 else [])
        h0 = torch.zeros((B, H), **f32)
        for d, (w_ih, w_hh) in enumerate(ws):
            carry = None
            bufs = [torch.empty((B, H), **f32), torch.empty((B, H), **f32)]
            for step in range(T - 1, -1, -1):
                t = step if d == 0 else T - 1 - step
                tp = t - 1 if d == 0 else t + 1
                hp = h0 if step == 0 else Y[tp * B:, d * H:]
                ldp = H if step == 0 else ldy
                out = bufs[step & 1]
                gi_t, gh_t = Gi[t * B:, d * 3 * H:], Gh[t * B:, d * 3 * H:]
                _cell_bwd(L, gi_t, gh_t, ldg, ldg, hp, ldp, dYc[t * B:, d * H:], ldy, carry, out, H, B, H)
                if step > 0:      # dL/dh_{prev} = dh * z + dgh W_hh
                    gemm(0, 0, B, H, 3 * H, gh_t, ldg, w_hh, H, out, H, beta=1.0)
                carry = out
        dGi, dGh = Gi, Gh
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, Din), **f32)
            for d, (w_ih, _) in enumerate(ws):
                gemm(0, 0, M, Din, 3 * H, dGi[:, d * 3 * H:], ldg, w_ih, Din, dx, Din, beta=1.0 if d else 0.0)
            dx = dx.view(T, B, Din)
        grads = []
        for d in range(ndir):
            gi_d, gh_d = dGi[:, d * 3 * H:], dGh[:, d * 3 * H:]
            dw_ih = torch.empty((3 * H, Din), **f32)
            gemm(1, 0, 3 * H, Din, M, gi_d, ldg, xc, Din, dw_ih, Din)
            dw_hh = torch.zeros((3 * H, H), **f32)
            if T > 1:
                Mh = (T - 1) * B
                if d == 0:      # h_{t-1} = Y[t-1]
                    gemm(1, 0, 3 * H, H, Mh, gh_d[B:], ldg, Y, ldy, dw_hh, H)
                else:           # reverse direction: previous state of t is Y[t+1]
                    gemm(1, 0, 3 * H, H, Mh, gh_d, ldg, Y[B:, H:], ldy, dw_hh, H)
            db_ih = db_hh = None
            if ctx.has_bias:
                db_ih, db_hh = torch.empty((3 * H,), **f32), torch.empty((3 * H,), **f32)
                colsum(gi_d, M, 3 * H, ldg, db_ih)
                colsum(gh_d, M, 3 * H, ldg, db_hh)
            grads.append((dw_ih, dw_hh, db_ih, db_hh))
        if ndir == 1:
            grads.append((None, None, None, None))
        return (dx,) + grads[0] + grads[1]


class GRURecLayerFn(Function):
    """Same contract as GRULayerFn, through the persistent kernels (asrk_gru_rec_fwd_f32 / _bwd_f32).

    Gate buffer G [T*B, ndir*4H], per direction four H-wide blocks:
        forward in : x W_ir^T + b_ir + b_hr | x W_iz^T + b_iz + b_hz | x W_in^T + b_in | b_hn (broadcast)
        forward out: r | z | n | W_hn h + b_hn
        BPTT out   : dr | dz | dn | dn*r   (input-side gradients = blocks 0..2, hidden-side = 0, 1, 3)
    so no second [T*B, 3H] buffer exists and every weight / bias gradient is a GEMM or a slice of the
    kernel's own column sums."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, pyr_rate=1,
                pyr_style=None):
        _require_gpu(x)
        L = _L()
        xc = _f32c(x)
        T, B, Din = xc.shape
        H = w_hh_f.shape[1]
        ndir = 2 if w_ih_r is not None else 1
        dev = x.device
        M = T * B
        ldg, ldy = ndir * 4 * H, ndir * H
        G = torch.empty((M, ldg), dtype=torch.float32, device=dev)
        ws = [(_f32c(w_ih_f), _f32c(w_hh_f), b_ih_f, b_hh_f)]
        if ndir == 2:
            ws.append((_f32c(w_ih_r), _f32c(w_hh_r), b_ih_r, b_hh_r))
        for d, (w_ih, w_hh, b_ih, b_hh) in enumerate(ws):
            Gd = G[:, d * 4 * H:]
            if b_ih is not None:
                bh = b_hh.detach()
                b2 = torch.cat((bh[:2 * H], bh.new_zeros(H)))
                gemm(0, 1, M, 3 * H, Din, xc, Din, w_ih, Din, Gd, ldg, bias=b_ih, bias2=b2)
                copy3d(bh[2 * H:].contiguous(), Gd[:, 3 * H:], 1, M, H, 0, 0, 0, ldg)   # b_hn to every row
            else:
                gemm(0, 1, M, 3 * H, Din, xc, Din, w_ih, Din, Gd, ldg)
                G.view(M, ndir, 4, H)[:, d, 3].zero_()
        Y = torch.empty((M, ldy), dtype=torch.float32, device=dev)
        wsp = _ops.lstm_workspace(dev)
        xc_ = _ops._Exchange(L, T, B, H, ndir, 0, dev)
        mode = {None: 0, 'concat': 1, 'drop': 2}[pyr_style if pyr_rate > 1 else None]
        Y2 = None
        if mode == 1:
            Y2 = torch.empty((T // pyr_rate, B, pyr_rate * ldy), dtype=torch.float32, device=dev)
        elif mode == 2:
            Y2 = torch.empty(((T + pyr_rate - 1) // pyr_rate, B, ldy), dtype=torch.float32, device=dev)
        _lib.check(L.asrk_gru_rec_fwd_f32(_p(G), _p(ws[0][1]), _p(ws[1][1] if ndir == 2 else None), _p(Y), T, B,
                                          H, ndir, _p(xc_.buf), xc_.prefilled, _p(wsp), _p(Y2), mode,
                                          max(1, pyr_rate), xc_.flags, _stream()), "gru_rec_fwd")
        xc_.done()
        ctx.pyr = (mode, max(1, pyr_rate))
        ctx.dims = (T, B, Din, H, ndir)
        ctx.has_bias = b_ih_f is not None
        ctx.bias_refs = (b_ih_f, b_hh_f, b_ih_r, b_hh_r)
        ctx.save_for_backward(xc, ws[0][0], ws[0][1], ws[1][0] if ndir == 2 else None,
                              ws[1][1] if ndir == 2 else None, G, Y)
        ctx.consumed = False
        return Y2 if mode else Y.view(T, B, ldy)

    @staticmethod
    def backward(ctx, dY):
        L = _L()
        xc, w_ih_f, w_hh_f, w_ih_r, w_hh_r, G, Y = ctx.saved_tensors
        if ctx.consumed:
            raise RuntimeError("GRURecLayerFn: backward twice (the gate buffer is reused in place)")
        ctx.consumed = True
        T, B, Din, H, ndir = ctx.dims
        dev = dY.device
        M = T * B
        ldg, ldy = ndir * 4 * H, ndir * H
        mode, rate = ctx.pyr
        dYc = _f32c(dY)
        wsp = _ops.lstm_workspace(dev)
        _ops._note_bptt_plan(L, T, B, H, ndir)
        xc_ = _ops._Exchange(L, T, B, H, ndir, 1, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        db_all = torch.empty((ndir, 4 * H), **f32) if ctx.has_bias else None
        db_in_kernel = ctx.has_bias and B <= 32          # <= 2 batch groups: order-independent atomics (ops.py)
        _lib.check(L.asrk_gru_rec_bwd_f32(_p(G), _p(w_hh_f), _p(w_hh_r), _p(Y), _p(dYc), T, B, H, ndir,
                                          _p(xc_.buf), xc_.prefilled, _p(wsp), _p(db_all if db_in_kernel else None),
                                          mode, rate, xc_.flags, _stream()), "gru_rec_bwd")
        xc_.done()
        if ctx.has_bias and not db_in_kernel:
            _ops.colsum(G, M, ndir * 4 * H, ndir * 4 * H, db_all)
        _ops._gemm_phase_begins()
        dG = G
        ws = [w_ih_f] + ([w_ih_r] if ndir == 2 else [])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, Din), **f32)
            for d, w_ih in enumerate(ws):
                gemm(0, 0, M, Din, 3 * H, dG[:, d * 4 * H:], ldg, w_ih, Din, dx, Din, beta=1.0 if d else 0.0)
            dx = dx.view(T, B, Din)

        def param_grads(d):
            dGd = dG[:, d * 4 * H:]
            dw_ih = torch.empty((3 * H, Din), **f32)
            gemm(1, 0, 3 * H, Din, M, dGd, ldg, xc, Din, dw_ih, Din)
            dw_hh = torch.zeros((3 * H, H), **f32)
            if T > 1:
                Mh = (T - 1) * B
                # h_{t-1} = Y[t-1] (forward) / Y[t+1] (reverse); hidden-side gradients: blocks 0, 1 and 3
                g_rows, y_rows = (dGd[B:], Y) if d == 0 else (dGd, Y[B:, H:])
                gemm(1, 0, 2 * H, H, Mh, g_rows, ldg, y_rows, ldy, dw_hh, H)
                gemm(1, 0, H, H, Mh, g_rows[:, 3 * H:], ldg, y_rows, ldy, dw_hh[2 * H:], H)
            db_ih = db_hh = None
            if ctx.has_bias:
                db_ih = db_all[d, :3 * H].clone()
                db_hh = torch.cat((db_all[d, :2 * H], db_all[d, 3 * H:]))
            return dw_ih, dw_hh, db_ih, db_hh

        if _ops._can_defer(w_ih_f, w_hh_f, w_ih_r, w_hh_r, *ctx.bias_refs):
            with _ops._SideStream(dev, (dG, xc, Y, db_all)) as side:   # off the next layer's critical path
                grads = [param_grads(d) for d in range(ndir)]
                side.keep(*[t for g in grads for t in g])
        else:
            grads = [param_grads(d) for d in range(ndir)]
        if ndir == 1:
            grads.append((None, None, None, None))
        return (dx,) + grads[0] + grads[1] + (None, None)


def persistent_ok(H):
    return H % 4 == 0 and _os.environ.get("ASRK_GRU_PERSISTENT", "1") != "0"


def gru_layer(x_tm, params_f, params_r=None, pyramid=None):
    """params_* = (w_ih, w_hh, b_ih, b_hh) in torch nn.GRU layout.  pyramid = (rate, style): also apply
    the time reduction of src/module.py:141-153 (fused into the persistent kernel's output store)."""
    pr = params_r if params_r is not None else (None, None, None, None)
    H = params_f[1].shape[1]
    fuse = pyramid is not None and pyramid[0] > 1
    if persistent_ok(H):
        if fuse and _os.environ.get("ASRK_FUSE_PYRAMID", "1") != "0":
            return GRURecLayerFn.apply(x_tm, *params_f, *pr, pyramid[0], pyramid[1])
        y = GRURecLayerFn.apply(x_tm, *params_f, *pr)
    else:
        y = GRULayerFn.apply(x_tm, *params_f, *pr)
    return _ops.PyramidFn.apply(y, pyramid[0], pyramid[1]) if fuse else y


class GRUCellFn(Function):
    """One step of an nn.GRU layer on a length-1 sequence (decoder, src/asr.py:218; RNN-LM)."""

    @staticmethod
    def forward(ctx, x, h, w_ih, w_hh, b_ih, b_hh):
        _require_gpu(x)
        L = _L()
        xc, hc, wi, wh = _f32c(x), _f32c(h), _f32c(w_ih), _f32c(w_hh)
        B, In = xc.shape
        H = wh.shape[1]
        dev = x.device
        gi = torch.empty((B, 3 * H), dtype=torch.float32, device=dev)
        gh = torch.empty((B, 3 * H), dtype=torch.float32, device=dev)
        gemm(0, 1, B, 3 * H, In, xc, In, wi, In, gi, 3 * H, bias=b_ih)
        gemm(0, 1, B, 3 * H, H, hc, H, wh, H, gh, 3 * H, bias=b_hh)
        h_new = torch.empty((B, H), dtype=torch.float32, device=dev)
        _cell_fwd(L, gi, gh, 3 * H, 3 * H, hc, H, h_new, H, B, H)
        ctx.save_for_backward(xc, hc, wi, wh, gi, gh)
        ctx.has_bias = b_ih is not None
        return h_new

    @staticmethod
    def backward(ctx, dh):
        L = _L()
        xc, hc, wi, wh, gi, gh = ctx.saved_tensors
        B, In = xc.shape
        H = wh.shape[1]
        dev = dh.device
        f32 = dict(dtype=torch.float32, device=dev)
        dgi, dgh = gi.clone(), gh.clone()
        dh_prev = torch.empty((B, H), **f32)
        _cell_bwd(L, dgi, dgh, 3 * H, 3 * H, hc, H, _f32c(dh), H, None, dh_prev, H, B, H)
        gemm(0, 0, B, H, 3 * H, dgh, 3 * H, wh, H, dh_prev, H, beta=1.0)
        dx = torch.empty((B, In), **f32)
        gemm(0, 0, B, In, 3 * H, dgi, 3 * H, wi, In, dx, In)
        dw_ih = torch.empty((3 * H, In), **f32)
        gemm(1, 0, 3 * H, In, B, dgi, 3 * H, xc, In, dw_ih, In)
        dw_hh = torch.empty((3 * H, H), **f32)
        gemm(1, 0, 3 * H, H, B, dgh, 3 * H, hc, H, dw_hh, H)
        db_ih = db_hh = None
        if ctx.has_bias:
            db_ih, db_hh = torch.empty((3 * H,), **f32), torch.empty((3 * H,), **f32)
            colsum(dgi, B, 3 * H, 3 * H, db_ih)
            colsum(dgh, B, 3 * H, 3 * H, db_hh)
        return dx, dh_prev, dw_ih, dw_hh, db_ih, db_hh


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    return GRUCellFn.apply(x, h, w_ih, w_hh, b_ih, b_hh)


@torch.no_grad()
def gru_cell_infer(x, h, w_ih, w_hh, b_ih, b_hh):
    return GRUCellFn.apply(x, h, w_ih, w_hh, b_ih, b_hh)
