"""GRU layers / cells (the `module: 'GRU'` option of src/module.py:112-113, src/asr.py:175-176 and
src/lm.py:20-21 — torch.nn.GRU, gate order r, z, n) over the gfx950 kernels: input projections are
one MFMA GEMM for the whole sequence, the recurrent projection of each step is a skinny
weight-streaming GEMM, the gate math is csrc/gru.hip.

There is no persistent GRU recurrence kernel yet: an encoder GRU layer is a host loop over the time
steps (2 launches per step and direction).  It is functional and parity-tested, not tuned — the
LSTM path is the one the benchmark configurations use."""
import torch
from torch.autograd import Function

from . import _lib
from .ops import _L, _p, _stream, _f32c, _require_gpu, gemm, colsum


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


def gru_layer(x_tm, params_f, params_r=None):
    """params_* = (w_ih, w_hh, b_ih, b_hh) in torch nn.GRU layout."""
    pr = params_r if params_r is not None else (None, None, None, None)
    return GRULayerFn.apply(x_tm, *params_f, *pr)


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
