"""Autograd operators for the convolutional prenets (reference: src/module.py:7-90) over the
channels-last kernels of csrc/conv.hip and csrc/conv3x3.hip.  The VGG prenet's 3x3 layers with 64 / 128 input
channels are implicit GEMMs (no patch matrix; forward, data gradient and weight gradient kernels, ReLU and its
backward fused); every other convolution = im2col gather + one MFMA GEMM against the reference-layout weight
(viewed [Cout, Cin*KH*KW]); 2x2 max pooling with stored arg-max.  No ATen math; the GEMM is asrk_gemm_f32."""
import os
import torch
from torch.autograd import Function

from . import _lib
from .ops import _L, _p, _stream, _f32c, _require_gpu, gemm, colsum, copy3d, zeros


class Geom:
    """Convolution geometry over an input addressed as x[b*sb + h*sh + w*sw + c*sc]."""

    def __init__(self, B, H, W, C, KH, KW, SH, SW, PH, PW, sb, sh, sw, sc):
        self.B, self.H, self.W, self.C = B, H, W, C
        self.KH, self.KW, self.SH, self.SW, self.PH, self.PW = KH, KW, SH, SW, PH, PW
        self.sb, self.sh, self.sw, self.sc = sb, sh, sw, sc
        L = _L()
        self.Ho = int(L.asrk_conv_out_size(H, KH, SH, PH))
        self.Wo = int(L.asrk_conv_out_size(W, KW, SW, PW))
        if self.Ho <= 0 or self.Wo <= 0:
            raise RuntimeError("convolution input {}x{} is smaller than the kernel {}x{} (pad {}x{})".format(
                H, W, KH, KW, PH, PW))
        self.M = B * self.Ho * self.Wo
        self.K = C * KH * KW

    def args(self):
        return (self.B, self.H, self.W, self.C, self.KH, self.KW, self.SH, self.SW, self.PH, self.PW,
                self.sb, self.sh, self.sw, self.sc)

    def channels_last_ok(self, *tensors):
        """the (kh, kw, cin) K order of asrk_im2col_cl_f32: channels contiguous, everything 16-byte steppable"""
        import os
        return (os.environ.get("ASRK_CONV_CL", "1") != "0" and self.sc == 1 and self.C % 4 == 0 and self.sb % 4 == 0
                and self.sh % 4 == 0 and self.sw % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in tensors)
                and self.M * (self.C // 4) < 2 ** 31 and self.B * self.H * self.W * (self.C // 4) < 2 ** 31)

    def direct3x3_ok(self, Cout, *tensors):
        """contiguous channels-last input of a 3x3 / stride 1 / pad 1 layer that csrc/conv3x3.hip has kernels for"""
        return (os.environ.get("ASRK_CONV_DIRECT", "1") != "0"
                and (self.KH, self.KW, self.SH, self.SW, self.PH, self.PW) == (3, 3, 1, 1, 1, 1)
                and (self.sc, self.sw, self.sh, self.sb) == (1, self.C, self.W * self.C, self.H * self.W * self.C)
                and all(t.data_ptr() % 16 == 0 for t in tensors)
                and bool(_L().asrk_conv3x3_supported(self.H, self.W, self.C, Cout)))

    def first3x3_ok(self, Cout, *tensors):
        """the 1-3 plane first layer of the VGG prenet, read in place through element strides"""
        return (os.environ.get("ASRK_CONV_DIRECT", "1") != "0"
                and (self.KH, self.KW, self.SH, self.SW, self.PH, self.PW) == (3, 3, 1, 1, 1, 1)
                and all(t.data_ptr() % 16 == 0 for t in tensors)
                and bool(_L().asrk_conv3x3_first_supported(self.H, self.W, self.C, Cout)))

    def check_extent(self, t):
        last = (self.B - 1) * self.sb + (self.H - 1) * self.sh + (self.W - 1) * self.sw + (self.C - 1) * self.sc
        if self.B > 0 and last >= t.numel():
            raise _lib.AsrkError("conv geometry addresses element {} of a {}-element tensor".format(last, t.numel()))


class ConvFn(Function):
    """x (any shape, addressed through `geom`) * weight [Cout, Cin, KH(, KW)] + bias -> [M, Cout]
    with rows ordered (b, ho, wo): the channels-last activation.  `relu` fuses max(., 0)."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, relu):
        _require_gpu(x)
        L = _L()
        xc = _f32c(x)
        geom.check_extent(xc)
        w = _f32c(weight)
        Cout = w.shape[0]
        if w.numel() != Cout * geom.K or bias.numel() != Cout:
            raise RuntimeError("conv: weight {} / bias {} do not match Cin*KH*KW = {}".format(
                tuple(weight.shape), tuple(bias.shape), geom.K))
        if geom.direct3x3_ok(Cout, xc, w):
            wf = torch.empty_like(w)
            _lib.check(L.asrk_conv3x3_weight_f32(_p(w), _p(wf), Cout, geom.C, 0, _stream()), "conv3x3_weight")
            y = torch.empty((geom.M, Cout), dtype=torch.float32, device=x.device)
            _lib.check(L.asrk_conv3x3_f32(_p(xc), None, _p(wf), _p(_f32c(bias)), _p(y), geom.B, geom.H, geom.W, geom.C,
                                          Cout, int(bool(relu)), _stream()), "conv3x3")
            ctx.save_for_backward(xc, w, y if relu else None)
            ctx.geom, ctx.relu, ctx.x_shape, ctx.w_shape, ctx.cl = geom, relu, tuple(x.shape), tuple(weight.shape), None
            return y
        if geom.first3x3_ok(Cout, w):
            y = torch.empty((geom.M, Cout), dtype=torch.float32, device=x.device)
            _lib.check(L.asrk_conv3x3_first_f32(_p(xc), _p(w), _p(_f32c(bias)), _p(y), geom.B, geom.H, geom.W, geom.C, Cout,
                                                geom.sb, geom.sh, geom.sw, geom.sc, int(bool(relu)), _stream()),
                       "conv3x3_first")
            ctx.save_for_backward(xc, w, y if relu else None)
            ctx.geom, ctx.relu, ctx.x_shape, ctx.w_shape, ctx.cl = geom, relu, tuple(x.shape), tuple(weight.shape), 'first'
            return y
        # K padded to a multiple of 4 (the first VGG layer: 9 .. 27): zero patch columns against zero weight columns, so
        # that all three GEMMs take 16-byte paths (the unpadded K = 27 weight gradient ran on the scalar kernel: 0.77 ms)
        Kp = (geom.K + 3) // 4 * 4
        col = torch.empty((geom.M, Kp), dtype=torch.float32, device=x.device)
        # channels-contiguous inputs (every layer but the one that reads the [B,T,C*F] feature tensor in place): patches
        # in (kh, kw, cin) order - whole 16-byte pieces, contiguous runs of C floats - against the weight in that order
        cl = geom.channels_last_ok(xc, col)
        if cl:
            wk = torch.empty_like(w.view(Cout, geom.K))
            _lib.check(L.asrk_conv_weight_reorder_f32(_p(w), _p(wk), Cout, geom.C, geom.KH * geom.KW, 0, _stream()),
                       "conv_weight_reorder")
            _lib.check(L.asrk_im2col_cl_f32(_p(xc), _p(col), *geom.args(), _stream()), "im2col_cl")
        else:
            if Kp != geom.K:
                wk = zeros((Cout, Kp), x.device)
                copy3d(w, wk, 1, Cout, geom.K, 0, geom.K, 0, Kp)
            else:
                wk = w
            _lib.check(L.asrk_im2col_ld_f32(_p(xc), _p(col), Kp, *geom.args(), _stream()), "im2col")
        y = torch.empty((geom.M, Cout), dtype=torch.float32, device=x.device)
        gemm(0, 1, geom.M, Cout, Kp, col, Kp, wk, Kp, y, Cout, bias=_f32c(bias))
        if relu:
            _lib.check(L.asrk_relu_fwd_f32(_p(y), y.numel(), _stream()), "relu")
        ctx.save_for_backward(col, wk, y if relu else None)
        ctx.geom, ctx.relu, ctx.x_shape, ctx.w_shape, ctx.cl = geom, relu, tuple(x.shape), tuple(weight.shape), cl
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        col, w, y = ctx.saved_tensors
        g = ctx.geom
        Cout = w.shape[0]
        dyc = _f32c(dy)
        if ctx.cl is None:                            # implicit-GEMM layer: `col` is the input itself
            xc, dx, dw, db = col, None, None, None
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                dw = torch.empty(ctx.w_shape, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[1] else None
                db = torch.empty((Cout,), dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[2] else None
                nws = int(L.asrk_conv3x3_wgrad_ws_bytes(g.B, g.H, g.W, g.C, Cout))
                ws = torch.empty((nws,), dtype=torch.uint8, device=dy.device)
                _lib.check(L.asrk_conv3x3_wgrad_f32(_p(xc), _p(dyc), _p(y), _p(dw), _p(db), g.B, g.H, g.W, g.C, Cout,
                                                    _p(ws), nws, _stream()), "conv3x3_wgrad")
            if ctx.needs_input_grad[0]:
                wt = torch.empty_like(w)
                _lib.check(L.asrk_conv3x3_weight_f32(_p(w), _p(wt), Cout, g.C, 1, _stream()), "conv3x3_weight")
                dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dy.device)
                _lib.check(L.asrk_conv3x3_f32(_p(dyc), _p(y), _p(wt), None, _p(dx), g.B, g.H, g.W, Cout, g.C, 0,
                                              _stream()), "conv3x3_dgrad")
            return dx, dw, db, None, None
        if ctx.cl == 'first':                         # few-plane first layer: `col` is the feature tensor itself
            xc, dx, dw, db = col, None, None, None
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                dw = torch.empty(ctx.w_shape, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[1] else None
                db = torch.empty((Cout,), dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[2] else None
                nws = int(L.asrk_conv3x3_first_wgrad_ws_bytes(g.B, g.H, g.W, g.C, Cout))
                ws = torch.empty((nws,), dtype=torch.uint8, device=dy.device)
                _lib.check(L.asrk_conv3x3_first_wgrad_f32(_p(xc), _p(dyc), _p(y), _p(dw), _p(db), g.B, g.H, g.W, g.C, Cout,
                                                          g.sb, g.sh, g.sw, g.sc, _p(ws), nws, _stream()),
                           "conv3x3_first_wgrad")
            if ctx.needs_input_grad[0]:               # gradient w.r.t. the FEATURES (tests, never training): patches
                if ctx.relu:
                    masked = torch.empty_like(dyc)
                    _lib.check(L.asrk_relu_bwd_f32(_p(y), _p(dyc), _p(masked), dyc.numel(), _stream()), "relu_bwd")
                    dyc = masked
                dcol = torch.empty((g.M, g.K), dtype=torch.float32, device=dy.device)
                gemm(0, 0, g.M, g.K, Cout, dyc, Cout, w.view(Cout, g.K), g.K, dcol, g.K)
                dx = torch.zeros(ctx.x_shape, dtype=torch.float32, device=dy.device)
                _lib.check(L.asrk_col2im_f32(_p(dcol), _p(dx), *g.args(), _stream()), "col2im")
            return dx, dw, db, None, None
        if ctx.relu:
            masked = torch.empty_like(dyc)
            _lib.check(L.asrk_relu_bwd_f32(_p(y), _p(dyc), _p(masked), dyc.numel(), _stream()), "relu_bwd")
            dyc = masked
        dx = dw = db = None
        Kp = col.shape[1]
        if ctx.needs_input_grad[1]:
            dw = torch.empty((Cout, Kp), dtype=torch.float32, device=dy.device)
            gemm(1, 0, Cout, Kp, g.M, dyc, Cout, col, Kp, dw, Kp)
            if ctx.cl:                                # back to the parameter's (cin, kh, kw) order
                dwp = torch.empty_like(dw)
                _lib.check(L.asrk_conv_weight_reorder_f32(_p(dw), _p(dwp), Cout, g.C, g.KH * g.KW, 1, _stream()),
                           "conv_weight_reorder")
                dw = dwp
            elif Kp != g.K:                           # drop the pad columns
                dwp = torch.empty((Cout, g.K), dtype=torch.float32, device=dy.device)
                copy3d(dw, dwp, 1, Cout, g.K, 0, Kp, 0, g.K)
                dw = dwp
            dw = dw.view(ctx.w_shape)
        if ctx.needs_input_grad[2]:
            db = torch.empty((Cout,), dtype=torch.float32, device=dy.device)
            colsum(dyc, g.M, Cout, Cout, db)
        if ctx.needs_input_grad[0]:
            dcol = torch.empty((g.M, g.K), dtype=torch.float32, device=dy.device)   # col2im reads unpadded rows
            gemm(0, 0, g.M, g.K, Cout, dyc, Cout, w.view(Cout, Kp), Kp, dcol, g.K)
            dx = torch.zeros(ctx.x_shape, dtype=torch.float32, device=dy.device)   # cropped frames: 0
            if ctx.cl and g.channels_last_ok(dcol, dx):
                _lib.check(L.asrk_col2im_cl_f32(_p(dcol), _p(dx), *g.args(), _stream()), "col2im_cl")
            elif ctx.cl:
                raise _lib.AsrkError("conv backward: gradient buffer not 16-byte aligned")
            else:
                _lib.check(L.asrk_col2im_f32(_p(dcol), _p(dx), *g.args(), _stream()), "col2im")
        return dx, dw, db, None, None


class MaxPool2x2Fn(Function):
    """x contiguous channels-last [B,H,W,C] -> max over 2x2 / stride 2 (floor), written to a fresh
    tensor of `out_shape` at strides `ostr` = (osb, osh, osw, osc)."""

    @staticmethod
    def forward(ctx, x, dims, out_shape, ostr):
        _require_gpu(x)
        B, H, W, C = dims
        xc = _f32c(x)
        assert xc.numel() == B * H * W * C
        y = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        assert y.numel() == B * (H // 2) * (W // 2) * C
        idx = torch.empty((B * (H // 2) * (W // 2) * C,), dtype=torch.uint8, device=x.device)
        _lib.check(_L().asrk_maxpool2x2_fwd_f32(_p(xc), _p(y), _p(idx), B, H, W, C, *ostr, _stream()),
                   "maxpool")
        ctx.save_for_backward(idx)
        ctx.meta = (dims, ostr, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        (B, H, W, C), ostr, x_shape = ctx.meta
        dyc = _f32c(dy)
        dx = torch.empty(x_shape, dtype=torch.float32, device=dy.device)
        _lib.check(_L().asrk_maxpool2x2_bwd_f32(_p(dyc), _p(idx), _p(dx), B, H, W, C, *ostr, _stream()),
                   "maxpool_bwd")
        return dx, None, None, None


def conv(x, weight, bias, geom, relu=False):
    return ConvFn.apply(x, weight, bias, geom, relu)


def maxpool2x2(x, dims, out_shape, ostr):
    return MaxPool2x2Fn.apply(x, dims, out_shape, ostr)
