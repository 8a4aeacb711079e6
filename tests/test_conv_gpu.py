"""GPU parity of the convolutional-prenet kernels (csrc/conv.hip via conv_ops): channels-last
im2col + MFMA GEMM convolution, col2im adjoint, fused ReLU, 2x2 max pool — against torch CPU fp64
conv2d / conv1d / max_pool2d on the same seeded inputs (1e-3 relative, north_star), and the two
extractor modules against the oracle restatement of src/module.py:7-90."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import asr_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def C(pkg):
    return importlib.import_module(pkg.__name__ + ".conv_ops")


@pytest.mark.parametrize("B,H,W,Cin,Cout,relu", [(2, 8, 13, 2, 64, True), (1, 5, 6, 64, 128, True),
                                                 (3, 12, 10, 7, 9, False), (2, 4, 4, 128, 128, True),
                                                 # the implicit-GEMM layers (csrc/conv3x3.hip): several row tiles with a
                                                 # ragged last one, the VGG widths 40 / 20 / 13 / 6, every (Cin, Cout) pair
                                                 (2, 50, 40, 64, 64, True), (2, 23, 20, 64, 128, True),
                                                 (1, 37, 20, 128, 128, False), (2, 9, 13, 128, 64, True),
                                                 (1, 3, 128, 64, 64, True), (2, 1, 1, 64, 64, True),
                                                 (3, 11, 6, 128, 128, True), (1, 7, 33, 64, 64, False),
                                                 # the few-plane first layer (one MFMA tile deep)
                                                 (2, 50, 40, 3, 64, True), (1, 9, 13, 1, 64, True), (2, 7, 40, 2, 128, False)])
def test_conv3x3_channels_last_matches_conv2d(ops, C, B, H, W, Cin, Cout, relu):
    g = torch.Generator().manual_seed(B * 100 + H * 10 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    dy = torch.randn(B, Cout, H, W, generator=g)
    xr, wr, br = [v.double().requires_grad_(True) for v in (x, w, b)]
    yr = F.conv2d(xr, wr, br, padding=1)
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.double())
    # channels-last device tensors
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    geom = C.Geom(B, H, W, Cin, 3, 3, 1, 1, 1, 1, H * W * Cin, W * Cin, Cin, 1)
    y = C.conv(xd, wd, bd, geom, relu=relu)
    assert y.shape == (B * H * W, Cout)
    y.backward(dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV))
    assert rel_err(y.detach().cpu().view(B, H, W, Cout).permute(0, 3, 1, 2), yr.detach()) < 1e-4
    assert rel_err(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad) < 1e-3
    assert rel_err(wd.grad.cpu(), wr.grad) < 1e-3
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-3


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 31, 40, 64, 64), (2, 14, 20, 64, 128), (1, 14, 20, 128, 128)])
def test_implicit_gemm_layers_equal_the_im2col_path(ops, C, monkeypatch, B, H, W, Cin, Cout):
    """same f32 products, another summation order: the two paths agree far inside the parity tolerance, and the
    implicit-GEMM weight gradient is bit-reproducible (fixed-order slab sums, no atomics)"""
    g = torch.Generator().manual_seed(W + Cin)
    x = torch.randn(B * H * W, Cin, generator=g).relu().to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    dy = torch.randn(B * H * W, Cout, generator=g).to(DEV)
    geom = C.Geom(B, H, W, Cin, 3, 3, 1, 1, 1, 1, H * W * Cin, W * Cin, Cin, 1)
    assert geom.direct3x3_ok(Cout, x, w)

    def run():
        xd, wd, bd = [v.clone().requires_grad_(True) for v in (x, w, b)]
        y = C.conv(xd, wd, bd, geom, relu=True)
        y.backward(dy)
        return [t.detach().cpu() for t in (y, xd.grad, wd.grad, bd.grad)]

    direct, again = run(), run()
    monkeypatch.setenv("ASRK_CONV_DIRECT", "0")
    assert not geom.direct3x3_ok(Cout, x, w)
    patches = run()
    ops.check_errors()
    for a, r in zip(direct, patches):
        assert rel_err(a, r) < 2e-6
    for a, r in zip(direct, again):
        assert torch.equal(a, r)


@pytest.mark.parametrize("B,T,Din,Cout", [(3, 30, 10, 12), (2, 9, 80, 32), (1, 4, 5, 3)])
def test_strided_conv1d_batch_as_width(ops, C, B, T, Din, Cout):
    """CNNExtractor geometry: reads batch-major [B,T,D] in place, emits time-major [T',B,Cout]"""
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, Din, generator=g)
    w = torch.randn(Cout, Din, 4, generator=g) / (2 * Din ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = [v.double().requires_grad_(True) for v in (x, w, b)]
    yr = F.conv1d(xr.transpose(1, 2), wr, br, stride=2, padding=1).transpose(1, 2)       # [B,T',Cout]
    dy = torch.randn(*yr.shape, generator=g)
    yr.backward(dy.double())
    xd, wd, bd = [v.to(DEV).requires_grad_(True) for v in (x, w, b)]
    geom = C.Geom(1, T, B, Din, 4, 1, 2, 1, 1, 0, 0, Din, T * Din, 1)
    y = C.conv(xd, wd, bd, geom)
    Tp = geom.Ho
    assert Tp == yr.shape[1] and y.shape == (Tp * B, Cout)
    y.backward(dy.transpose(0, 1).reshape(-1, Cout).contiguous().to(DEV))
    assert rel_err(y.detach().cpu().view(Tp, B, Cout).transpose(0, 1), yr.detach()) < 1e-4
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-3
    assert rel_err(wd.grad.cpu(), wr.grad) < 1e-3
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-3


@pytest.mark.parametrize("B,H,W,Ch", [(2, 8, 13, 64), (1, 5, 7, 3), (3, 4, 6, 128)])
def test_maxpool2x2_matches_torch(ops, C, B, H, W, Ch):
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, Ch, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, stride=2)
    dy = torch.randn(*yr.shape, generator=g)
    yr.backward(dy)
    Ho, Wo = H // 2, W // 2
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    # (a) channels-last output, (b) the VGG tail layout [Ho, B, C*Wo] (time-major, channel-major)
    for out_shape, ostr, back in (
            ((B, Ho, Wo, Ch), (Ho * Wo * Ch, Wo * Ch, Ch, 1), lambda t: t.permute(0, 3, 1, 2)),
            ((Ho, B, Ch * Wo), (Ch * Wo, B * Ch * Wo, 1, Wo), lambda t: t.view(Ho, B, Ch, Wo).permute(1, 2, 0, 3))):
        xd.grad = None
        y = C.maxpool2x2(xd, (B, H, W, Ch), out_shape, ostr)
        assert torch.equal(back(y.detach().cpu()), yr.detach())                         # exact
        dyd = torch.empty(out_shape)
        back(dyd).copy_(dy)
        y.backward(dyd.to(DEV))
        assert torch.equal(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad)


@pytest.mark.parametrize("D,T", [(26, 23), (40, 16), (80, 21)])
def test_vgg_extractor_matches_oracle(ops, pkg, D, T):
    module = importlib.import_module(pkg.__name__ + ".src.module")
    torch.manual_seed(D)
    vgg = module.VGGExtractor(D)
    sd = {'x.' + k: v.detach().clone() for k, v in vgg.state_dict().items()}
    x = torch.randn(2, T, D)
    xlen = torch.tensor([T, T - 5])
    xr = x.clone().requires_grad_(True)
    sdr = {k: v.requires_grad_(True) for k, v in sd.items()}
    yr, lr = O.vgg_forward(sdr, xr, xlen, 'x.extractor.')
    dy = torch.randn(*yr.shape)
    yr.backward(dy)
    vgg = vgg.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y, l = vgg(xd, xlen.to(DEV))
    assert y.shape == yr.shape == (2, (T - T % 4) // 4, vgg.out_dim) and torch.equal(l.cpu(), lr)
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-3
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-3
    for n, p in vgg.named_parameters():
        assert rel_err(p.grad.cpu(), sdr['x.' + n].grad) < 1e-3, n


def test_cnn_extractor_matches_oracle(ops, pkg):
    module = importlib.import_module(pkg.__name__ + ".src.module")
    torch.manual_seed(4)
    cnn = module.CNNExtractor(10, 12)
    sd = {'x.' + k: v.detach().clone().requires_grad_(True) for k, v in cnn.state_dict().items()}
    x = torch.randn(3, 30, 10)
    xlen = torch.tensor([30, 22, 9])
    xr = x.clone().requires_grad_(True)
    yr, lr = O.cnn_forward(sd, xr, xlen, 'x.extractor.')
    dy = torch.randn(*yr.shape)
    yr.backward(dy)
    cnn = cnn.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y, l = cnn(xd, xlen.to(DEV))
    assert y.shape == yr.shape and torch.equal(l.cpu(), lr)
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-3
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-3
    for n, p in cnn.named_parameters():
        assert rel_err(p.grad.cpu(), sd['x.' + n].grad) < 1e-3, n


def test_conv_rejects_bad_geometry(ops, C):
    with pytest.raises(RuntimeError):
        C.Geom(1, 2, 2, 1, 3, 3, 1, 1, 0, 0, 4, 2, 1, 1)          # kernel larger than the input
    x = torch.zeros(10, device=DEV)
    g = C.Geom(1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 16, 4, 1, 1)
    with pytest.raises(RuntimeError):
        C.conv(x, torch.zeros(2, 1, 3, 3, device=DEV), torch.zeros(2, device=DEV), g)   # 16 > 10 elements
