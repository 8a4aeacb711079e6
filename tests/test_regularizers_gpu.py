"""GPU parity of the per-frame regularisers (csrc/norm.hip): LayerNorm fwd/bwd against the fp32/fp64
torch CPU op (1e-3 relative, north_star), dropout against the numpy Philox oracle (bit-exact mask,
exact fp32 values) plus the properties nn.Dropout guarantees (keep rate, 1/(1-p) scale, identical
mask in forward and backward, identity in eval)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import regularizer_oracle as R
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("shape", [(7, 3, 36), (3, 1, 5), (250, 32, 1024), (5, 2050), (33, 4, 129)])
def test_layer_norm_matches_torch(ops, shape):
    g = torch.Generator().manual_seed(sum(shape))
    D = shape[-1]
    x = (torch.randn(*shape, generator=g) * 3 + 0.5)
    w = torch.randn(D, generator=g)
    b = torch.randn(D, generator=g)
    dy = torch.randn(*shape, generator=g)
    xr, wr, br = [v.double().requires_grad_(True) for v in (x, w, b)]
    yr = F.layer_norm(xr, (D,), wr, br, 1e-5)
    yr.backward(dy.double())
    xd, wd, bd = [v.to(DEV).requires_grad_(True) for v in (x, w, b)]
    y = ops.layer_norm(xd, wd, bd, 1e-5)
    y.backward(dy.to(DEV))
    assert y.shape == x.shape
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-4
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-3
    assert rel_err(wd.grad.cpu(), wr.grad) < 1e-3
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-3


def test_layer_norm_rejects_bad_shape(ops):
    with pytest.raises(RuntimeError):
        ops.layer_norm(torch.zeros(4, 8, device=DEV), torch.ones(7, device=DEV), torch.zeros(7, device=DEV), 1e-5)


@pytest.mark.parametrize("n,p,seed", [(4096, 0.1, 1), (1003, 0.5, 2 ** 40 + 17), (1, 0.3, 5),
                                      (3 * 250 * 1024, 0.2, 2 ** 61 + 3), (7, 0.0001, 9)])
def test_dropout_mask_is_the_documented_philox_function(ops, n, p, seed):
    x = torch.randn(n, generator=torch.Generator().manual_seed(n)) + 3.0      # no zeros
    y = ops.dropout(x.to(DEV).requires_grad_(True), p, True, seed=seed)
    ref = R.dropout(x.numpy(), p, seed)
    assert np.array_equal(y.detach().cpu().numpy(), ref)                      # mask AND values exact
    # unaligned base pointer takes the scalar path: same mask
    xo = torch.empty(n + 1, device=DEV)
    xo[1:] = x.to(DEV)
    y2 = ops.dropout(xo[1:], p, True, seed=seed)
    assert np.array_equal(y2.cpu().numpy(), ref)


def test_dropout_properties(ops):
    torch.manual_seed(0)
    n, p = 1 << 20, 0.3
    x = torch.ones(n, device=DEV, requires_grad=True)
    y = ops.dropout(x, p, True)
    kept = (y != 0)
    assert abs(kept.float().mean().item() - (1 - p)) < 5e-3
    assert torch.allclose(y[kept], torch.full_like(y[kept], 1 / (1 - p)))
    dy = torch.randn(n, device=DEV)
    y.backward(dy)
    assert torch.equal(x.grad != 0, kept & (dy != 0))                          # same mask in backward
    assert torch.allclose(x.grad[kept], dy[kept] / (1 - p))
    # a fresh call draws a fresh seed from torch's CPU generator; re-seeding reproduces it
    torch.manual_seed(0)
    y_again = ops.dropout(x.detach(), p, True)
    y_other = ops.dropout(x.detach(), p, True)
    assert torch.equal(y_again, y.detach()) and not torch.equal(y_other, y.detach())
    # eval mode / p == 0: identity, same tensor
    assert ops.dropout(x, p, False) is x and ops.dropout(x, 0.0, True) is x
    with pytest.raises(ValueError):
        ops.dropout(x, 1.0, True)


def test_encoder_layer_with_dropout_trains_and_evals(ops, pkg):
    """RNNLayer(dropout>0, layer_norm=True): train-mode output differs by the mask only; eval equals
    the no-dropout layer (reference: src/module.py:135-138)."""
    import importlib
    module = importlib.import_module(pkg.__name__ + ".src.module")
    torch.manual_seed(3)
    layer = module.RNNLayer(12, 'LSTM', 16, True, 0.25, True, 2, 'drop', True).to(DEV)
    x = torch.randn(4, 10, 12, device=DEV, requires_grad=True)
    xlen = torch.tensor([10, 9, 7, 4], device=DEV)
    layer.eval()
    y_eval, l_eval = layer(x, xlen)
    layer.train()
    torch.manual_seed(11)
    y_tr, l_tr = layer(x, xlen)
    assert y_tr.shape == y_eval.shape == (4, 5, 32) and torch.equal(l_tr, xlen // 2)
    assert not torch.allclose(y_tr, y_eval)
    y_tr.sum().backward()
    ops.check_errors()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())
