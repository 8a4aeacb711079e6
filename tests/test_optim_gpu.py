"""GPU parity of the fused optimiser kernels (csrc/optim.hip) against torch.optim's own CPU
single-tensor implementations over several steps, with and without folded gradient clipping
(1e-6 relative: same arithmetic, different evaluation order of a few products), and state_dict
interchange in both directions (checkpoint portability, src/solver.py:163-186)."""
import importlib

import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(7,), (33, 5), (128, 64), (3, 4, 3, 3), (1,)]


def _pair(cls_ref, cls_fused, seed, **kw):
    g = torch.Generator().manual_seed(seed)
    ps_ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in SHAPES]
    ps_dev = [p.detach().clone().to(DEV).requires_grad_(True) for p in ps_ref]
    return ps_ref, ps_dev, cls_ref(ps_ref, foreach=False, **kw), cls_fused(ps_dev, **kw), g


@pytest.mark.parametrize("name,kw", [("Adadelta", dict(lr=1.0, eps=1e-8)), ("Adadelta", dict(lr=0.5, rho=0.95, eps=1e-6)),
                                     ("Adam", dict(lr=1e-3, eps=1e-8)), ("Adam", dict(lr=0.02, betas=(0.8, 0.9), eps=1e-6))])
@pytest.mark.parametrize("clip", [None, 0.7])
def test_fused_step_matches_torch(ops, pkg, name, kw, clip):
    fo = importlib.import_module(pkg.__name__ + ".fused_optim")
    ps_ref, ps_dev, o_ref, o_dev, g = _pair(getattr(torch.optim, name), fo.FUSED[name], 11, **kw)
    for step in range(6):
        for pr, pd in zip(ps_ref, ps_dev):
            gr = torch.randn(*pr.shape, generator=g) * (3.0 if step % 2 else 0.2)
            pr.grad, pd.grad = gr.clone(), gr.clone().to(DEV)
        if clip is None:
            o_ref.step()
            o_dev.step()
        else:
            n_ref = torch.nn.utils.clip_grad_norm_(ps_ref, clip)
            o_ref.step()
            n_dev = o_dev.clip_and_step(clip)
            assert abs(float(n_dev) - float(n_ref)) <= 1e-5 * float(n_ref)
            assert torch.equal(ps_dev[-1].grad.cpu(), gr)      # fused path: gradients are left untouched
        for pr, pd in zip(ps_ref, ps_dev):
            assert rel_err(pd.detach().cpu(), pr.detach()) < 2e-6
    sd_ref, sd_dev = o_ref.state_dict(), o_dev.state_dict()
    assert sd_ref['param_groups'][0].keys() == sd_dev['param_groups'][0].keys()
    for i in sd_ref['state']:
        assert sd_ref['state'][i].keys() == sd_dev['state'][i].keys()
        for k, v in sd_ref['state'][i].items():
            assert rel_err(sd_dev['state'][i][k].cpu().float(), v.float()) < 2e-6, k


@pytest.mark.parametrize("name", ["Adadelta", "Adam"])
def test_fused_step_many_tensors_and_missing_grads(ops, pkg, name):
    """more tensors than one launch table holds (24), some without a gradient, one large enough for
    the per-tensor block cap"""
    fo = importlib.import_module(pkg.__name__ + ".fused_optim")
    g = torch.Generator().manual_seed(5)
    shapes = [((3 + i, 5) if i % 9 != 4 else (0, 5)) for i in range(53)] + [(2100, 1100)]   # incl. empty tensors
    ps_ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    ps_dev = [p.detach().clone().to(DEV).requires_grad_(True) for p in ps_ref]
    o_ref = getattr(torch.optim, name)(ps_ref, foreach=False, lr=0.5)
    o_dev = fo.FUSED[name](ps_dev, lr=0.5)
    for step in range(3):
        for i, (pr, pd) in enumerate(zip(ps_ref, ps_dev)):
            if i % 7 == 3 and step != 1:          # no gradient this step (joins / skips steps)
                pr.grad, pd.grad = None, None
                continue
            gr = torch.randn(*pr.shape, generator=g)
            pr.grad, pd.grad = gr.clone(), gr.clone().to(DEV)
        o_ref.step()
        o_dev.step()
        for pr, pd in zip(ps_ref, ps_dev):
            if pr.numel():
                assert rel_err(pd.detach().cpu(), pr.detach()) < 2e-6


def test_state_dict_round_trip_between_torch_and_fused(ops, pkg):
    fo = importlib.import_module(pkg.__name__ + ".fused_optim")
    g = torch.Generator().manual_seed(3)
    p_t = [torch.randn(16, 8, generator=g).to(DEV).requires_grad_(True)]
    p_f = [p_t[0].detach().clone().requires_grad_(True)]
    o_t, o_f = torch.optim.Adadelta(p_t, lr=1.0, eps=1e-8), fo.FusedAdadelta(p_f, lr=1.0, eps=1e-8)
    for o, p in ((o_t, p_t), (o_f, p_f)):
        p[0].grad = torch.ones_like(p[0])
        o.step()
    # fused state -> torch optimiser and back: the next steps agree
    o_t.load_state_dict(o_f.state_dict())
    o_f.load_state_dict(o_t.state_dict())
    for o, p in ((o_t, p_t), (o_f, p_f)):
        p[0].grad = torch.full_like(p[0], 0.3)
        o.step()
    assert rel_err(p_f[0].detach().cpu(), p_t[0].detach().cpu()) < 2e-6


def test_wrapper_selects_fused_and_folds_clipping(ops, pkg):
    optim = importlib.import_module(pkg.__name__ + ".src.optim")
    fo = importlib.import_module(pkg.__name__ + ".fused_optim")
    p = torch.nn.Parameter(torch.randn(32, 4, device=DEV))
    o = optim.Optimizer([{'params': iter([p])}], 'Adadelta', 1.0, 1e-8, 'fixed')
    assert o.fused and isinstance(o.opt, fo.FusedAdadelta)
    o.pre_step(0)
    p.grad = torch.full_like(p, 10.0)
    before = p.detach().clone()
    norm = fo.total_grad_norm([p])
    o.step(norm, 5.0)
    assert float(norm) == pytest.approx(10.0 * (128 ** 0.5), rel=1e-6)
    assert not torch.equal(p.detach(), before) and torch.equal(p.grad, torch.full_like(p, 10.0))


@pytest.mark.parametrize("name", ["Adadelta", "Adam"])
def test_nan_gradient_norm_skips_the_update_on_the_device(ops, pkg, name):
    """the NaN guard of src/solver.py:85-89 as a device-side predicate (csrc/optim.hip skip_of): a NaN anywhere in
    the gradients makes the norm and hence the clipping coefficient NaN, and the fused update then leaves parameters
    AND optimiser state bit-identical - with nothing read back by the host; the next clean step proceeds normally"""
    fo = importlib.import_module(pkg.__name__ + ".fused_optim")
    g = torch.Generator().manual_seed(9)
    ps = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in SHAPES]
    opt = fo.FUSED[name](ps, lr=0.5)
    for p in ps:
        p.grad = torch.randn(*p.shape, generator=g).to(DEV)
    opt.clip_and_step(5.0)                                        # creates the state
    snap_p = [p.detach().clone() for p in ps]
    snap_s = [{k: v.clone() for k, v in opt.state[p].items() if torch.is_tensor(v) and v.is_cuda} for p in ps]
    for p in ps:
        p.grad = torch.randn(*p.shape, generator=g).to(DEV)
    ps[1].grad[3, 2] = float("nan")
    norm, coef = fo.grad_norm_and_coef(ps, 5.0)
    assert torch.isnan(norm) and torch.isnan(coef).all()
    opt.step(clip_coef=coef)
    for p, sp, ss in zip(ps, snap_p, snap_s):
        assert torch.equal(p.detach(), sp)
        for k, v in ss.items():
            assert torch.equal(opt.state[p][k], v), k
    for p in ps:
        p.grad = torch.randn(*p.shape, generator=g).to(DEV)
    opt.clip_and_step(5.0)
    assert all(torch.isfinite(p).all() and not torch.equal(p.detach(), sp) for p, sp in zip(ps, snap_p))


def test_solver_backward_is_syncless_and_skips_nan_steps(ops, pkg):
    """BaseSolver.backward with the fused Adadelta: the norm comes back as a 0-d DEVICE tensor (no host read on the
    step), a NaN loss leaves the model untouched, and poll_device_errors reports it afterwards"""
    solver_mod = importlib.import_module(pkg.__name__ + ".src.solver")
    optim = importlib.import_module(pkg.__name__ + ".src.optim")
    util = importlib.import_module(pkg.__name__ + ".src.util")
    msgs = []

    class Paras:
        verbose = True
    s = object.__new__(solver_mod.BaseSolver)
    s.paras, s.rank, s.world, s.dist, s.dp, s.step = Paras(), 0, 1, None, None, 0
    s.GRAD_CLIP, s.timer = 5.0, util.Timer()
    s.verbose = lambda m: msgs.append(m)
    torch.manual_seed(0)
    s.model = torch.nn.Linear(6, 3).to(DEV)
    s.optimizer = optim.Optimizer(s.model.parameters(), 'Adadelta', 1.0, 1e-8, 'fixed')
    assert s.optimizer.fused and s.optimizer.device_nan_skip
    x = torch.randn(4, 6, device=DEV)
    for step, poison in enumerate([False, True, False]):
        s.optimizer.pre_step(step)
        before = [p.detach().clone() for p in s.model.parameters()]
        loss = s.model(x).pow(2).mean() * (float("nan") if poison else 1.0)
        gn = s.backward(loss)
        s.step += 1
        assert torch.is_tensor(gn) and gn.is_cuda and gn.dim() == 0
        same = all(torch.equal(a, b.detach()) for a, b in zip(before, s.model.parameters()))
        assert same == poison
    assert not msgs
    s.poll_device_errors(force=True)
    assert msgs == ['Error : grad norm is NaN @ step 1']
