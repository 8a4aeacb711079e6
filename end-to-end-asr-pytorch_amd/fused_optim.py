"""Fused optimiser steps over the gfx950 kernels of csrc/optim.hip.

`FusedAdadelta` / `FusedAdam` subclass the torch optimisers (same constructor, same `state` /
`state_dict()` layout, so checkpoints written by either side load in the other) and replace
`step()` by one streaming kernel per parameter.  `clip_and_step()` folds
`torch.nn.utils.clip_grad_norm_` (src/solver.py:84) into that pass: the global norm is reduced once,
the clipping coefficient stays on the device and the gradients are never rewritten."""
import torch

from . import _lib
from .ops import _L, _p, _stream


def _grads(params):
    return [p.grad for p in params if p.grad is not None]


def total_grad_norm(params):
    """2-norm over all gradients (what clip_grad_norm_ returns), as a device scalar"""
    gs = _grads(params)
    if not gs:
        return torch.zeros((), device=params[0].device if params else "cpu")
    return torch.linalg.vector_norm(torch.stack(torch._foreach_norm(gs, 2.0)), 2.0)


class _FusedMixin:
    def _launch(self, group, p, coef):
        raise NotImplementedError

    @torch.no_grad()
    def step(self, closure=None, clip_coef=None):
        """clip_coef: optional device scalar max_norm / (total_norm + 1e-6)"""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get('weight_decay', 0) != 0 or group.get('maximize', False) or group.get('amsgrad', False):
                raise NotImplementedError("fused step: weight_decay / maximize / amsgrad are not used by the "
                                          "reference configs")
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    raise _lib.AsrkError("fused optimiser needs contiguous f32 parameters on the GPU")
                self._launch(group, p, clip_coef)
        return loss

    def clip_and_step(self, max_norm):
        """clip_grad_norm_(params, max_norm) + step() in one pass over the gradients; returns the
        total gradient norm (device scalar).  A NaN norm poisons nothing: the caller decides whether to
        call this at all (src/solver.py:85-89 checks the norm first)."""
        params = [p for g in self.param_groups for p in g['params']]
        norm = total_grad_norm(params)
        coef = (max_norm / (norm + 1e-6)).to(torch.float32).reshape(1)
        self.step(clip_coef=coef)
        return norm


class FusedAdadelta(_FusedMixin, torch.optim.Adadelta):
    def _launch(self, group, p, coef):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.zeros((), dtype=torch.float32)
            st['square_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['acc_delta'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st['step'] += 1
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        _lib.check(_L().asrk_adadelta_step_f32(_p(p), _p(g), _p(st['square_avg']), _p(st['acc_delta']),
                                               p.numel(), float(group['lr']), float(group['rho']),
                                               float(group['eps']), _p(coef), _stream()), "adadelta_step")


class FusedAdam(_FusedMixin, torch.optim.Adam):
    def _launch(self, group, p, coef):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.zeros((), dtype=torch.float32)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st['step'] += 1
        b1, b2 = group['betas']
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        _lib.check(_L().asrk_adam_step_f32(_p(p), _p(g), _p(st['exp_avg']), _p(st['exp_avg_sq']), p.numel(),
                                           float(group['lr']), float(b1), float(b2), float(group['eps']),
                                           int(st['step'].item()), _p(coef), _stream()), "adam_step")


FUSED = {'Adadelta': FusedAdadelta, 'Adam': FusedAdam}
