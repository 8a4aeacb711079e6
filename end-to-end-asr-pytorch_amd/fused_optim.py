"""Fused optimiser steps over the gfx950 kernels of csrc/optim.hip.

`FusedAdadelta` / `FusedAdam` subclass the torch optimisers (same constructor, same `state` /
`state_dict()` layout, so checkpoints written by either side load in the other) and replace
`step()` by one streaming kernel launch per param_group (all its tensors in one call).  `clip_and_step()` folds
`torch.nn.utils.clip_grad_norm_` (src/solver.py:84) into that pass: the global norm is reduced once,
the clipping coefficient stays on the device and the gradients are never rewritten."""
import ctypes

import torch

from . import _lib
from .ops import _L, _p, _stream


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _grads(params):
    return [p.grad for p in params if p.grad is not None]


def grad_norm_and_coef(params, max_norm):
    """(2-norm over all gradients, max_norm / (norm + 1e-6)) as device scalars: what clip_grad_norm_ computes
    (src/solver.py:84), by the two-stage deterministic reduction of csrc/optim.hip (asrk_grad_norm_multi_f32)"""
    gs = [g if g.is_contiguous() else g.contiguous() for g in _grads(params)]
    dev = gs[0].device if gs else (params[0].device if params else torch.device("cpu"))
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    if not gs or not gs[0].is_cuda or any(g.dtype != torch.float32 for g in gs):
        raise _lib.AsrkError("gradient norm: f32 gradients on the GPU expected")
    L = _L()
    numel = (ctypes.c_int64 * len(gs))(*[g.numel() for g in gs])
    nws = int(L.asrk_grad_norm_ws_bytes(len(gs), numel))
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    _lib.check(L.asrk_grad_norm_multi_f32(len(gs), _ptr_array(gs), numel, float(max_norm), _p(out[0:1]), _p(out[1:2]),
                                          _p(ws), nws, _stream()), "grad_norm")
    return out[0], out[1:2]


def total_grad_norm(params):
    """2-norm over all gradients (what clip_grad_norm_ returns), as a device scalar"""
    if not _grads(params):
        return torch.zeros((), device=params[0].device if params else "cpu")
    return grad_norm_and_coef(params, 1.0)[0]


class _FusedMixin:
    _state_keys = ()

    def _launch_group(self, group, params, grads, st0, st1, numel, coef):
        raise NotImplementedError

    @torch.no_grad()
    def step(self, closure=None, clip_coef=None):
        """clip_coef: optional device scalar max_norm / (total_norm + 1e-6).  All tensors of a
        param_group go to the device in ONE call (asrk_*_multi_f32)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        k0, k1 = self._state_keys
        for group in self.param_groups:
            if group.get('weight_decay', 0) != 0 or group.get('maximize', False) or group.get('amsgrad', False):
                raise NotImplementedError("fused step: weight_decay / maximize / amsgrad are not used by the "
                                          "reference configs")
            params, grads, st0, st1 = [], [], [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    raise _lib.AsrkError("fused optimiser needs contiguous f32 parameters on the GPU")
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st[k0] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st[k1] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                params.append(p)
                grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                st0.append(st[k0])
                st1.append(st[k1])
            if not params:
                continue
            numel = (ctypes.c_int64 * len(params))(*[p.numel() for p in params])
            self._launch_group(group, params, _ptr_array(params), _ptr_array(grads), _ptr_array(st0),
                               _ptr_array(st1), numel, clip_coef)
        return loss

    def clip_and_step(self, max_norm):
        """clip_grad_norm_(params, max_norm) + step() in one pass over the gradients; returns the
        total gradient norm (device scalar).  A NaN norm poisons nothing: the caller decides whether to
        call this at all (src/solver.py:85-89 checks the norm first)."""
        params = [p for g in self.param_groups for p in g['params']]
        if not _grads(params):
            return torch.zeros((), device=params[0].device if params else "cpu")
        norm, coef = grad_norm_and_coef(params, max_norm)
        self.step(clip_coef=coef)
        return norm


class FusedAdadelta(_FusedMixin, torch.optim.Adadelta):
    _state_keys = ('square_avg', 'acc_delta')

    def _launch_group(self, group, params, pp, gp, s0, s1, numel, coef):
        _lib.check(_L().asrk_adadelta_multi_f32(len(params), pp, gp, s0, s1, numel, float(group['lr']),
                                                float(group['rho']), float(group['eps']), _p(coef),
                                                _stream()), "adadelta_multi")


class FusedAdam(_FusedMixin, torch.optim.Adam):
    _state_keys = ('exp_avg', 'exp_avg_sq')

    def _launch_group(self, group, params, pp, gp, s0, s1, numel, coef):
        b1, b2 = group['betas']
        # one bias correction per call: tensors that joined the optimiser later (different step
        # counts) go in separate calls
        by_step = {}
        for i, p in enumerate(params):
            by_step.setdefault(int(self.state[p]['step'].item()), []).append(i)
        for step, idx in by_step.items():
            def sub(arr, ty=ctypes.c_void_p):
                return (ty * len(idx))(*[arr[i] for i in idx])
            _lib.check(_L().asrk_adam_multi_f32(len(idx), sub(pp), sub(gp), sub(s0), sub(s1),
                                                sub(numel, ctypes.c_int64), float(group['lr']), float(b1),
                                                float(b2), float(group['eps']), step, _p(coef), _stream()),
                       "adam_multi")


FUSED = {'Adadelta': FusedAdadelta, 'Adam': FusedAdam}
