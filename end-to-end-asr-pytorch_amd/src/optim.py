"""Optimizer wrapper with learning-rate and teacher-forcing schedules — mirror of the reference's
src/optim.py:6-82 (same constructor kwargs as the `hparas` config block, same `pre_step` /
`step` / state-dict surface).  The update itself is torch.optim (elementwise HBM-bound work on
resident parameters); the schedules are host scalars.
"""
import math
from functools import partial

import torch


def warmup_scheduler(step, init_lr, warmup_step=4000.0):
    ''' "Noam" warm-up, scaled so that the peak equals init_lr (src/optim.py:20-25) '''
    s = step + 1
    return init_lr * warmup_step ** 0.5 * min(s * warmup_step ** -1.5, s ** -0.5)


def speech_aug_scheduler(step, s_r, s_i, s_f, peak_lr):
    ''' SpecAugment schedule: ramp to peak_lr over s_r steps, hold until s_i, base-10 exponential
        decay to 0.01*peak_lr at s_f (src/optim.py:65-82) '''
    final_lr_ratio = 0.01
    cur_step = step + 1
    if cur_step < s_r:
        return peak_lr * float(cur_step) / s_r
    if cur_step < s_i:
        return peak_lr
    if cur_step <= s_f:
        decay = -math.log10(final_lr_ratio) / (s_f - s_i)
        return peak_lr * 10.0 ** (-decay * (cur_step - s_i))
    return peak_lr * final_lr_ratio


class Optimizer():
    def __init__(self, parameters, optimizer, lr, eps, lr_scheduler, tf_start=1, tf_end=1, tf_step=1, **kwargs):
        # scheduled sampling: linear from tf_start to tf_end over tf_step steps
        self.tf_type = tf_end != 1
        self.tf_rate = lambda step: max(tf_end, tf_start - (tf_start - tf_end) * step / tf_step)
        self.opt_type = optimizer
        self.init_lr = lr
        self.sch_type = lr_scheduler
        opt = getattr(torch.optim, optimizer)
        # Adadelta / Adam on GPU parameters: one fused streaming kernel per tensor (same state layout)
        self.fused = False
        # True: the fused update skips itself on a NaN clipping coefficient and keeps no host-side step counter that
        # a skipped step would corrupt (Adadelta); Adam's bias correction counts steps on the host -> host-side guard
        self.device_nan_skip = False
        # materialise the (possibly generator-valued) parameter spec so it can be inspected
        parameters = [dict(g, params=list(g['params'])) if isinstance(g, dict) else g for g in parameters]
        tensors = [p for g in parameters for p in (g['params'] if isinstance(g, dict) else [g])]
        if tensors and all(p.is_cuda for p in tensors) and kwargs.get('fused_step', True):
            from ..fused_optim import FUSED
            if optimizer in FUSED:
                opt = FUSED[optimizer]
                self.fused = True
                self.device_nan_skip = optimizer == 'Adadelta'
        if lr_scheduler == 'warmup':
            self.lr_scheduler = partial(warmup_scheduler, init_lr=lr)
            self.opt = opt(parameters, lr=1.0)
        elif lr_scheduler == 'spec-aug-basic':
            self.lr_scheduler = partial(speech_aug_scheduler, s_r=500, s_i=20000, s_f=80000, peak_lr=lr)
            self.opt = opt(parameters, lr=lr, eps=eps)
        elif lr_scheduler == 'spec-aug-double':
            self.lr_scheduler = partial(speech_aug_scheduler, s_r=1000, s_i=40000, s_f=160000, peak_lr=lr)
            self.opt = opt(parameters, lr=lr, eps=eps)
        else:
            self.lr_scheduler = None
            self.opt = opt(parameters, lr=lr, eps=eps)

    def get_opt_state_dict(self):
        return self.opt.state_dict()

    def load_opt_state_dict(self, state_dict):
        self.opt.load_state_dict(state_dict)

    def pre_step(self, step):
        ''' set this step's lr, clear gradients, return the teacher-forcing rate '''
        if self.lr_scheduler is not None:
            cur_lr = self.lr_scheduler(step)
            for param_group in self.opt.param_groups:
                param_group['lr'] = cur_lr
        self.opt.zero_grad()
        return self.tf_rate(step)

    def step(self, grad_norm=None, max_norm=None, coef=None):
        """plain step, or (fused optimisers) step with gradient clipping folded in: pass the total
        gradient norm (device scalar) and the clip threshold, or the ready clipping coefficient
        max_norm / (norm + 1e-6) as a device tensor [1] (fused_optim.grad_norm_and_coef)"""
        if self.fused and (grad_norm is not None or coef is not None):
            if coef is None:
                coef = (max_norm / (grad_norm + 1e-6)).to(torch.float32).reshape(1)
            self.opt.step(clip_coef=coef)
        else:
            self.opt.step()

    def create_msg(self):
        return ['Optim.spec.| Algo. = {}\t| Lr = {}\t (Scheduler = {})| Scheduled sampling = {}'
                .format(self.opt_type, self.init_lr, self.sch_type, self.tf_type)]
