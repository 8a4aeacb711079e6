"""Host-side utilities mirroring the reference's src/util.py (only what the hot path needs)."""
import math
import time

import numpy as np
import torch
from torch import nn


class Timer():
    """Wall-clock buckets rd/fw/bw (reference: src/util.py:13-42).  Unlike the reference this one
    can synchronise the device so the buckets mean something under async HIP execution."""

    def __init__(self, sync=False):
        self.sync = sync
        self.prev_t = time.time()
        self.clear()

    def _now(self):
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        return time.time()

    def set(self):
        self.prev_t = self._now()

    def cnt(self, mode):
        self.time_table[mode] += self._now() - self.prev_t
        self.set()
        if mode == 'bw':
            self.click += 1

    def show(self):
        total_time = sum(self.time_table.values())
        self.time_table['avg'] = total_time / max(self.click, 1)
        for k in ('rd', 'fw', 'bw'):
            self.time_table[k] = 100 * self.time_table[k] / max(total_time, 1e-12)
        msg = '{avg:.3f} sec/step (rd {rd:.1f}% | fw {fw:.1f}% | bw {bw:.1f}%)'.format(
            **self.time_table)
        self.clear()
        return msg

    def clear(self):
        self.time_table = {'rd': 0, 'fw': 0, 'bw': 0}
        self.click = 0


def init_weights(module):
    """Reference: src/util.py:47-70 (applied through Module.apply when the optimizer is Adadelta,
    src/asr.py:41-42).  Embedding ~ N(0,1); biases 0; 2-D ~ N(0, 1/sqrt(fan_in)); conv 3/4-D ~
    N(0, 1/sqrt(in*k))."""
    if type(module) == nn.Embedding:
        module.weight.data.normal_(0, 1)
    else:
        for p in module.parameters():
            data = p.data
            if data.dim() == 1:
                data.zero_()
            elif data.dim() == 2:
                n = data.size(1)
                data.normal_(0, 1. / math.sqrt(n))
            elif data.dim() in [3, 4]:
                n = data.size(1)
                for k in data.size()[2:]:
                    n *= k
                data.normal_(0, 1. / math.sqrt(n))
            else:
                raise NotImplementedError


def init_gate(bias):
    """Forget-gate bias = 1 (reference: src/util.py:73-77)."""
    n = bias.size(0)
    start, end = n // 4, n // 2
    bias.data[start:end].fill_(1.)
    return bias


def human_format(num):
    magnitude = 0
    while num >= 1000:
        magnitude += 1
        num /= 1000.0
    return '{:3.1f}{}'.format(num, [' ', 'K', 'M', 'G', 'T', 'P'][magnitude])


def edit_distance(a, b):
    """Levenshtein distance between two sequences (replaces the absent `editdistance` package
    used at src/util.py:113-127)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def _argmax(pred):
    """hypothesis read-out: the gfx950 top-1 kernel for device tensors (host tensors only occur in
    CPU-side unit tests of this metric helper)"""
    if pred.is_cuda:
        from .. import ops
        return ops.argmax(pred)
    return pred.argmax(dim=-1)


def _cal_er_device(tokenizer, pred, truth, mode, ctc):
    """the batch on the device: arg-max ids -> crop / CTC collapse of hypotheses AND references in one launch
    each (csrc/metrics.hip), ONE D2H of the compacted ids; the tokenizer turns ids into text on the host (a
    sentencepiece model is host code by nature), the words / characters are interned to integers, and the B
    Levenshtein programmes run as one device launch (asrk_edit_distance_i64)."""
    from .. import ops
    import torch
    dev = pred.device
    truth = truth.to(dev)
    hyp, hyp_n = ops.token_crop(pred, tokenizer.pad_idx, tokenizer.eos_idx, ignore_repeat=ctc)
    ref, ref_n = ops.token_crop(truth, tokenizer.pad_idx, tokenizer.eos_idx, ignore_repeat=False)
    hyp, hyp_n, ref, ref_n = hyp.cpu(), hyp_n.cpu().tolist(), ref.cpu(), ref_n.cpu().tolist()
    B = len(hyp_n)
    intern, seqs = {}, []
    for rows, lens in ((hyp, hyp_n), (ref, ref_n)):
        for b in range(B):
            s = tokenizer.decode(rows[b, :lens[b]].tolist())
            units = s.split(' ') if mode == 'wer' else list(s)
            seqs.append([intern.setdefault(u, len(intern)) for u in units])
    hs, rs = seqs[:B], seqs[B:]
    La, Lb = max(1, max(len(x) for x in hs)), max(1, max(len(x) for x in rs))
    a = torch.zeros((B, La), dtype=torch.int64)
    b_ = torch.zeros((B, Lb), dtype=torch.int64)
    for i in range(B):
        a[i, :len(hs[i])] = torch.tensor(hs[i], dtype=torch.int64)
        b_[i, :len(rs[i])] = torch.tensor(rs[i], dtype=torch.int64)
    a_len = torch.tensor([len(x) for x in hs], dtype=torch.int32)
    b_len = torch.tensor([len(x) for x in rs], dtype=torch.int32)
    d = ops.edit_distance(a.to(dev), a_len.to(dev), b_.to(dev), b_len.to(dev)).cpu().tolist()
    er = [float(d[i]) / len(rs[i]) for i in range(B)]
    return sum(er) / len(er)


def cal_er(tokenizer, pred, truth, mode='wer', ctc=False):
    """Batch error rate (reference: src/util.py:113-127)."""
    if pred is None:
        return np.nan
    elif len(pred.shape) >= 3:
        pred = _argmax(pred)
    if pred.is_cuda:
        return _cal_er_device(tokenizer, pred, truth, mode, ctc)
    # host tensors only occur in CPU-side unit tests of this metric helper
    er = []
    for p, t in zip(pred, truth):
        p = tokenizer.decode(p.tolist(), ignore_repeat=ctc)
        t = tokenizer.decode(t.tolist())
        if mode == 'wer':
            p = p.split(' ')
            t = t.split(' ')
        er.append(float(edit_distance(p, t)) / len(t))
    return sum(er) / len(er)
