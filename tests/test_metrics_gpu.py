"""GPU: validation read-out kernels (csrc/metrics.hip) - hypothesis crop / CTC collapse and batched Levenshtein
distance - against the host rules of src/text.py / src/util.py and the REFERENCE's cal_er values
(tests/golden/host.json, produced by running /root/reference's src/util.py:113-127).  Integer work: exact."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import PKG_NAME, GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mod(name):
    return importlib.import_module(PKG_NAME + "." + name)


def _host_crop(ids, pad, eos, ignore_repeat):
    out, prev = [], None
    for t, i in enumerate(ids):
        if i == eos:
            break
        if i != pad and not (ignore_repeat and t > 0 and i == prev):
            out.append(i)
        prev = i
    return out


@pytest.mark.parametrize("B,T,V", [(1, 1, 4), (7, 63, 5), (32, 200, 6), (5, 64, 3), (3, 65, 40), (33, 1000, 8)])
@pytest.mark.parametrize("ignore_repeat", [False, True])
def test_token_crop_matches_text_encoder_rule(ops, B, T, V, ignore_repeat):
    g = torch.Generator().manual_seed(B * 1000 + T)
    ids = torch.randint(0, V, (B, T), generator=g)
    ids[0] = 0                                  # an all-pad row
    if B > 1:
        ids[1] = 2
        ids[1, T // 2:] = 1                     # eos in the middle, repeats before it
    if B > 2:
        ids[2, 0] = 1                           # eos first: empty hypothesis
    out, n = ops.token_crop(ids.to(DEV), 0, 1, ignore_repeat)
    out, n = out.cpu(), n.cpu().tolist()
    for b in range(B):
        want = _host_crop(ids[b].tolist(), 0, 1, ignore_repeat)
        assert n[b] == len(want) and out[b, :n[b]].tolist() == want, (b, want)


def test_token_crop_strided_view(ops):
    """rows that are a view (the arg-max of [B,T,V] logits sliced): stride(0) != T"""
    g = torch.Generator().manual_seed(3)
    big = torch.randint(0, 5, (6, 90), generator=g).to(DEV)
    view = big[:, :70]
    out, n = ops.token_crop(view, 0, 1, True)
    for b in range(6):
        want = _host_crop(view[b].tolist(), 0, 1, True)
        assert out[b, :int(n[b])].tolist() == want


@pytest.mark.parametrize("B,La,Lb,V", [(1, 0, 0, 3), (4, 1, 1, 2), (9, 13, 70, 4), (32, 64, 64, 3), (5, 65, 129, 6),
                                       (8, 300, 257, 30), (2, 5, 1500, 3)])
def test_edit_distance_matches_host_levenshtein(ops, B, La, Lb, V):
    util = _mod("src.util")
    g = torch.Generator().manual_seed(B + La * 7 + Lb)
    a = torch.randint(0, V, (B, max(La, 1)), generator=g)
    b = torch.randint(0, V, (B, max(Lb, 1)), generator=g)
    al = torch.randint(0, La + 1, (B,), generator=g)
    bl = torch.randint(0, Lb + 1, (B,), generator=g)
    al[0], bl[0] = La, Lb
    if B > 1:
        al[1] = 0
    if B > 2:
        bl[2] = 0
    if B > 3:                                     # identical sequences -> 0
        a[3, :min(La, Lb)] = b[3, :min(La, Lb)]
        al[3] = bl[3] = min(La, Lb)
    d = ops.edit_distance(a.to(DEV), al.to(DEV), b.to(DEV), bl.to(DEV)).cpu().tolist()
    for i in range(B):
        want = util.edit_distance(a[i, :int(al[i])].tolist(), b[i, :int(bl[i])].tolist())
        assert d[i] == want, (i, int(al[i]), int(bl[i]))


def test_cal_er_on_device_equals_reference_values(ops, tmp_path):
    """the reference's own cal_er outputs (host.json) through the device path: wer / cer, with and without CTC
    repeat merging, 3-D logits and 2-D ids"""
    gold = json.load(open(os.path.join(GOLDEN, "host.json")))
    util, text = _mod("src.util"), _mod("src.text")
    vf = str(tmp_path / "char.txt")
    with open(vf, "w") as f:
        f.write(gold["text.char_vocab"])
    enc = text.load_text_encoder("character", vf)
    logits = torch.tensor(gold["util.cal_er.logits"]).view(3, 9, enc.vocab_size)
    truth = torch.tensor(gold["util.cal_er.truth"])
    for mode in ("wer", "cer"):
        for ctc in (False, True):
            for tag, pred in (("3d", logits), ("2d", logits.argmax(-1))):
                got = util.cal_er(enc, pred.to(DEV), truth.to(DEV), mode=mode, ctc=ctc)
                assert abs(got - gold["util.cal_er.%s.ctc%d.%s" % (mode, int(ctc), tag)]) < 1e-12, (mode, ctc, tag)


def test_cal_er_device_equals_host_on_subword_batches(ops):
    """random hypotheses over the sentencepiece fixture: device path == host path, WER and CER"""
    util, text = _mod("src.util"), _mod("src.text")
    enc = text.load_text_encoder("subword", os.path.join(GOLDEN, "spm_tiny.model"))
    g = torch.Generator().manual_seed(17)
    V = enc.vocab_size
    pred = torch.randint(0, V, (16, 40), generator=g)
    truth = torch.randint(3, V, (16, 25), generator=g)
    truth[:, -1] = 1
    truth[3, 10:] = 0
    truth[3, 9] = 1
    for mode in ("wer", "cer"):
        for ctc in (False, True):
            host = util.cal_er(enc, pred, truth, mode=mode, ctc=ctc)
            dev = util.cal_er(enc, pred.to(DEV), truth.to(DEV), mode=mode, ctc=ctc)
            assert abs(host - dev) < 1e-12, (mode, ctc, host, dev)
