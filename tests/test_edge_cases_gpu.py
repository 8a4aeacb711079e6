"""GPU edge cases the reference's own path admits: single-frame / single-utterance sequences, empty
batches, infeasible CTC alignments (inf loss like torch, zero_infinity=False), zero-length targets,
beam search on a one-frame encoder output."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import PKG_NAME
from oracle import asr_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lstm_params(D, H, g, scale=0.3):
    return tuple((torch.randn(*s, generator=g) * scale) for s in ((4 * H, D), (4 * H, H), (4 * H,), (4 * H,)))


@pytest.mark.parametrize("T,B", [(1, 1), (1, 5), (2, 1)])
def test_lstm_single_step_and_single_utterance(ops, T, B):
    g = torch.Generator().manual_seed(T * 10 + B)
    D, H = 6, 8
    pf, pr = _lstm_params(D, H, g), _lstm_params(D, H, g)
    x = torch.randn(B, T, D, generator=g)
    sd = {}
    for sfx, ps in (('', pf), ('_reverse', pr)):
        for n, p in zip(('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'), ps):
            sd['l.' + n + sfx] = p
    ref = O.lstm_layer(x, sd, 'l.', True)
    y = ops.lstm_layer(x.transpose(0, 1).contiguous().to(DEV), tuple(p.to(DEV) for p in pf),
                       tuple(p.to(DEV) for p in pr))
    ops.check_errors()
    assert rel_err(y.cpu().transpose(0, 1), ref) < 1e-4


def test_empty_batches_are_no_ops(ops):
    z = torch.zeros((0, 16), device=DEV)
    w, b = torch.randn(8, 16, device=DEV), torch.randn(8, device=DEV)
    assert ops.linear(z, w, b).shape == (0, 8)
    assert ops.log_softmax(z).shape == (0, 16)
    assert ops.tanh(z).shape == (0, 16)
    v, i = ops.topk(torch.zeros((0, 9), device=DEV), 3)
    assert v.shape == (0, 3) and i.shape == (0, 3)


def test_ctc_infeasible_and_empty_targets_match_torch(ops):
    """T' < needed frames -> +inf loss (zero_infinity=False, bin/train_asr.py:49); an all-blank target
    (length 0) scores the blank path"""
    g = torch.Generator().manual_seed(3)
    T, B, V = 5, 4, 7
    lp = torch.randn(T, B, V, generator=g).log_softmax(-1)
    tgt = torch.tensor([[3, 3, 4, 5, 2, 2], [1, 2, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [4, 0, 0, 0, 0, 0]])
    tl = torch.tensor([6, 2, 0, 1])            # row 0 needs 6 + 2 repeats = 8 frames > 5: infeasible
    il = torch.tensor([5, 5, 4, 1])
    ref = F.ctc_loss(lp, tgt, il, tl, blank=0, reduction='none', zero_infinity=False)
    got = ops.CTCLossFn.apply(lp.to(DEV), tgt.to(DEV), il.to(DEV), tl.to(DEV), 0, 'none').cpu()
    assert torch.isinf(ref[0]) and torch.isinf(got[0])
    assert torch.allclose(got[1:], ref[1:], rtol=1e-4, atol=1e-5)
    # finite rows still get correct gradients when another row is infeasible
    lpd = lp.clone().to(DEV).requires_grad_(True)
    lpr = lp.clone().requires_grad_(True)
    ops.CTCLossFn.apply(lpd, tgt.to(DEV), il.to(DEV), tl.to(DEV), 0, 'none')[1:].sum().backward()
    F.ctc_loss(lpr, tgt, il, tl, blank=0, reduction='none')[1:].sum().backward()
    assert rel_err(lpd.grad.cpu()[:, 1:], lpr.grad[:, 1:]) < 1e-3


def test_model_on_minimum_length_input(ops):
    """4 input frames through a x4 pyramid leave ONE encoder frame: CTC + attention still run"""
    asr = importlib.import_module(PKG_NAME + ".src.asr")
    decode = importlib.import_module(PKG_NAME + ".src.decode")
    torch.manual_seed(1)
    model = asr.ASR(8, 12, True, 0.5,
                    dict(prenet='', module='LSTM', bidirection=True, dim=[8, 8], dropout=[0, 0],
                         layer_norm=[False, False], proj=[False, False], sample_rate=[2, 2], sample_style='drop'),
                    dict(mode='loc', dim=6, num_head=1, v_proj=False, temperature=1.0, loc_kernel_size=2,
                         loc_kernel_num=2),
                    dict(module='LSTM', dim=8, layer=1, dropout=0)).to(DEV)
    feat = torch.randn(1, 4, 8, device=DEV)
    flen = torch.tensor([4], device=DEV)
    txt = torch.tensor([[5, 1]], device=DEV)
    ctc_out, enc_len, att_out, att_seq, _ = model(feat, flen, 2, tf_rate=1.0, teacher=txt)
    assert ctc_out.shape == (1, 1, 12) and enc_len.tolist() == [1] and att_out.shape == (1, 2, 12)
    loss = ops.CrossEntropyLoss(ignore_index=0)(att_out.view(2, -1), txt.view(-1))
    loss.backward()
    ops.check_errors()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    model.eval()
    hyps = decode.BeamDecoder(model, None, beam_size=3, min_len_ratio=0.0, max_len_ratio=1.0, ctc_weight=0.3)(feat, flen)
    assert 1 <= len(hyps) <= 3 and all(len(h.outIndex) >= 1 for h in hyps)


@pytest.mark.parametrize("Lmax", [300, 700, 1023])
def test_ctc_long_character_transcripts_match_torch(ops, Lmax):
    """character-level transcripts of the longest utterances: padded target width up to 1023
    (S = 2L+1 <= 2048 lattice states, 16 / 32 states per lane)"""
    g = torch.Generator().manual_seed(Lmax)
    B, V = 3, 31
    T = int(1.3 * Lmax) + 64          # ~1.3 encoder frames per character, as in read speech
    lp = torch.randn(T, B, V, generator=g).log_softmax(-1)
    tl = torch.tensor([Lmax, Lmax // 2, 5])
    il = torch.tensor([T, T - 7, T // 2])
    tgt = torch.zeros(B, Lmax, dtype=torch.long)
    for b in range(B):
        tgt[b, :tl[b]] = torch.randint(1, V, (int(tl[b]),), generator=g)
    lpr = lp.double().requires_grad_(True)      # float64 ATen as the yardstick for these long chains
    ref = F.ctc_loss(lpr, tgt, il, tl, blank=0, reduction='none')
    ref.sum().backward()
    lpd = lp.clone().to(DEV).requires_grad_(True)
    got = ops.CTCLossFn.apply(lpd, tgt.to(DEV), il.to(DEV), tl.to(DEV), 0, 'none')
    got.sum().backward()
    assert torch.allclose(got.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-3)
    # the f32 log-domain lattices hold values of magnitude ~nll (here 2000-4500 nats): one ulp there is
    # 1.2e-4..4.9e-4, and the gradient exponentiates alpha + beta + nll - lp, so a few ulps of lattice
    # rounding are a few 1e-3 RELATIVE in the gradient for any f32 implementation (ATen's f32 CPU kernel
    # differs from its own f64 by the same amount); 1e-3 holds up to nll ~ 1000 (L = 300 here)
    tol = 1e-3 if float(ref.max()) < 1500 else 6e-3
    assert rel_err(lpd.grad.cpu(), lpr.grad) < tol
    with pytest.raises(Exception):      # beyond the supported width: loud, never silent
        ops.CTCLossFn.apply(lp.to(DEV), torch.zeros(B, 1024, dtype=torch.long, device=DEV), il.to(DEV),
                            tl.to(DEV), 0, 'none')


def test_ctc_out_of_range_label_is_flagged_not_read(ops):
    """torch.nn.CTCLoss raises on a label outside [0,V); the device path never reads out of bounds and
    turns that utterance's loss into NaN (which the solver's NaN guard skips), others unaffected"""
    g = torch.Generator().manual_seed(8)
    T, B, V = 20, 3, 9
    lp = torch.randn(T, B, V, generator=g).log_softmax(-1)
    tgt = torch.tensor([[3, 4, 5], [2, V + 100, 1], [1, -3, 0]])
    tl, il = torch.tensor([3, 3, 2]), torch.tensor([T, T, T])
    got = ops.CTCLossFn.apply(lp.to(DEV), tgt.to(DEV), il.to(DEV), tl.to(DEV), 0, 'none').cpu()
    ref0 = F.ctc_loss(lp[:, :1], tgt[:1], il[:1], tl[:1], blank=0, reduction='none')
    assert torch.allclose(got[:1], ref0, rtol=1e-4, atol=1e-5)
    assert torch.isnan(got[1]) and torch.isnan(got[2])


@pytest.mark.parametrize("H", [512, 1024])
def test_recurrence_with_genuine_nans_does_not_hang(ops, H):
    """a diverged run feeds NaNs into the persistent recurrence kernels, whose hand-off uses a NaN bit
    pattern as the 'not written yet' sentinel: a genuine NaN must flow through (slow path: reload and
    compare against the sentinel's exact bit pattern), not spin until the 3 s timeout - the solver then
    sees a NaN gradient norm and skips the step like the reference (src/solver.py:85-89)"""
    import time
    g = torch.Generator().manual_seed(1)
    T, B, D = 24, 32, 64
    x = torch.randn(T, B, D, generator=g)
    x[5, 3, 7] = float("nan")
    ps = [torch.randn(4 * H, D, generator=g) / D ** 0.5, torch.randn(4 * H, H, generator=g) / H ** 0.5,
          torch.zeros(4 * H), torch.zeros(4 * H)]
    pf = tuple(p.clone().to(DEV).requires_grad_(True) for p in ps)
    pr = tuple((p * 0.9).to(DEV).requires_grad_(True) for p in ps)
    xd = x.to(DEV).requires_grad_(True)
    torch.cuda.synchronize()
    t0 = time.time()
    y = ops.lstm_layer(xd, pf, pr)
    y.sum().backward()
    ops.check_errors()                          # raises on ASRK_ETIMEOUT
    torch.cuda.synchronize()
    assert time.time() - t0 < 2.0
    yc = y.detach().cpu()
    assert torch.isnan(yc[5:, 3, :H]).all() and torch.isnan(yc[:6, 3, H:]).all()     # forward / reverse directions
    assert torch.isfinite(yc[:, 4]).all()                                            # other utterances untouched
    assert torch.isnan(pf[1].grad).any()


def test_round6_entries_refuse_what_they_cannot_do(ops):
    """argument / shape / workspace errors of the round-6 entries (implicit-GEMM convolutions, device beam bookkeeping):
    a code, never a launch that reads out of bounds"""
    import ctypes
    from conftest import PKG_NAME
    import importlib
    L = importlib.import_module(PKG_NAME + "._lib").load()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    EINVAL, ESHAPE, EWS = -1, -2, -3
    assert L.asrk_conv3x3_supported(400, 20, 64, 128) == 1 and L.asrk_conv3x3_supported(400, 20, 64, 96) == 0
    assert L.asrk_conv3x3_supported(400, 129, 64, 64) == 0 and L.asrk_conv3x3_supported(400, 20, 3, 64) == 0
    assert L.asrk_conv3x3_first_supported(800, 40, 3, 64) == 1 and L.asrk_conv3x3_first_supported(800, 40, 4, 64) == 0
    x = torch.zeros(1 * 4 * 8 * 64, device=DEV)
    w = torch.zeros(64 * 64 * 9, device=DEV)
    y = torch.zeros(1 * 4 * 8 * 64, device=DEV)
    assert L.asrk_conv3x3_f32(p(x), None, p(w), None, p(y), 1, 4, 8, 64, 96, 0, None) == ESHAPE
    assert L.asrk_conv3x3_f32(p(x), None, p(w), None, p(y), 1, 4, 8, 0, 64, 0, None) == EINVAL
    assert L.asrk_conv3x3_f32(p(x), None, None, None, p(y), 1, 4, 8, 64, 64, 0, None) == EINVAL
    assert L.asrk_conv3x3_f32(None, None, None, None, None, 0, 4, 8, 64, 64, 0, None) == 0          # empty batch
    need = L.asrk_conv3x3_wgrad_ws_bytes(1, 4, 8, 64, 64)
    assert need > 0
    ws = torch.zeros(need // 4, device=DEV)
    dw, db = torch.zeros(64 * 64 * 9, device=DEV), torch.zeros(64, device=DEV)
    assert L.asrk_conv3x3_wgrad_f32(p(x), p(y), None, p(dw), p(db), 1, 4, 8, 64, 64, p(ws), need - 4, None) == EWS
    assert L.asrk_conv3x3_wgrad_f32(p(x), p(y), None, p(dw), p(db), 1, 4, 8, 64, 64, None, 0, None) == EWS
    assert L.asrk_conv3x3_wgrad_f32(p(x), p(y), None, p(dw), p(db), 1, 4, 8, 64, 64, p(ws), need, None) == 0
    dw.fill_(1.0)
    assert L.asrk_conv3x3_wgrad_f32(None, None, None, p(dw), p(db), 0, 4, 8, 64, 64, None, 0, None) == 0   # empty batch:
    torch.cuda.synchronize()
    assert float(dw.abs().max()) == 0.0                                                                  # zero gradients
    assert L.asrk_conv3x3_first_f32(p(x), p(w), None, p(y), 1, 4, 8, 4, 64, 256, 64, 1, 8, 0, None) == ESHAPE
    # beam bookkeeping: beam / candidate limits, position outside the history
    z = torch.zeros(4096, device=DEV)
    zi = torch.zeros(4096, dtype=torch.int64, device=DEV)
    args = lambda B, C, t, lmax: (p(z), p(zi), p(z), p(zi), 1, B, C, t, lmax, 8) + (p(zi),) * 19 + (None,)
    assert L.asrk_beam_select_f32(*args(33, 0, 0, 4)) == ESHAPE
    assert L.asrk_beam_select_f32(*args(4, 49, 0, 4)) == ESHAPE
    assert L.asrk_beam_select_f32(*args(4, 6, 4, 4)) == EINVAL
    ops.check_errors()
