"""GPU parity, every gradient, at the FULL size of the two BASELINE configurations that round 3 only covered in
miniature:
  * configs[0]: the architecture the reference ships (config/libri/asr_example.yaml:34-59) - VGG prenet on 120-dim
    fbank+delta+delta-delta (src/module.py:7-66: 3 x 40 -> 1280), 5 x BLSTM-512 each followed by tanh(Linear)
    (src/module.py:125-158), location-aware attention 300 / 201 taps x 10, LSTM-512 decoder, attention only,
    V = 16000 (subword-16k.model), batch 16 (asr_example.yaml:8), T = 800 frames, L = 40;
  * configs[1] "cfg2": 2 x pBLSTM-512 [2,2] concat, CTC only, B = 32, T = 1000, V = 5000 (SURVEY §8 shorthand).
Oracle: oracle/asr_oracle.py on ATen lstm / native conv2d / ctc_loss (host).  Tolerances (north_star): 1e-3 relative
on outputs and losses, 2e-3 per tensor on the input gradient and every parameter gradient.

Gradient conditioning of the shipped architecture: with five BLSTM + tanh-projection layers over 200 frames behind two
ReLU / max-pool stages, back-propagation amplifies f32 rounding by ~1e5 on the EARLY tensors: the oracle's own f32 run
(the reference's arithmetic) differs from its float64 run by 2e-2 on the prenet bias gradients and 4e-1 on
d(audio_feature) (max-abs over max-abs; measured in this container, same seeds) while every output and the loss agree
to 1e-6.  No f32 implementation can be pinned to 2e-3 on those tensors, so the gradient gate is stated against the
float64 truth: per tensor, the HIP path must be within max(2e-3, 3 x the reference-arithmetic f32 oracle's own distance
to float64) - CAPPED at 5e-2 (max-abs over max-abs): however far the f32 oracle strays, the HIP path never gets more
room than that.  The tensors whose uncapped allowance would exceed the cap (d(audio_feature), the prenet biases: the
ill-conditioned ones) are held to two more metrics against float64: cosine >= 0.999, and a relative L2 error no worse
than 3 x the f32 oracle's own (floor 2e-3)."""
import importlib

import pytest
import torch

from conftest import PKG_NAME
from oracle import asr_oracle as O
from oracle.gen_golden import synth_batch, SHIPPED_MODEL, CFG2_MODEL, CNN_MODEL
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL_CAP = 5e-2          # no gradient tensor is ever accepted with a max-abs error above 5 % of its largest element


def _oracle_step(cfg, D, V, B, T, L, seed, dtype=torch.float32):
    feat, feat_len, txt = synth_batch(B, T, D, V, L, seed=seed)
    sd = O.make_state_dict(cfg, D, V, seed=seed + 1)
    sdr = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    fr = feat.to(dtype).clone().requires_grad_(True)
    c, l, a, s, _ = O.asr_forward(sdr, cfg, fr, feat_len, L, teacher=txt, lstm_impl="aten")
    total, _, _ = O.asr_losses(cfg, c, l, a, txt)
    total.backward()
    return dict(feat=feat, feat_len=feat_len, txt=txt, sd=sd, ctc_out=None if c is None else c.detach(),
                enc_len=l, att_out=None if a is None else a.detach(), att_seq=None if s is None else s.detach(),
                total=float(total.detach()), dfeat=fr.grad, grads={k: v.grad for k, v in sdr.items()})


def _device_step(ops, cfg, D, V, L, r):
    asr = importlib.import_module(PKG_NAME + ".src.asr")
    model = asr.ASR(D, V, True, cfg["ctc_weight"], cfg["encoder"], cfg["attention"], cfg["decoder"])
    model.load_state_dict(r["sd"], strict=True)
    model = model.to(DEV).train()
    fg = r["feat"].clone().to(DEV).requires_grad_(True)
    txt = r["txt"].to(DEV)
    txt_len = torch.sum(txt != 0, dim=-1)
    ctc_out, enc_len, att_out, att_seq, _ = model(fg, r["feat_len"].to(DEV), L, tf_rate=1.0, teacher=txt)
    total = 0
    if ctc_out is not None:
        total = total + ops.CTCLoss(blank=0)(ctc_out.transpose(0, 1), txt, enc_len, txt_len) * model.ctc_weight
    if att_out is not None:
        b, t, _ = att_out.shape
        total = total + ops.CrossEntropyLoss(ignore_index=0)(att_out.view(b * t, -1), txt.view(-1)) \
            * (1 - model.ctc_weight)
    total.backward()
    ops.join_deferred()
    ops.check_errors()
    return model, fg, ctc_out, enc_len, att_out, att_seq, total


def _check(model, fg, total, r, r64=None):
    """r: the f32 oracle step (reference arithmetic).  r64: the same step in float64 - when given, gradients are gated
    against IT with the per-tensor tolerance max(2e-3, 3 x |f32 oracle - float64|) (module docstring)."""
    assert abs(total.item() - r["total"]) < 1e-3 * abs(r["total"])
    assert set(n for n, _ in model.named_parameters()) == set(r["grads"])
    bad, report = {}, {}
    items = [("d(audio_feature)", fg.grad.cpu(), r["dfeat"], None if r64 is None else r64["dfeat"])]
    items += [(n, p.grad.cpu(), r["grads"][n], None if r64 is None else r64["grads"][n])
              for n, p in model.named_parameters()]
    for n, got, ref32, ref64 in items:
        ref = ref32 if ref64 is None else ref64
        scale = float(ref.abs().max())
        err = float((got.to(ref.dtype) - ref).abs().max())
        tol = 2e-3
        if ref64 is not None:
            loose = 3.0 * float((ref32.double() - ref64).abs().max()) / max(scale, 1e-30)
            tol = min(max(tol, loose), TOL_CAP)
            if loose > TOL_CAP and scale >= 1e-6:
                # ill-conditioned tensor: direction and energy of the whole gradient against float64 as well
                g, t, o = got.double().reshape(-1), ref64.reshape(-1), ref32.double().reshape(-1)
                cos = float(torch.dot(g, t) / (g.norm() * t.norm()).clamp_min(1e-300))
                l2 = float((g - t).norm() / t.norm().clamp_min(1e-300))
                l2_ref = float((o - t).norm() / t.norm().clamp_min(1e-300))
                report[n + " [cos, relL2, f32-oracle relL2]"] = (cos, l2, l2_ref)
                if cos < 0.999 or l2 > max(3.0 * l2_ref, 2e-3):
                    bad[n + " (cosine / rel-L2 vs float64)"] = (cos, l2, l2_ref)
        report[n] = (err / max(scale, 1e-30), tol)
        if (err > 1e-6) if scale < 1e-6 else (err > tol * scale):
            bad[n] = (err, scale, tol)
    assert not bad, bad
    return report


def test_shipped_architecture_full_size_every_gradient_vs_oracle(ops):
    """config/libri/asr_example.yaml at its own batch size: B=16, T=800 (-> 200 encoder frames), D=120, V=16000, L=40"""
    D, V, B, T, L = 120, 16000, 16, 800, 40
    r = _oracle_step(SHIPPED_MODEL, D, V, B, T, L, seed=51)
    r64 = _oracle_step(SHIPPED_MODEL, D, V, B, T, L, seed=51, dtype=torch.float64)
    model, fg, ctc_out, enc_len, att_out, att_seq, total = _device_step(ops, SHIPPED_MODEL, D, V, L, r)
    assert ctc_out is None and r["ctc_out"] is None                 # ctc_weight 0: no CTC head (src/asr.py:27-30)
    assert torch.equal(enc_len.cpu(), r["enc_len"]) and int(enc_len[0]) == T // 4
    assert att_out.shape == (B, L, V) and att_seq.shape == (B, 1, L, T // 4)
    assert model.encoder.layers[0].out_dim == 1280                  # VGG on 3 x 40 (src/module.py:32-42)
    assert rel_err(att_out.detach().cpu(), r["att_out"]) < 1e-3
    assert rel_err(att_seq.detach().cpu(), r["att_seq"]) < 1e-3
    assert rel_err(att_out.detach().cpu().double(), r64["att_out"]) < 1e-3
    report = _check(model, fg, total, r, r64)
    # the well-conditioned tensors (everything behind the encoder) hold the plain 2e-3 against the f32 oracle as well
    for n, p in model.named_parameters():
        if n.startswith(("decoder.", "attention.", "pre_embed.")) and float(r["grads"][n].abs().max()) > 1e-6:
            assert rel_err(p.grad.cpu(), r["grads"][n]) < 2e-3, n
    print("shipped-architecture gradient report (err vs float64, tolerance | cos, relL2, f32-oracle relL2):",
          {k: tuple("%.2e" % x for x in v) for k, v in report.items() if len(v) == 3 or v[1] > 2e-3 or v[0] > 5e-4})


def test_cnn_prenet_full_size_every_gradient_vs_oracle(ops):
    """the reference's second prenet (src/module.py:68-90: Conv1d(120 -> 512, 4, stride 2, pad 1) twice, no activation)
    in front of the shipped 5 x BLSTM-512 + projection stack at the shipped batch: B=16, T=800 (-> 200 frames), D=120,
    V=16000, L=40 - outputs, loss, input gradient and every parameter gradient (round 4 had this prenet at B=3, T=30
    only).  Same float64-anchored gate as the VGG variant: the BLSTM stack behind the prenet is the same."""
    D, V, B, T, L = 120, 16000, 16, 800, 40
    r = _oracle_step(CNN_MODEL, D, V, B, T, L, seed=71)
    r64 = _oracle_step(CNN_MODEL, D, V, B, T, L, seed=71, dtype=torch.float64)
    model, fg, ctc_out, enc_len, att_out, att_seq, total = _device_step(ops, CNN_MODEL, D, V, L, r)
    assert ctc_out is None and torch.equal(enc_len.cpu(), r["enc_len"]) and int(enc_len[0]) == T // 4
    assert model.encoder.layers[0].out_dim == 512                   # CNNExtractor(out_dim = dim[0]) (src/asr.py:336-340)
    assert att_out.shape == (B, L, V) and att_seq.shape == (B, 1, L, T // 4)
    assert rel_err(att_out.detach().cpu(), r["att_out"]) < 1e-3
    assert rel_err(att_seq.detach().cpu(), r["att_seq"]) < 1e-3
    report = _check(model, fg, total, r, r64)
    for n, p in model.named_parameters():
        if n.startswith(("decoder.", "attention.", "pre_embed.")) and float(r["grads"][n].abs().max()) > 1e-6:
            assert rel_err(p.grad.cpu(), r["grads"][n]) < 2e-3, n
    print("cnn-prenet gradient report (err vs float64, tolerance | cos, relL2, f32-oracle relL2):",
          {k: tuple("%.2e" % x for x in v) for k, v in report.items() if len(v) == 3 or v[1] > 2e-3 or v[0] > 5e-4})


def test_cfg2_full_size_every_gradient_vs_oracle(ops):
    """BASELINE configs[1] at its own size: B=32, T=1000 (-> 250 frames), CTC only, L=64"""
    D, V, B, T, L = 80, 5000, 32, 1000, 64
    r = _oracle_step(CFG2_MODEL, D, V, B, T, L, seed=61)
    model, fg, ctc_out, enc_len, att_out, att_seq, total = _device_step(ops, CFG2_MODEL, D, V, L, r)
    assert att_out is None and att_seq is None                       # ctc_weight 1: no decoder at all (src/asr.py:22-23)
    assert torch.equal(enc_len.cpu(), r["enc_len"]) and ctc_out.shape == (B, T // 4, V)
    assert rel_err(ctc_out.detach().cpu(), r["ctc_out"]) < 1e-3
    _check(model, fg, total, r)
