"""GPU parity of the drop-in model surface (ASR / Encoder / CTCLoss) against golden vectors made
by the REAL reference and against the CPU oracle at BASELINE-sized shapes."""
import importlib

import numpy as np
import pytest
import torch

from conftest import PKG_NAME
from oracle import asr_oracle as O
from helpers import CASES, load_golden, golden_state_dict, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_model(cfg, D, V, adadelta=True):
    asr = importlib.import_module(PKG_NAME + ".src.asr")
    return asr.ASR(D, V, adadelta, cfg["ctc_weight"], cfg["encoder"], cfg["attention"] or {},
                   cfg["decoder"] or {})


def run_train_step(model, ops, cfg, feat, feat_len, txt):
    txt_len = torch.sum(txt != 0, dim=-1)
    ctc_out, enc_len, att_out, att_seq, _ = model(feat, feat_len, int(txt_len.max()), tf_rate=1.0,
                                                  teacher=txt)
    total = 0
    ctc_loss = att_loss = None
    if ctc_out is not None:
        ctc_loss = ops.CTCLoss(blank=0)(ctc_out.transpose(0, 1), txt, enc_len, txt_len)
        total = total + ctc_loss * model.ctc_weight
    if att_out is not None:
        b, t, _ = att_out.shape
        att_loss = ops.CrossEntropyLoss(ignore_index=0)(att_out.view(b * t, -1), txt.view(-1))
        total = total + att_loss * (1 - model.ctc_weight)
    total.backward()
    return ctc_out, enc_len, att_out, att_seq, total


@pytest.mark.parametrize("name", ["las_hybrid_loc", "las_att_dot_mh", "las_gru", "las_loc_mh"])
def test_greedy_decode_matches_reference_golden(ops, name):
    """inference path (no teacher): argmax feedback, src/asr.py:136-142 / bin/test_asr.py:101-121"""
    g = load_golden(name)
    cfg, D, V = CASES[name][0], CASES[name][1], CASES[name][2]
    model = build_model(cfg, D, V)
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(DEV).eval()
    steps = g["greedy_att_output"].shape[1]
    with torch.no_grad():
        _, _, att_out, _, _ = model(torch.from_numpy(g["feat"]).to(DEV),
                                    torch.from_numpy(g["feat_len"]).to(DEV), steps)
    ops.check_errors()
    assert np.array_equal(att_out.argmax(-1).cpu().numpy(), g["greedy_att_output"].argmax(-1))
    assert rel_err(att_out.cpu(), g["greedy_att_output"]) < 1e-3


@pytest.mark.parametrize("name", ["enc_ctc_concat", "enc_ctc_drop_proj", "las_hybrid_loc",
                                  "las_att_dot_mh", "enc_ctc_ln", "enc_vgg_ctc", "enc_cnn_ctc", "las_gru",
                                  "las_loc_mh", "enc_gru_uni_ctc"])
def test_model_matches_reference_golden(ops, name):
    g = load_golden(name)
    cfg, D, V = CASES[name][0], CASES[name][1], CASES[name][2]
    model = build_model(cfg, D, V)
    missing = model.load_state_dict(golden_state_dict(g), strict=True)   # key-for-key compatible
    model = model.to(DEV).train()
    feat = torch.from_numpy(g["feat"]).to(DEV).requires_grad_(True)
    ctc_out, enc_len, att_out, att_seq, total = run_train_step(
        model, ops, cfg, feat, torch.from_numpy(g["feat_len"]).to(DEV), torch.from_numpy(g["txt"]).to(DEV))
    ops.check_errors()
    assert np.array_equal(enc_len.cpu().numpy(), g["encode_len"])
    if ctc_out is not None:
        assert rel_err(ctc_out.detach().cpu(), g["ctc_output"]) < 1e-3
    if att_out is not None:
        assert rel_err(att_out.detach().cpu(), g["att_output"]) < 1e-3
        assert rel_err(att_seq.detach().cpu(), g["att_seq"]) < 1e-3
    assert abs(total.item() - float(g["total_loss"])) < 1e-3 * abs(float(g["total_loss"]))
    assert rel_err(feat.grad.cpu(), g["grad_feat"]) < 1e-3
    for n, p in model.named_parameters():
        ref = g["grad." + n]
        if np.max(np.abs(ref)) < 1e-7:
            continue
        assert rel_err(p.grad.cpu(), ref) < 1e-3, n


def test_cfg2_shapes_vs_oracle(ops):
    """BASELINE cfg2 architecture (2 x pBLSTM-512 concat, CTC-only, V=5000) at B=8, T=200."""
    cfg = dict(ctc_weight=1.0,
               encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[512, 512], dropout=[0, 0],
                            layer_norm=[False, False], proj=[False, False], sample_rate=[2, 2],
                            sample_style='concat'), attention=None, decoder=None)
    D, V, B, T, L = 80, 5000, 8, 200, 20
    from oracle.gen_golden import synth_batch
    feat, feat_len, txt = synth_batch(B, T, D, V, L, seed=3)
    sd = O.make_state_dict(cfg, D, V, seed=1)
    model = build_model(cfg, D, V)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    fg = feat.clone().to(DEV).requires_grad_(True)
    ctc_out, enc_len, _, _, total = run_train_step(model, ops, cfg, fg, feat_len.to(DEV), txt.to(DEV))
    ops.check_errors()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fr = feat.clone().requires_grad_(True)
    c_ref, l_ref, _, _, _ = O.asr_forward(sdr, cfg, fr, feat_len, L, teacher=txt, lstm_impl="aten")
    t_ref, _, _ = O.asr_losses(cfg, c_ref, l_ref, None, txt)
    t_ref.backward()
    assert torch.equal(enc_len.cpu(), l_ref)
    assert rel_err(ctc_out.detach().cpu(), c_ref.detach()) < 1e-3
    assert abs(total.item() - t_ref.item()) < 1e-3 * abs(t_ref.item())
    assert rel_err(fg.grad.cpu(), fr.grad) < 1e-3
    for n, p in model.named_parameters():
        assert rel_err(p.grad.cpu(), sdr[n].grad) < 2e-3, n


@pytest.mark.parametrize("tag,cfg", [("lstm", dict(emb_tying=False, emb_dim=8, module='LSTM', dim=12, n_layers=2, dropout=0.0)),
                                     ("gru", dict(emb_tying=True, emb_dim=12, module='GRU', dim=12, n_layers=1, dropout=0.0))])
def test_rnnlm_training_step_matches_reference_golden(ops, tag, cfg):
    """whole-sequence RNN-LM step of bin/train_lm.py:62-70 (persistent recurrence kernel per layer)"""
    import os
    from helpers import GOLDEN
    lm_mod = importlib.import_module(PKG_NAME + ".src.lm")
    g = np.load(os.path.join(GOLDEN, "lm_train.npz"))
    V = g[tag + ".pred"].shape[-1]
    lm = lm_mod.RNNLM(V, **cfg)
    lm.load_state_dict({k[len(tag) + 7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".param.")},
                       strict=True)
    lm = lm.to(DEV).train()
    data = torch.from_numpy(g["data"])
    txt = torch.cat((torch.zeros((data.shape[0], 1), dtype=torch.long), data), dim=1).to(DEV)
    pred, _ = lm(txt[:, :-1], torch.sum(data != 0, dim=-1))
    loss = ops.CrossEntropyLoss(ignore_index=0)(pred.view(-1, V), txt[:, 1:].reshape(-1))
    loss.backward()
    ops.check_errors()
    valid = (txt[:, 1:] != 0).cpu()
    assert rel_err(pred.detach().cpu()[valid], torch.from_numpy(g[tag + ".pred"])[valid]) < 1e-3
    assert abs(loss.item() - float(g[tag + ".loss"])) < 1e-3 * abs(float(g[tag + ".loss"]))
    for n, p in lm.named_parameters():
        assert rel_err(p.grad.cpu(), g["%s.grad.%s" % (tag, n)]) < 1e-3, n


@pytest.mark.parametrize("name", ["sched_las_hybrid_loc", "sched_las_gru"])
def test_scheduled_sampling_matches_reference_golden(ops, name, monkeypatch):
    """0 < tf_rate < 1 (src/asr.py:119-135): per step a torch.rand(1) decision between the teacher's character
    and one SAMPLED from the model's own softmax.  The golden was produced by the real reference with
    Categorical.sample replaced by an inverse-CDF draw whose uniforms come from the default CPU generator
    (oracle/gen_golden.py::inverse_cdf_sample); the same stand-in here makes both sides take the same decisions
    and the same draws, so outputs, alignments, loss and every gradient must agree."""
    from torch.distributions.categorical import Categorical
    from oracle.gen_golden import SCHED_CASES, inverse_cdf_sample
    g = load_golden(name)
    base = SCHED_CASES[name][0]
    cfg, D, V = CASES[base][0], CASES[base][1], CASES[base][2]
    assert float(g["differs_from_teacher_forcing"]) > 1e-2          # the golden really left the teacher's path
    model = build_model(cfg, D, V)
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(DEV).train()
    feat = torch.from_numpy(g["feat"]).to(DEV).requires_grad_(True)
    txt = torch.from_numpy(g["txt"]).to(DEV)
    txt_len = torch.sum(txt != 0, dim=-1)
    monkeypatch.setattr(Categorical, "sample", inverse_cdf_sample)
    two_pass, inner = [], model._scheduled_sampling_inputs
    model._scheduled_sampling_inputs = lambda *a, **k: (two_pass.append(1), inner(*a, **k))[1]
    torch.manual_seed(int(g["seed"]))
    ctc_out, enc_len, att_out, att_seq, _ = model(feat, torch.from_numpy(g["feat_len"]).to(DEV), int(txt_len.max()),
                                                  tf_rate=float(g["tf_rate"]), teacher=txt)
    b, t, _ = att_out.shape
    total = ops.CrossEntropyLoss(ignore_index=0)(att_out.view(b * t, -1), txt.view(-1)) * (1 - model.ctc_weight)
    if ctc_out is not None:
        total = total + ops.CTCLoss(blank=0)(ctc_out.transpose(0, 1), txt, enc_len, txt_len) * model.ctc_weight
    total.backward()
    ops.join_deferred()
    ops.check_errors()
    # the LSTM decoder with location-aware attention runs the two-pass fused loop, the GRU decoder the per-step kernels
    assert len(two_pass) == (1 if name == "sched_las_hybrid_loc" else 0)
    assert rel_err(att_out.detach().cpu(), g["att_output"]) < 1e-3
    assert rel_err(att_seq.detach().cpu(), g["att_seq"]) < 1e-3
    assert abs(float(total) - float(g["total_loss"])) < 1e-3 * abs(float(g["total_loss"]))
    assert rel_err(feat.grad.cpu(), g["grad_feat"]) < 2e-3
    for n, p in model.named_parameters():
        ref = g["grad." + n]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        if np.abs(ref).max() < 1e-7:
            assert np.abs(got).max() < 1e-6, n
        else:
            assert rel_err(got, ref) < 2e-3, n
