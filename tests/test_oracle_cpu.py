"""CPU: the oracle restatement against golden vectors produced by the REAL reference
(oracle/gen_golden.py), plus the float64 numpy restatements of ATen lstm / ctc_loss."""
import numpy as np
import pytest
import torch

from oracle import asr_oracle as O
from helpers import CASES, load_golden, golden_state_dict, rel_err

TOL = 1e-4  # oracle vs reference on CPU: same primitives, only op-order noise


@pytest.mark.parametrize("name", list(CASES.keys()))
@pytest.mark.parametrize("impl", ["loop", "aten"])
def test_oracle_forward_backward_matches_reference(name, impl):
    g = load_golden(name)
    cfg = CASES[name][0]
    sd = {k: v.clone().requires_grad_(True) for k, v in golden_state_dict(g).items()}
    feat = torch.from_numpy(g["feat"]).clone().requires_grad_(True)
    feat_len = torch.from_numpy(g["feat_len"])
    txt = torch.from_numpy(g["txt"])
    L = int((txt != 0).sum(-1).max())
    ctc_out, enc_len, att_out, att_seq, _ = O.asr_forward(sd, cfg, feat, feat_len, L, teacher=txt,
                                                          lstm_impl=impl)
    assert np.array_equal(enc_len.numpy(), g["encode_len"])
    if ctc_out is not None:
        assert rel_err(ctc_out.detach(), g["ctc_output"]) < TOL
    if att_out is not None:
        assert rel_err(att_out.detach(), g["att_output"]) < TOL
        assert rel_err(att_seq.detach(), g["att_seq"]) < TOL
    total, ctc_loss, att_loss = O.asr_losses(cfg, ctc_out, enc_len, att_out, txt)
    assert rel_err(total.detach(), g["total_loss"]) < TOL
    total.backward()
    assert rel_err(feat.grad, g["grad_feat"]) < 1e-3
    for k, p in sd.items():
        ref = g["grad." + k]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        if np.max(np.abs(ref)) < 1e-7:
            assert np.max(np.abs(got)) < 1e-6, k
        else:
            assert rel_err(got, ref) < 1e-3, k


@pytest.mark.parametrize("name", ["las_hybrid_loc", "las_att_dot_mh", "las_loc_mh"])
def test_oracle_greedy_matches_reference(name):
    g = load_golden(name)
    cfg = CASES[name][0]
    sd = golden_state_dict(g)
    feat = torch.from_numpy(g["feat"])
    feat_len = torch.from_numpy(g["feat_len"])
    steps = g["greedy_att_output"].shape[1]
    with torch.no_grad():
        _, _, att_out, _, _ = O.asr_forward(sd, cfg, feat, feat_len, steps, teacher=None)
    assert np.array_equal(att_out.argmax(-1).numpy(), g["greedy_att_output"].argmax(-1))
    assert rel_err(att_out, g["greedy_att_output"]) < TOL


def test_ctc_numpy_matches_aten_golden():
    g = load_golden("ctc_loss")
    nll = O.ctc_numpy(g["log_probs"], g["targets"], g["input_lengths"], g["target_lengths"])
    assert rel_err(nll, g["nll"]) < 1e-5
    loss = np.mean(nll / np.maximum(g["target_lengths"], 1))
    assert abs(loss - float(g["loss"])) < 1e-5


def test_lstm_numpy_matches_oracle_lstm():
    gen = torch.Generator().manual_seed(3)
    B, T, D, H = 2, 9, 5, 8
    x = torch.randn(B, T, D, generator=gen)
    sd = {"p.weight_ih_l0": torch.randn(4 * H, D, generator=gen) * 0.3,
          "p.weight_hh_l0": torch.randn(4 * H, H, generator=gen) * 0.3,
          "p.bias_ih_l0": torch.randn(4 * H, generator=gen) * 0.1,
          "p.bias_hh_l0": torch.randn(4 * H, generator=gen) * 0.1}
    for k in list(sd):
        sd[k + "_reverse"] = sd[k].flip(0) * 0.9
    y = O.lstm_layer(x, sd, "p.", True, impl="aten").numpy()
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
    f = O.lstm_numpy(x.numpy(), *[sd["p." + n].numpy() for n in names])
    r = O.lstm_numpy(x.numpy(), *[sd["p." + n + "_reverse"].numpy() for n in names], reverse=True)
    assert rel_err(y[..., :H], f) < 1e-5
    assert rel_err(y[..., H:], r) < 1e-5


@pytest.mark.parametrize("tag,cfg", [("lstm", dict(emb_tying=False, module='LSTM', n_layers=2)),
                                     ("gru", dict(emb_tying=True, module='GRU', n_layers=1))])
def test_lm_oracle_matches_reference(tag, cfg):
    """whole-sequence RNN-LM training step (bin/train_lm.py:62-70): loss + gradients vs the reference"""
    import os
    from helpers import GOLDEN
    g = np.load(os.path.join(GOLDEN, "lm_train.npz"))
    data = torch.from_numpy(g["data"])
    txt = torch.cat((torch.zeros((data.shape[0], 1), dtype=torch.long), data), dim=1)
    sd = {k[len(tag) + 7:]: torch.from_numpy(g[k]).clone().requires_grad_(True)
          for k in g.files if k.startswith(tag + ".param.")}
    pred = O.lm_forward(sd, cfg, txt[:, :-1])
    valid = (txt[:, 1:] != 0)
    assert rel_err(pred.detach()[valid], torch.from_numpy(g[tag + ".pred"])[valid]) < TOL
    loss = torch.nn.functional.cross_entropy(pred.reshape(-1, pred.shape[-1]), txt[:, 1:].reshape(-1), ignore_index=0)
    assert abs(loss.item() - float(g[tag + ".loss"])) < TOL * abs(float(g[tag + ".loss"]))
    loss.backward()
    for k, p in sd.items():
        assert rel_err(p.grad, g["%s.grad.%s" % (tag, k)]) < 1e-3, k
