"""GPU parity of the decode path (config 5): device CTC prefix scorer, batched joint
CTC-attention(+RNN-LM) beam search and pure-CTC beam search vs hypotheses produced by the REAL
reference decoders (tests/golden/decode.npz)."""
import importlib

import numpy as np
import pytest
import torch
import yaml

from conftest import PKG_NAME
from oracle import decode_oracle as DO
from helpers import CASES, load_golden, golden_state_dict, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mod(name):
    return importlib.import_module(PKG_NAME + "." + name)


def test_prefix_score_kernel_vs_reference(ops):
    g = load_golden("decode")
    ctc = _mod("src.ctc")
    ps = ctc.CTCPrefixScore(torch.from_numpy(g["ps_x"]).to(DEV))
    r0 = ps.init_state()
    assert np.allclose(r0, g["ps_r0"], rtol=1e-6, atol=1e-6)
    psi1, r1 = ps.cheap_compute([], r0, [3, 1, 5, 8])
    assert rel_err(psi1, g["ps_psi1"]) < 1e-5 and rel_err(r1, g["ps_r1"]) < 1e-5
    psi2, r2 = ps.cheap_compute([3], r1[0], [3, 4, 1, 2])
    assert rel_err(psi2, g["ps_psi2"]) < 1e-5 and rel_err(r2, g["ps_r2"]) < 1e-5
    psi3, r3 = ps.cheap_compute([3, 3], r2[0], [1, 7, 3])
    assert rel_err(psi3, g["ps_psi3"]) < 1e-5 and rel_err(r3, g["ps_r3"]) < 1e-5


def test_prefix_score_full_compute_vs_reference(ops):
    """CTCPrefixScore.full_compute (all tokens, no <eos> override) chained over four prefixes"""
    g = load_golden("prefix_full")
    ps = _mod("src.ctc").CTCPrefixScore(torch.from_numpy(g["x"]).to(DEV))
    r = ps.init_state()
    for k, (prefix, pick) in enumerate([([], 3), ([3], 3), ([3, 3], 7), ([3, 3, 7], None)], 1):
        psi, rn = ps.full_compute(prefix, r)
        assert rel_err(psi, g["psi%d" % k]) < 1e-5 and rel_err(rn, g["r%d" % k]) < 1e-5, k
        if pick is not None:
            r = rn[pick]


def test_prefix_score_kernel_batched_vs_oracle(ops):
    rng = np.random.RandomState(1)
    T, V, n, C = 200, 500, 16, 24
    x = torch.from_numpy(rng.randn(1, T, V).astype(np.float32)).log_softmax(-1)
    ctc = _mod("src.ctc")
    ps = ctc.CTCPrefixScore(x.to(DEV))
    xs = x[0].numpy()
    r0 = DO.init_state(xs)
    # build n different prefixes of length 2 by chaining the oracle
    prefixes, states = [], []
    for h in range(n):
        a, b = int(rng.randint(2, V)), int(rng.randint(2, V))
        if h % 4 == 0:
            b = a                                      # repeated last token
        _, r1 = DO.prefix_scores(xs, [], r0, [a])
        _, r2 = DO.prefix_scores(xs, [a], r1[0], [b])
        prefixes.append([a, b]); states.append(r2[0])
    cands = rng.randint(1, V, size=(n, C)).astype(np.int32)
    cands[:, 0] = 1                                    # <eos> among the candidates
    cands[::4, 1] = [p[-1] for p in prefixes[::4]]     # last token among the candidates
    psi, r = ps.cheap_compute_batch([2] * n, [p[-1] for p in prefixes],
                                    torch.from_numpy(np.stack(states)).to(DEV), torch.from_numpy(cands))
    for h in range(n):
        pr, rr = DO.prefix_scores(xs, prefixes[h], states[h], list(cands[h]))
        assert np.allclose(psi[h].cpu().numpy(), pr, rtol=1e-4, atol=1e-3)
        assert np.allclose(r[h].cpu().numpy(), rr, rtol=1e-4, atol=1e-3)


def _asr(name):
    g = load_golden(name)
    cfg, D, V = CASES[name][0], CASES[name][1], CASES[name][2]
    model = _mod("src.asr").ASR(D, V, True, cfg["ctc_weight"], cfg["encoder"], cfg["attention"] or {},
                                cfg["decoder"] or {})
    model.load_state_dict(golden_state_dict(g), strict=True)
    feat = torch.from_numpy(g["feat"])[:1].to(DEV)
    flen = torch.from_numpy(g["feat_len"])[:1].to(DEV)
    return model.to(DEV).eval(), feat, flen, V


@pytest.mark.parametrize("tag,kw,use_lm", [
    ("beam_ctc", dict(beam_size=3, ctc_weight=0.4), False),
    ("beam_att", dict(beam_size=4, ctc_weight=0.0), False),
    ("beam_ctc_lm", dict(beam_size=3, ctc_weight=0.4, lm_weight=0.3), True),
])
def test_beam_decoder_matches_reference(ops, tmp_path, tag, kw, use_lm):
    g = load_golden("decode")
    model, feat, flen, V = _asr("las_hybrid_loc")
    if use_lm:
        lm_cfg = dict(emb_tying=False, emb_dim=10, module="LSTM", dim=14, n_layers=2, dropout=0.0)
        yaml.safe_dump({"model": lm_cfg}, open(tmp_path / "lm.yaml", "w"))
        sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("lm.")}
        torch.save({"model": sd}, tmp_path / "lm.pth")       # reference-layout LM checkpoint
        kw = dict(kw, lm_path=str(tmp_path / "lm.pth"), lm_config=str(tmp_path / "lm.yaml"))
    dec = _mod("src.decode").BeamDecoder(model, None, min_len_ratio=0.01, max_len_ratio=0.5, **kw)
    hyps = dec(feat, flen)
    ops.check_errors()
    assert len(hyps) == int(g[tag + ".n"])
    for i, h in enumerate(hyps):
        assert h.outIndex == g["%s.hyp%d" % (tag, i)].tolist(), (tag, i)
        ref = g["%s.score%d" % (tag, i)]
        assert np.allclose(np.asarray(h.output_scores, np.float32), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("name,kw", [
    ("las_loc_mh", dict(beam_size=3, ctc_weight=0.3)),     # 2-head location-aware attention, 2-layer decoder
    ("las_loc_mh", dict(beam_size=4, ctc_weight=0.0)),
    ("las_gru", dict(beam_size=3, ctc_weight=0.0)),        # GRU encoder + 2-layer GRU decoder
])
def test_beam_decoder_more_models(ops, name, kw):
    """tests/golden/decode_more.npz (oracle/gen_golden.py --decode-more): second utterance of the case"""
    g = load_golden("decode_more")
    gm = load_golden(name)
    cfg, D, V = CASES[name][0], CASES[name][1], CASES[name][2]
    model = _mod("src.asr").ASR(D, V, True, cfg["ctc_weight"], cfg["encoder"], cfg["attention"] or {},
                                cfg["decoder"] or {})
    model.load_state_dict(golden_state_dict(gm), strict=True)
    model = model.to(DEV).eval()
    feat = torch.from_numpy(gm["feat"])[1:2].to(DEV)
    flen = torch.from_numpy(gm["feat_len"])[1:2].to(DEV)
    tag = "%s.b%d.w%d" % (name, kw["beam_size"], int(10 * kw["ctc_weight"]))
    dec = _mod("src.decode").BeamDecoder(model, None, min_len_ratio=0.01, max_len_ratio=0.6, **kw)
    hyps = dec(feat, flen)
    ops.check_errors()
    assert len(hyps) == int(g[tag + ".n"])
    for i, h in enumerate(hyps):
        assert h.outIndex == g["%s.hyp%d" % (tag, i)].tolist(), (tag, i)
        ref = g["%s.score%d" % (tag, i)]
        assert np.allclose(np.asarray(h.output_scores, np.float32), ref, rtol=2e-3, atol=2e-3)


def test_ctc_beam_decoder_matches_reference(ops):
    g = load_golden("decode")
    model, feat, flen, V = _asr("enc_ctc_concat")
    dec = _mod("src.ctc").CTCBeamDecoder(model, [1] + list(range(3, V)), beam_size=3, vocab_candidate=4)
    hyps = dec(feat, flen)
    assert len(hyps) == int(g["ctcbeam.n"])
    for i, y in enumerate(hyps):
        assert list(y) == g["ctcbeam.hyp%d" % i].tolist()


@pytest.mark.parametrize("T,use_lm", [(800, True), (800, False), (1600, True), (1600, False)])
def test_beam_decoder_at_cfg5_widths_matches_reference(ops, tmp_path, T, use_lm):
    """BASELINE configs[4]: full LAS widths (4 x pBLSTM-1024, location-aware attention 300 / 201 taps x
    10 kernels, LSTM-1024 decoder, V=5000), beam 16, CTC weight 0.5 (24 candidates), 2 x LSTM-1024 RNN-LM
    weight 0.5, max_len_ratio 0.07 - all 16 hypotheses and their per-token scores equal what the REAL
    reference BeamDecoder produced on the same seeded weights (tests/golden/decode_cfg5.npz,
    oracle/gen_golden.py --decode-cfg5)."""
    from oracle.gen_golden import CFG3_MODEL, CFG5_LM, CFG5_DECODE, cfg5_weights, cfg5_utterance
    g = load_golden("decode_cfg5")
    sd, lm_sd = cfg5_weights()
    model = _mod("src.asr").ASR(80, 5000, True, CFG3_MODEL["ctc_weight"], CFG3_MODEL["encoder"],
                                CFG3_MODEL["attention"], CFG3_MODEL["decoder"])
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    kw = dict(CFG5_DECODE)
    if use_lm:
        yaml.safe_dump({"model": CFG5_LM}, open(tmp_path / "lm.yaml", "w"))
        torch.save({"model": lm_sd}, tmp_path / "lm.pth")
        kw.update(lm_path=str(tmp_path / "lm.pth"), lm_config=str(tmp_path / "lm.yaml"))
    else:
        kw["lm_weight"] = 0.0
    feat, flen = cfg5_utterance(T)
    dec = _mod("src.decode").BeamDecoder(model, None, **kw)
    hyps = dec(feat.to(DEV), flen.to(DEV))
    ops.check_errors()
    tag = "T%d.%s" % (T, "lm" if use_lm else "nolm")
    assert len(hyps) == int(g[tag + ".n"]) == 16
    for i, h in enumerate(hyps):
        assert h.outIndex == g["%s.hyp%d" % (tag, i)].tolist(), (tag, i)
        ref = g["%s.score%d" % (tag, i)]
        assert np.allclose(np.asarray(h.output_scores, np.float32), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("tag,lm_cfg", [
    ("lstm", dict(emb_tying=False, emb_dim=6, module='LSTM', dim=9, n_layers=1, dropout=0.0)),
    ("gru", dict(emb_tying=True, emb_dim=8, module='GRU', dim=8, n_layers=2, dropout=0.0)),
])
def test_ctc_beam_decoder_with_lm_matches_reference(ops, tmp_path, tag, lm_cfg):
    """pure-CTC prefix beam search WITH RNN-LM fusion (src/ctc.py:241-352, lm_weight > 0): the candidate
    ranking runs as a batched device top-k over ctc + w*lm; hypotheses equal the real reference's
    (tests/golden/ctcbeam_lm.npz, oracle/gen_golden.py --ctc-lm)"""
    g = load_golden("ctcbeam_lm")
    gm = load_golden("enc_ctc_concat")
    cfg, D, V = CASES["enc_ctc_concat"][0], CASES["enc_ctc_concat"][1], CASES["enc_ctc_concat"][2]
    model = _mod("src.asr").ASR(D, V, True, cfg["ctc_weight"], cfg["encoder"], {}, {})
    model.load_state_dict(golden_state_dict(gm), strict=True)
    model = model.to(DEV).eval()
    yaml.safe_dump({"model": lm_cfg}, open(tmp_path / "lm.yaml", "w"))
    pre = tag + ".lm."
    torch.save({"model": {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}},
               tmp_path / "lm.pth")
    dec = _mod("src.ctc").CTCBeamDecoder(model, [1] + list(range(3, V)), beam_size=4, vocab_candidate=5,
                                         lm_path=str(tmp_path / "lm.pth"), lm_config=str(tmp_path / "lm.yaml"),
                                         lm_weight=0.6, device=DEV)
    for u in (0, 1, 2):
        feat = torch.from_numpy(gm["feat"])[u:u + 1].to(DEV)
        flen = torch.from_numpy(gm["feat_len"])[u:u + 1].to(DEV)
        hyps = dec(feat, flen)
        assert len(hyps) == int(g["%s.u%d.n" % (tag, u)])
        for i, y in enumerate(hyps):
            assert list(y) == g["%s.u%d.hyp%d" % (tag, u, i)].tolist(), (tag, u, i)


# ------------------------------------------------------------------------------ device prefix beam (§8 f4)
class _StubASR:
    """CTCBeamDecoder only needs these of its acoustic model when the search is driven directly"""
    enable_ctc = True

    def __init__(self, V):
        self.vocab_size = V


@pytest.mark.parametrize("name", ["collide", "wide", "eos", "lm", "lm_gru"])
def test_device_prefix_beam_equals_real_reference(ops, tmp_path, name):
    """the gfx950 prefix-beam kernel (csrc/prefix_beam.hip) on the SAME log-probabilities the real reference
    CTCBeamDecoder searched (tests/golden/ctcbeam_big.npz, oracle/gen_golden.py --ctc-beam-big): ambiguous
    decimal-string sort keys, beam 20 x 30 candidates over V = 5000 for 120 frames in ONE launch, finished
    hypotheses, LSTM / tied-GRU LM fusion with the LM stepped on the device between per-frame launches -
    every surviving hypothesis, in order"""
    from oracle.gen_golden import CTC_BEAM_BIG, ctc_beam_big_logits
    V, T, beam, cand, seed, hot, lm_cfg, lm_w = CTC_BEAM_BIG[name]
    g = load_golden("ctcbeam_big")
    x = torch.log_softmax(ctc_beam_big_logits(name), dim=-1)           # what src/ctc.py:250 hands the loop
    kw = {}
    if lm_cfg is not None:
        pre = name + ".lm."
        yaml.safe_dump({"model": lm_cfg}, open(tmp_path / "lm.yaml", "w"))
        torch.save({"model": {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}},
                   tmp_path / "lm.pth")
        kw = dict(lm_path=str(tmp_path / "lm.pth"), lm_config=str(tmp_path / "lm.yaml"), lm_weight=lm_w, device=DEV)
    dec = _mod("src.ctc").CTCBeamDecoder(_StubASR(V), [1] + list(range(3, V)), beam, cand, **kw)
    assert dec._device_search_ok(V)
    hyps = dec.search_device(x.to(DEV).contiguous())
    want = [g["%s.hyp%d" % (name, i)].tolist() for i in range(int(g[name + ".n"]))]
    assert hyps == want


def test_device_prefix_beam_all_blank_utterance(ops):
    """every frame's arg-max is blank: the reference skips all frames and returns the single empty hypothesis"""
    V, T = 50, 12
    x = torch.full((T, V), -8.0)
    x[:, 0] = 5.0
    dec = _mod("src.ctc").CTCBeamDecoder(_StubASR(V), [1] + list(range(3, V)), 4, 5)
    assert dec.search_device(torch.log_softmax(x, -1).to(DEV).contiguous()) == [[]]


def test_ctc_beam_host_bookkeeping_path_still_matches_reference(ops, monkeypatch):
    """configurations outside the kernel's limits fall back to host bookkeeping (ASRK_CTC_BEAM_DEVICE=0 forces
    it): same hypotheses on the toy golden"""
    monkeypatch.setenv("ASRK_CTC_BEAM_DEVICE", "0")
    g = load_golden("decode")
    model, feat, flen, V = _asr("enc_ctc_concat")
    dec = _mod("src.ctc").CTCBeamDecoder(model, [1] + list(range(3, V)), beam_size=3, vocab_candidate=4)
    hyps = dec(feat, flen)
    assert [list(y) for y in hyps] == [g["ctcbeam.hyp%d" % i].tolist() for i in range(int(g["ctcbeam.n"]))]


@pytest.mark.parametrize("V", [6000, 16000])
def test_device_prefix_beam_large_vocabulary_vs_oracle(ops, V):
    """vocabularies beyond the one-wave-per-row ranking (V > 5120: the workgroup-wide ranking from an LDS score
    row; 16000 = the reference's subword-16k model): hypotheses equal the CPU oracle's (the restated reference loop)"""
    from oracle import ctc_beam_oracle as CBO
    g = torch.Generator().manual_seed(V)
    T = 25
    logits = torch.randn(T, V, generator=g) * 2.0
    logits[:, 0] += 4.0
    logits[:2, 0] += 10.0
    for k in (3, 34, 345, 5999, 5120, 64 * 80):
        logits[:, k] += 3.0 * torch.rand(T, generator=g)
    x = torch.log_softmax(logits, -1)
    vr = [1] + list(range(3, V))
    want = CBO.prefix_beam_search(x.numpy(), vr, 6, 9)
    dec = _mod("src.ctc").CTCBeamDecoder(_StubASR(V), vr, 6, 9)
    assert dec._device_search_ok(V)
    assert dec.search_device(x.to(DEV).contiguous()) == want


# ------------------------------------------------------------------------------ several utterances per device step
def _same_hyps(a, b, tol=2e-3):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.outIndex == y.outIndex, i
        assert np.allclose(np.asarray(x.output_scores, np.float32), np.asarray(y.output_scores, np.float32),
                           rtol=tol, atol=tol), i


def test_packed_encoder_equals_one_utterance_at_a_time(ops):
    """Encoder.forward(packed=True) on a zero-padded batch of utterances of DIFFERENT lengths = the encoder run on
    each utterance alone and unpadded (how the reference encodes for decoding: src/decode.py:88 on batch 1,
    bin/test_asr.py:163-167): pyramid 'concat' with odd lengths, 'drop', LayerNorm + projection, both recurrence
    kernels (H = 16 -> f32 MFMA, H = 512 -> bf16x6); frames beyond an utterance's length come back zero when
    nothing but the recurrence touches them"""
    asr = _mod("src.asr")
    g = torch.Generator().manual_seed(3)
    for enc_cfg, D, lens in (
            (dict(prenet='', module='LSTM', bidirection=True, dim=[16, 16, 16], dropout=[0] * 3,
                  layer_norm=[False] * 3, proj=[False] * 3, sample_rate=[2, 2, 1], sample_style='concat'), 9,
             [37, 23, 30, 8]),
            (dict(prenet='', module='LSTM', bidirection=True, dim=[16, 12], dropout=[0] * 2,
                  layer_norm=[True, False], proj=[True, True], sample_rate=[2, 1], sample_style='drop'), 7,
             [21, 40, 5]),
            (dict(prenet='', module='LSTM', bidirection=True, dim=[512, 512], dropout=[0] * 2,
                  layer_norm=[False] * 2, proj=[False] * 2, sample_rate=[2, 2], sample_style='concat'), 40,
             [130, 77, 101, 130, 64])):
        torch.manual_seed(1)
        enc = asr.Encoder(D, **enc_cfg).to(DEV).eval()
        assert enc.supports_packed()
        U, T = len(lens), max(lens)
        feat = torch.zeros(U, T, D)
        for u, l in enumerate(lens):
            feat[u, :l] = torch.randn(l, D, generator=g)
        with torch.no_grad():
            out, out_len = enc(feat.to(DEV), torch.tensor(lens).to(DEV), packed=True)
            for u, l in enumerate(lens):
                one, one_len = enc(feat[u:u + 1, :l].to(DEV), torch.tensor([l]).to(DEV))
                lo, fr = int(one_len[0]), one.shape[1]
                # 'drop' on an odd length keeps ceil(T/r) frames but reports T // r (as the reference); the batch-1
                # LSTM of the next layer runs over ALL frames of its input tensor, and so does the packed one
                assert int(out_len[u]) == lo and int(enc.packed_frames[u]) == fr and fr - lo in (0, 1)
                assert rel_err(out[u, :fr].cpu(), one[0].cpu()) < 2e-6, (enc_cfg["dim"], u)
                lo = fr
                if not any(enc_cfg["layer_norm"]) and not any(enc_cfg["proj"]):
                    assert float(out[u, lo:].abs().max().cpu() if lo < out.shape[1] else 0.0) == 0.0
        ops.check_errors()


def test_multi_utterance_prefix_scores_equal_per_utterance(ops):
    """asrk_ctc_prefix_score_multi_f32: hypotheses of three utterances with different frame counts in one launch =
    three single-utterance launches (src/ctc.py:76-116 per hypothesis)"""
    dops = _mod("decoder_ops")
    rng = np.random.RandomState(4)
    V, C, Ts = 60, 7, [50, 33, 41]
    T = max(Ts)
    x = torch.zeros(3, T, V)
    for u, tu in enumerate(Ts):
        x[u, :tu] = torch.from_numpy(rng.randn(tu, V).astype(np.float32)).log_softmax(-1)
    xd = x.to(DEV)
    rows = [0, 2, 1, 1, 0, 2, 2]
    n = len(rows)
    r_prev = torch.full((n, T, 2), -1e8)
    plen, last = [], []
    for h, u in enumerate(rows):
        r0 = DO.init_state(x[u, :Ts[u]].numpy())
        a = int(rng.randint(2, V))
        _, r1 = DO.prefix_scores(x[u, :Ts[u]].numpy(), [], r0, [a])
        r_prev[h, :Ts[u]] = torch.from_numpy(r1[0])
        plen.append(1)
        last.append(a)
    cands = rng.randint(1, V, size=(n, C)).astype(np.int32)
    cands[:, 0] = 1
    cands[::2, 1] = last[::2]
    psi, r = dops.ctc_prefix_scores(xd, r_prev.to(DEV), plen, last, torch.from_numpy(cands), 0, 1, -1e8,
                                    row_mem=torch.tensor(rows, dtype=torch.int32),
                                    mem_len=torch.tensor(Ts, dtype=torch.int32))
    for h, u in enumerate(rows):
        p1, r1 = dops.ctc_prefix_scores(xd[u, :Ts[u]].contiguous(), r_prev[h:h + 1, :Ts[u]].contiguous().to(DEV),
                                        plen[h:h + 1], last[h:h + 1], torch.from_numpy(cands[h:h + 1]), 0, 1, -1e8)
        assert torch.equal(psi[h], p1[0])
        assert torch.equal(r[h, :, :Ts[u]], r1[0])


@pytest.mark.parametrize("kw,use_lm", [(dict(beam_size=3, ctc_weight=0.4), False),
                                       (dict(beam_size=4, ctc_weight=0.0), False),
                                       (dict(beam_size=1, ctc_weight=0.4), False),
                                       (dict(beam_size=3, ctc_weight=0.4, lm_weight=0.3), True)])
def test_forward_batch_equals_forward_on_the_golden_model(ops, tmp_path, monkeypatch, kw, use_lm):
    """BeamDecoder.forward_batch over all utterances of the reference golden case (different lengths; device-side beam
    bookkeeping, no read-back per position) returns, per utterance, exactly what forward() returns for it alone with the
    HOST record loop (_expand_beam, the bookkeeping pinned on the REAL reference decoder, src/decode.py:64-173)"""
    monkeypatch.setenv("ASRK_DECODE_HOST_BEAM", "1")
    gm = load_golden("las_hybrid_loc")
    model, _, _, V = _asr("las_hybrid_loc")
    if use_lm:
        lm_cfg = dict(emb_tying=False, emb_dim=10, module="LSTM", dim=14, n_layers=2, dropout=0.0)
        yaml.safe_dump({"model": lm_cfg}, open(tmp_path / "lm.yaml", "w"))
        torch.manual_seed(5)
        torch.save({"model": _mod("src.lm").RNNLM(V, **lm_cfg).state_dict()}, tmp_path / "lm.pth")
        kw = dict(kw, lm_path=str(tmp_path / "lm.pth"), lm_config=str(tmp_path / "lm.yaml"))
    dec = _mod("src.decode").BeamDecoder(model, None, min_len_ratio=0.01, max_len_ratio=0.5, **kw)
    assert dec.batchable()
    feat, flen = torch.from_numpy(gm["feat"]).to(DEV), torch.from_numpy(gm["feat_len"]).to(DEV)
    assert len(set(flen.tolist())) > 1
    got = dec.forward_batch(feat, flen)
    ops.check_errors()
    assert len(got) == feat.shape[0]
    for u in range(feat.shape[0]):
        l = int(flen[u])
        _same_hyps(got[u], dec(feat[u:u + 1, :l].contiguous(), flen[u:u + 1]), tol=1e-4)


def test_gru_decoder_beam_search_fused_steps_equal_step_kernels(ops, monkeypatch):
    """single-layer GRU decoder (src/asr.py:172 with module 'GRU') under joint CTC-attention beam search: the fused step
    (asrk_speller_step_f32 with asrk_speller_t::cell = 1), one utterance at a time and several per device step, returns
    the hypotheses of the per-step kernels (ASRK_SPELLER=0), which the reference's 2-layer GRU golden pins"""
    cfg = dict(ctc_weight=0.4,
               encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[24, 24], dropout=[0, 0],
                            layer_norm=[False, False], proj=[False, False], sample_rate=[2, 1], sample_style='drop'),
               attention=dict(mode='loc', dim=20, num_head=1, v_proj=False, temperature=0.6,
                              loc_kernel_size=5, loc_kernel_num=3),
               decoder=dict(module='GRU', dim=28, layer=1, dropout=0))
    Dm, Vm, B, T = 13, 31, 3, 64
    torch.manual_seed(11)
    model = _mod("src.asr").ASR(Dm, Vm, True, cfg["ctc_weight"], cfg["encoder"], cfg["attention"], cfg["decoder"])
    model = model.to(DEV).eval()
    g = torch.Generator().manual_seed(12)
    feat = torch.randn(B, T, Dm, generator=g).to(DEV)
    flen = torch.tensor([64, 50, 38]).to(DEV)
    kw = dict(beam_size=4, min_len_ratio=0.01, max_len_ratio=0.4, ctc_weight=0.4)
    monkeypatch.setenv("ASRK_SPELLER", "0")
    dec0 = _mod("src.decode").BeamDecoder(model, None, **kw)
    want = [dec0(feat[u:u + 1, :int(flen[u])].contiguous(), flen[u:u + 1]) for u in range(B)]
    monkeypatch.setenv("ASRK_SPELLER", "1")
    dec1 = _mod("src.decode").BeamDecoder(model, None, **kw)
    assert dec1.batchable()
    for u in range(B):
        _same_hyps(dec1(feat[u:u + 1, :int(flen[u])].contiguous(), flen[u:u + 1]), want[u], tol=1e-4)
    got = dec1.forward_batch(feat, flen)
    ops.check_errors()
    for u in range(B):
        _same_hyps(got[u], want[u], tol=1e-4)
    assert any(len(h.outIndex) > 1 for h in want[0])          # real hypotheses, not empty strings


def test_forward_batch_at_cfg5_widths_matches_reference(ops, tmp_path):
    """BASELINE configs[4] widths, both golden utterances (T = 800 and 1600) plus two more lengths DECODED TOGETHER
    (beam 16, CTC 0.5, 2 x LSTM-1024 LM 0.5): the T = 800 / 1600 results equal the REAL reference's hypotheses
    (tests/golden/decode_cfg5.npz), the others equal forward() alone"""
    from oracle.gen_golden import CFG3_MODEL, CFG5_LM, CFG5_DECODE, cfg5_weights, cfg5_utterance
    g = load_golden("decode_cfg5")
    sd, lm_sd = cfg5_weights()
    model = _mod("src.asr").ASR(80, 5000, True, CFG3_MODEL["ctc_weight"], CFG3_MODEL["encoder"],
                                CFG3_MODEL["attention"], CFG3_MODEL["decoder"])
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    yaml.safe_dump({"model": CFG5_LM}, open(tmp_path / "lm.yaml", "w"))
    torch.save({"model": lm_sd}, tmp_path / "lm.pth")
    dec = _mod("src.decode").BeamDecoder(model, None, lm_path=str(tmp_path / "lm.pth"),
                                         lm_config=str(tmp_path / "lm.yaml"), **CFG5_DECODE)
    # 8 utterances x beam 16 = 128 rows: the many-row path of a decode position (decoder / LM cells and vocabulary
    # projections as bf16x6 panel GEMMs against cached weight panels, decoder_ops.LSTM_CELL_GEMM_ROWS) - the single
    # utterances it is compared with run 16 rows through the weight-streaming kernels
    Ts = [1600, 800, 1203, 414, 640, 333, 910, 720]
    assert len(Ts) * CFG5_DECODE["beam_size"] >= _mod("decoder_ops").LSTM_CELL_GEMM_ROWS
    feat = torch.zeros(len(Ts), max(Ts), 80)
    for u, T in enumerate(Ts):
        feat[u, :T] = cfg5_utterance(T)[0][0]
    got = dec.forward_batch(feat.to(DEV), torch.tensor(Ts).to(DEV))
    ops.check_errors()
    for u, T in enumerate(Ts):
        if T in (800, 1600):
            tag = "T%d.lm" % T
            assert len(got[u]) == int(g[tag + ".n"]) == 16
            for i, h in enumerate(got[u]):
                assert h.outIndex == g["%s.hyp%d" % (tag, i)].tolist(), (tag, i)
                assert np.allclose(np.asarray(h.output_scores, np.float32), g["%s.score%d" % (tag, i)],
                                   rtol=2e-3, atol=2e-3)
        else:
            _same_hyps(got[u], dec(feat[u:u + 1, :T].to(DEV), torch.tensor([T]).to(DEV)), tol=1e-4)


def test_ctc_beam_forward_batch_equals_forward(ops):
    """CTCBeamDecoder.forward_batch (packed encoder + one prefix-beam launch per utterance on its own stream) =
    forward() on every utterance alone (whose hypotheses test_ctc_beam_decoder_matches_reference pins on the real
    reference, src/ctc.py:241-352)"""
    gm = load_golden("enc_ctc_concat")
    cfg, D, V = CASES["enc_ctc_concat"][0], CASES["enc_ctc_concat"][1], CASES["enc_ctc_concat"][2]
    model = _mod("src.asr").ASR(D, V, True, cfg["ctc_weight"], cfg["encoder"], {}, {})
    model.load_state_dict(golden_state_dict(gm), strict=True)
    model = model.to(DEV).eval()
    dec = _mod("src.ctc").CTCBeamDecoder(model, [1] + list(range(3, V)), beam_size=3, vocab_candidate=4)
    feat, flen = torch.from_numpy(gm["feat"]).to(DEV), torch.from_numpy(gm["feat_len"]).to(DEV)
    got = dec.forward_batch(feat, flen)
    ops.check_errors()
    assert len(got) == feat.shape[0] >= 3
    for u in range(feat.shape[0]):
        l = int(flen[u])
        assert got[u] == dec(feat[u:u + 1, :l].contiguous(), flen[u:u + 1]), u


def test_device_beam_bookkeeping_equals_the_record_loop(ops):
    """asrk_beam_select_f32 (one decode position of every utterance on the device: forward_batch's bookkeeping) against
    BeamDecoder._expand_beam per utterance (the record loop pinned on the reference's hypotheses through forward(),
    src/decode.py:150-167, 209-239): same survivors in the same order with the same parents, labels, scores, candidate
    columns, CTC prefix probabilities and float64 score sums; the same finished hypotheses in the same order; the same
    end-of-utterance decisions - over 200 random positions with tied scores, duplicate labels, <eos> among the top-k,
    labels missing from the CTC candidates, dead rows, several utterances, beam 1 and the length limits"""
    import copy
    import ctypes
    import random
    D = _mod("src.decode")
    L = importlib.import_module(PKG_NAME + "._lib").load()
    p_ = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    for trial in range(200):
        rng = random.Random(trial)
        nr = np.random.RandomState(trial)
        d = object.__new__(D.BeamDecoder)
        B = d.beam_size = rng.choice([1, 2, 3, 5, 16])
        d.apply_ctc = rng.random() < 0.7
        C = int(1.5 * B) if d.apply_ctc else 0
        t = rng.randint(0, 4)
        U = rng.randint(1, 4)
        lmax = 6
        R = U * B
        hi = 9 if B < 8 else 40                                   # label range (duplicates likely for small beams)
        min_len = [rng.randint(0, 5) for _ in range(U)]
        max_len = [rng.randint(1, 6) for _ in range(U)]
        for u in range(U):
            max_len[u] = max(max_len[u], t + 1)                     # the utterance is still searching at position t
        alive = np.zeros(R, np.int32)
        ssum = np.zeros(R, np.float64)
        topv = np.zeros((R, B), np.float32)
        topi = np.ones((R, B), np.int64)
        psi = nr.uniform(-5, 0, (R, max(C, 1))).astype(np.float32)
        cand = nr.randint(1, hi, (R, max(C, 1))).astype(np.int64)
        hyps = {}
        for u in range(U):
            n_u = rng.randint(1, B) if t > 0 else 1
            for i in range(n_u):
                row = u * B + i
                alive[row] = 1
                sc = [float(np.float32(rng.uniform(-3, 0))) for _ in range(t)]
                hyps[row] = D.Hypothesis(None, [rng.randint(3, 9) for _ in range(t)], sc, None, None, 0.0, None)
                ssum[row] = hyps[row].score_sum
                toks = nr.choice(np.arange(1, hi), B, replace=False) if rng.random() < 0.8 else nr.randint(1, hi, B)
                scs = np.sort(nr.uniform(-3, 0, B).astype(np.float32))[::-1]
                if rng.random() < 0.3:
                    scs[1:] = scs[0]                                # ties
                topv[row], topi[row] = scs, toks
                if d.apply_ctc and rng.random() < 0.7:              # most labels are among the candidates
                    cand[row, :B] = toks
        dv = lambda a: torch.from_numpy(a).to(DEV)
        alive_d, ssum_d, topv_d, topi_d, psi_d, cand_d = dv(alive), dv(ssum), dv(topv), dv(topi), dv(psi), dv(cand)
        i32 = dict(dtype=torch.int32, device=DEV)
        utt_done = torch.zeros(U, **i32)
        prev_token, parent, col = (torch.zeros(R, dtype=torch.int64, device=DEV) for _ in range(3))
        pctc = torch.zeros(R, dtype=torch.float32, device=DEV)
        hist_tok, hist_par = torch.zeros((lmax, R), **i32), torch.zeros((lmax, R), **i32)
        hist_sc = torch.zeros((lmax, R), dtype=torch.float32, device=DEV)
        fcap = B * (lmax + 2)
        fin_count = torch.zeros(U, **i32)
        fin_kind, fin_t, fin_row = (torch.zeros((U, fcap), **i32) for _ in range(3))
        fin_term = torch.zeros((U, fcap), dtype=torch.float32, device=DEV)
        fin_ssum = torch.zeros((U, fcap), dtype=torch.float64, device=DEV)
        live = torch.full((1,), U, **i32)
        min_d, max_d = dv(np.asarray(min_len, np.int32)), dv(np.asarray(max_len, np.int32))
        rc = L.asrk_beam_select_f32(p_(topv_d), p_(topi_d), p_(psi_d) if C else None, p_(cand_d) if C else None, U, B, C, t,
                                    lmax, fcap, p_(min_d), p_(max_d),
                                    p_(alive_d), p_(ssum_d), p_(utt_done), p_(prev_token), p_(parent), p_(col), p_(pctc),
                                    p_(hist_tok), p_(hist_sc), p_(hist_par), p_(fin_count), p_(fin_kind), p_(fin_t),
                                    p_(fin_row), p_(fin_term), p_(fin_ssum), p_(live), None)
        assert rc == 0
        torch.cuda.synchronize()
        a2, s2, tk, pa, co, pc = [v.cpu().numpy() for v in (alive_d, ssum_d, prev_token, parent, col, pctc)]
        n_done = 0
        for u in range(U):
            rows = [r for r in range(u * B, (u + 1) * B) if alive[r]]
            prev = [copy.deepcopy(hyps[r]) for r in rows]
            packed = [list(map(float, topv[r])) + list(map(float, topi[r])) +
                      (list(map(float, psi[r, :C])) + list(map(float, cand[r, :C])) if C else []) for r in rows]
            finals = []
            nxt, done = d._expand_beam(prev, packed, t, min_len[u], finals, C)
            ending = done or not nxt or t + 1 >= max_len[u]
            want_fin = [(0, t, rows[prev.index(h)], h.output_scores[-1], h.score_sum) for h in finals]
            if ending and not done:
                want_fin += [(1, t, u * B + j, 0.0, h.score_sum) for j, h in enumerate(nxt)]
            n = int(fin_count[u])
            got_fin = [(int(fin_kind[u, j]), int(fin_t[u, j]), int(fin_row[u, j]), float(fin_term[u, j]),
                        float(fin_ssum[u, j])) for j in range(n)]
            assert got_fin == want_fin, (trial, u)
            assert int(utt_done[u]) == int(ending), (trial, u)
            n_done += int(ending)
            if ending:
                assert not a2[u * B:(u + 1) * B].any()
                continue
            for j, h in enumerate(nxt):
                slot = u * B + j
                assert a2[slot] == 1
                assert (int(pa[slot]), int(tk[slot]), int(co[slot])) == (rows[h.parent], h.output_seq[-1], h.cand)
                assert s2[slot] == h.score_sum                      # float64, same additions in the same order
                assert float(hist_sc[t, slot]) == h.output_scores[-1] and int(hist_par[t, slot]) == rows[h.parent]
                if C:
                    assert float(pc[slot]) == h.ctc_prob
            assert not a2[u * B + len(nxt):(u + 1) * B].any()
        assert int(live) == U - n_done


# max_len_ratio <= 0.25: never more labels than encoder frames (time reduction 4).  Beyond that every CTC candidate is
# infeasible, all scores sit at LOG_ZERO scale where f32 resolves 0.5 and the survivors are decided by rounding noise - the
# reference itself dies there (list.index ValueError, src/decode.py:225); nothing to compare.
@pytest.mark.parametrize("beam,ctc_w,lm_w,minr,maxr", [(2, 0.3, 0.0, 0.0, 0.25), (5, 0.6, 0.4, 0.05, 0.25), (7, 0.0, 0.5, 0.0, 0.2),
                                                       (1, 0.5, 0.3, 0.1, 0.25), (8, 0.2, 0.0, 0.0, 0.25)])
def test_device_beam_loop_equals_host_record_loop_on_random_batches(ops, tmp_path, monkeypatch, beam, ctc_w, lm_w, minr, maxr):
    """BeamDecoder.forward_batch (device bookkeeping, fixed row slots, dead rows, early-ending utterances, the <eos> log)
    on batches of random utterances of very different lengths - down to fewer frames than the encoder's time reduction -
    against forward() with the HOST record loop on every utterance alone: the golden toy model has a 10-odd label
    vocabulary, so <eos> shows up among the top-k all the time and utterances end at different positions"""
    model, _, _, V = _asr("las_hybrid_loc")
    kw = dict(beam_size=beam, ctc_weight=ctc_w, min_len_ratio=minr, max_len_ratio=maxr)
    if lm_w > 0:
        lm_cfg = dict(emb_tying=False, emb_dim=10, module="LSTM", dim=14, n_layers=2, dropout=0.0)
        yaml.safe_dump({"model": lm_cfg}, open(tmp_path / "lm.yaml", "w"))
        torch.manual_seed(7)
        torch.save({"model": _mod("src.lm").RNNLM(V, **lm_cfg).state_dict()}, tmp_path / "lm.pth")
        kw.update(lm_weight=lm_w, lm_path=str(tmp_path / "lm.pth"), lm_config=str(tmp_path / "lm.yaml"))
    dec = _mod("src.decode").BeamDecoder(model, None, **kw)
    assert dec.batchable()
    gm = load_golden("las_hybrid_loc")
    D = gm["feat"].shape[2]
    g = torch.Generator().manual_seed(beam * 10 + int(ctc_w * 10))
    lens = [47, 3, 19, 1, 33, 8, 26]
    feat = torch.zeros(len(lens), max(lens), D)
    for u, l in enumerate(lens):
        feat[u, :l] = torch.randn(l, D, generator=g) * 1.5
    got = dec.forward_batch(feat.to(DEV), torch.tensor(lens).to(DEV))
    ops.check_errors()
    monkeypatch.setenv("ASRK_DECODE_HOST_BEAM", "1")
    n_eos = 0
    for u, l in enumerate(lens):
        if l < 8:                     # fewer frames than the encoder's time reduction: only the batched (packed) path
            assert isinstance(got[u], list)          # encodes them (forward() alone raises, as the reference does)
            continue
        want = dec(feat[u:u + 1, :l].contiguous().to(DEV), torch.tensor([l]).to(DEV))
        _same_hyps(got[u], want, tol=1e-4)
        n_eos += sum(1 for h in want if h.outIndex and h.outIndex[-1] == 1)
    ops.check_errors()
    print("hypotheses finished by <eos> in the compared beams:", n_eos)
