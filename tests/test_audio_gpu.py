"""GPU parity of the audio front end (csrc/audio.hip + 2 GEMMs) vs the CPU oracle and vs the
reference's own Delta/CMVN/Postprocess outputs (golden)."""
import importlib

import numpy as np
import pytest
import torch

from conftest import PKG_NAME
from oracle import fbank_oracle as FO
from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def audio(ops):
    return importlib.import_module(PKG_NAME + ".src.audio")


def test_fbank_sample_wav_vs_oracle(audio):
    g = load_golden("audio_post")
    w = torch.from_numpy(g["wave_i16"].astype(np.float32) / 32768.0).unsqueeze(0).to(DEV)
    fb = audio.kaldi_fbank(w, int(g["sample_rate"]), num_mel_bins=40, frame_length=25, frame_shift=10,
                           dither=0)
    assert fb.shape == (392, 40)
    # log-mel values span ~[-16, 3]; compare in absolute terms too (f32 DFT vs f64 FFT)
    assert torch.max(torch.abs(fb.cpu() - torch.from_numpy(g["fbank"]))).item() < 2e-3
    assert rel_err(fb.cpu(), g["fbank"]) < 1e-3


@pytest.mark.parametrize("n,sr,nmel", [(16000 * 10, 16000, 80), (401, 16000, 40), (399, 16000, 40),
                                       (8000 * 3 + 17, 8000, 23)])
def test_fbank_synthetic_vs_oracle(audio, n, sr, nmel):
    rng = np.random.RandomState(n % 97)
    t = np.arange(n) / sr
    x = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.randn(n) + 0.01
    ref = FO.kaldi_fbank(x, sr, num_mel_bins=nmel)
    fb = audio.kaldi_fbank(torch.from_numpy(x.astype(np.float32)).unsqueeze(0).to(DEV), sr,
                           num_mel_bins=nmel, dither=0)
    assert tuple(fb.shape) == ref.shape
    if ref.shape[0]:
        assert torch.max(torch.abs(fb.cpu().double() - torch.from_numpy(ref))).item() < 2e-3


@pytest.mark.parametrize("order", [0, 1, 2])
def test_delta_cmvn_postprocess_vs_reference_golden(audio, order):
    g = load_golden("audio_post")
    x = torch.from_numpy(g["fbank"].T.copy()).unsqueeze(0).to(DEV)     # [1, D, T]
    mods = ([audio.Delta(order, 2)] if order >= 1 else []) + [audio.CMVN(), audio.Postprocess()]
    y = x
    for m in mods:
        y = m(y)
    assert rel_err(y.cpu(), g["post_order%d" % order]) < 1e-3
    if order == 2:
        y2 = audio.Postprocess()(audio.Delta(2, 2)(x))
        assert rel_err(y2.cpu(), g["delta2_nocmvn"]) < 1e-5


def test_create_transform_contract(audio, tmp_path):
    """create_transform(audio_config) -> (callable(filepath) -> [T, D'], feat_dim)  (audio.py:115-133)"""
    import wave
    g = load_golden("audio_post")
    path = str(tmp_path / "u.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(g["sample_rate"]))
        w.writeframes(g["wave_i16"].astype("<i2").tobytes())
    cfg = dict(feat_type="fbank", feat_dim=40, frame_length=25, frame_shift=10, dither=0,
               apply_cmvn=True, delta_order=2, delta_window_size=2)
    tr, dim = audio.create_transform(cfg)
    assert dim == 120
    y = tr(path)
    assert tuple(y.shape) == (392, 120)                                  # tests/test_audio.py:89-103
    ref = FO.audio_transform(g["wave_i16"].astype(np.float64) / 32768.0, int(g["sample_rate"]), 40,
                             delta_order=2)
    assert rel_err(y.cpu(), ref) < 2e-3
    assert np.allclose(y.cpu().numpy().mean(0), 0, atol=5e-5)            # tests/test_audio.py:41-55


@pytest.mark.parametrize("n,sr,nmel,nceps", [(16000 * 2, 16000, 13, 13), (16000 + 123, 16000, 26, 13), (399, 16000, 13, 13)])
def test_mfcc_synthetic_vs_oracle(audio, n, sr, nmel, nceps):
    """feat_type 'mfcc' (src/audio.py:96): fbank chain + DCT/lifter GEMM vs the float64 oracle"""
    rng = np.random.RandomState(n % 89)
    t = np.arange(n) / sr
    x = 0.3 * np.sin(2 * np.pi * 300 * t) + 0.05 * rng.randn(n)
    ref = FO.kaldi_mfcc(x, sr, num_mel_bins=nmel, num_ceps=nceps)
    y = audio.kaldi_mfcc(torch.from_numpy(x.astype(np.float32)).unsqueeze(0).to(DEV), sr, num_mel_bins=nmel,
                         num_ceps=nceps, dither=0)
    assert tuple(y.shape) == ref.shape
    if ref.shape[0]:
        assert torch.max(torch.abs(y.cpu().double() - torch.from_numpy(ref))).item() < 5e-3


def test_mfcc_transform_feeds_vgg_layout(audio):
    """create_transform(feat_type='mfcc', feat_dim=13, delta_order=2) -> [T, 39]: the 13-bin layout the
    VGG prenet's check_dim expects (src/module.py:34-36)"""
    tr, dim = audio.create_transform(dict(feat_type='mfcc', feat_dim=13, frame_length=25, frame_shift=10,
                                          dither=0, apply_cmvn=True, delta_order=2, delta_window_size=2))
    assert dim == 39
    rng = np.random.RandomState(0)
    wav = torch.from_numpy((0.1 * rng.randn(1, 16000)).astype(np.float32))
    feat = tr((wav, 16000))
    assert feat.shape == (98, 39) and torch.isfinite(feat).all()
    assert torch.allclose(feat.mean(0).cpu(), torch.zeros(39), atol=1e-3)      # CMVN'd


# ---------------------------------------------------------------------------------------------------
# The HIP feature pipeline against checks that share no code with oracle/fbank_oracle.py
# (tests/fbank_independent.py: scipy second implementation + closed-form known answers)
import math

import fbank_independent as FI


def _hip_fbank(audio, x, sr, nmel):
    return audio.kaldi_fbank(torch.from_numpy(np.asarray(x, np.float32)).unsqueeze(0).to(DEV), sr,
                             num_mel_bins=nmel, frame_length=25, frame_shift=10, dither=0).cpu().double().numpy()


def test_fbank_hip_known_answers_closed_form(audio):
    sr = 16000
    fb = _hip_fbank(audio, np.full(16000, 0.37), sr, 40)          # constant input -> log floor everywhere
    assert fb.shape == (98, 40) and np.allclose(fb, FI.LOG_FLOOR, atol=1e-5)
    rng = np.random.RandomState(1)
    x = 0.1 * rng.randn(4000)
    assert np.allclose(_hip_fbank(audio, 3.0 * x, sr, 40), _hip_fbank(audio, x, sr, 40) + 2 * math.log(3.0),
                       atol=2e-3)
    for f0 in (1000.0, 2000.0, 500.0):                            # Parseval total of a mid-band tone
        fb = _hip_fbank(audio, FI.tone(f0, 400 + 160 * 3, sr), sr, 40)
        assert np.allclose(np.exp(fb).sum(axis=1), FI.tone_frame_energy(f0, sr), rtol=1e-3), f0


@pytest.mark.parametrize("sr,nmel,n", [(16000, 40, 16000), (16000, 80, 5000), (16000, 23, 401), (8000, 23, 3000)])
def test_fbank_hip_equals_scipy_implementation(audio, sr, nmel, n):
    rng = np.random.RandomState(n)
    t = np.arange(n) / sr
    x = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 2750 * t + 1.0) + 0.05 * rng.randn(n) + 0.02
    a, b = _hip_fbank(audio, x, sr, nmel), FI.scipy_fbank(x, sr, nmel)
    assert a.shape == b.shape and np.max(np.abs(a - b)) < 2e-3


def test_fbank_hip_sample_wav_equals_scipy_implementation(audio):
    """the reference's fixture utterance: 392 x 40 (tests/test_audio.py:13-24), absolute values vs the
    independent implementation, CMVN statistics (tests/test_audio.py:41-55), delta channel identity
    (tests/test_audio.py:57-87)"""
    g = load_golden("audio_post")
    x = g["wave_i16"].astype(np.float64) / 32768.0
    a, b = _hip_fbank(audio, x, int(g["sample_rate"]), 40), FI.scipy_fbank(x, int(g["sample_rate"]), 40)
    assert a.shape == (392, 40) and np.max(np.abs(a - b)) < 2e-3
    tr0, _ = audio.create_transform(dict(feat_type="fbank", feat_dim=40, frame_length=25, frame_shift=10,
                                         dither=0, apply_cmvn=True, delta_order=0))
    tr1, _ = audio.create_transform(dict(feat_type="fbank", feat_dim=40, frame_length=25, frame_shift=10,
                                         dither=0, apply_cmvn=True, delta_order=1, delta_window_size=2))
    wav = (torch.from_numpy(x.astype(np.float32)).unsqueeze(0), int(g["sample_rate"]))
    y0, y1 = tr0(wav).cpu().numpy(), tr1(wav).cpu().numpy()
    assert y0.shape == (392, 40) and y1.shape == (392, 80)
    assert np.allclose(y0.mean(0), 0.0, atol=5e-5) and np.allclose(y0.std(0, ddof=1), 1.0, atol=1e-4)
    assert np.allclose(y1[:, :40], y0, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------ whole-batch front end (§8 f1)
def _write_wav(path, pcm, sr):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(np.asarray(pcm, dtype="<i2").tobytes())


def _ragged_pcm(lengths, seed):
    rng = np.random.RandomState(seed)
    out = []
    for i, n in enumerate(lengths):
        t = np.arange(n) / 16000.0
        x = 0.25 * np.sin(2 * np.pi * (200 + 37 * i) * t) + 0.05 * rng.randn(n) + 0.01
        out.append(np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16))
    return out


@pytest.mark.parametrize("feat_type,feat_dim,order,cmvn", [("fbank", 80, 0, True), ("fbank", 40, 2, True),
                                                          ("fbank", 40, 1, False), ("mfcc", 13, 2, True)])
def test_batch_front_end_equals_per_file_transform(audio, feat_type, feat_dim, order, cmvn):
    """BatchFeatureTransform (padded int16 PCM of the whole batch, 7 launches) against the per-file module chain
    of create_transform on every utterance: same values (CMVN sums in another order: 1e-4 on utterances of a
    handful of frames), zero padding beyond
    each utterance, frame counts"""
    cfg = dict(feat_type=feat_type, feat_dim=feat_dim, frame_length=25, frame_shift=10, dither=0, apply_cmvn=cmvn,
               delta_order=order, delta_window_size=2)
    tr, dim = audio.create_transform(dict(cfg))
    assert tr.batch is not None and tr.batch.out_dim == dim
    lengths = [16000 * 3 + 11, 16000 * 3 + 11, 16000 * 2, 400 + 160 * 5, 401, 16000 + 7]
    pcm = _ragged_pcm(lengths, seed=order + feat_dim)
    feat, flen = tr.batch(pcm, 16000)
    assert feat.shape == (len(pcm), int(flen.max()), dim) and flen.dtype == torch.int64
    for b, x in enumerate(pcm):
        ref = tr((torch.from_numpy(x.astype(np.float32) / 32768.0).unsqueeze(0), 16000))
        m = ref.shape[0]
        assert int(flen[b]) == m
        if m > 1:
            assert rel_err(feat[b, :m].cpu(), ref.cpu()) < 1e-4, b
        assert float(feat[b, m:].abs().max().cpu()) == 0.0 if m < feat.shape[1] else True


def test_batch_front_end_vs_oracle_on_reference_fixture(audio):
    """the reference's fixture utterance inside a batch, against the float64 oracle chain (fbank -> delta ->
    CMVN -> postprocess) and the reference-generated delta/CMVN golden"""
    g = load_golden("audio_post")
    sr = int(g["sample_rate"])
    cfg = dict(feat_type="fbank", feat_dim=40, frame_length=25, frame_shift=10, dither=0, apply_cmvn=True,
               delta_order=2, delta_window_size=2)
    bt = audio.BatchFeatureTransform(cfg)
    wav = g["wave_i16"].astype(np.int16)
    feat, flen = bt([wav[:30000], wav, wav[:5000]], sr)
    assert flen.tolist() == [1 + (30000 - 400) // 160, 392, 1 + (5000 - 400) // 160]
    ref = FO.audio_transform(wav.astype(np.float64) / 32768.0, sr, 40, delta_order=2)
    assert rel_err(feat[1, :392].cpu(), ref) < 2e-3
    assert np.allclose(feat[1, :392].cpu().numpy().mean(0), 0, atol=5e-5)


def test_collate_through_batch_front_end_equals_per_file_collate(audio, tmp_path, monkeypatch):
    """collect_audio_batch (src/data.py:14-43) over real wav files: names, order (descending length, stable),
    halving rule, lengths and features are the same through the whole-batch front end and through the
    per-file path (ASRK_BATCH_FBANK=0)"""
    data = importlib.import_module(PKG_NAME + ".src.data")
    lengths = [16000 * 9, 16000 * 2, 16000 * 5, 16000 * 5, 16000 * 1, 16000 * 7]      # first > 800 frames -> halved
    pcm = _ragged_pcm(lengths, seed=4)
    batch = []
    for i, x in enumerate(pcm):
        p = tmp_path / ("utt%d.wav" % i)
        _write_wav(p, x, 16000)
        batch.append((str(p), [3 + i, 4, 1]))
    cfg = dict(feat_type="fbank", feat_dim=40, frame_length=25, frame_shift=10, dither=0, apply_cmvn=True,
               delta_order=1, delta_window_size=2)
    tr, _ = audio.create_transform(dict(cfg))
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ASRK_BATCH_FBANK", flag)
        for mode in ("train", "test"):
            outs[flag, mode] = data.collect_audio_batch(list(batch), tr, mode, n_jobs=2)
    for mode in ("train", "test"):
        (n1, f1, l1, t1), (n0, f0, l0, t0) = outs["1", mode], outs["0", mode]
        assert n1 == n0 and l1.tolist() == l0.tolist() and torch.equal(t1, t0)
        assert len(n1) == (3 if mode == "train" else 6)
        assert f1.is_cuda and f1.shape == f0.shape
        assert rel_err(f1.cpu(), f0.cpu()) < 2e-5
