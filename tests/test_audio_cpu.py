"""CPU: audio oracle. Delta/CMVN/Postprocess are pinned to the reference's own classes
(tests/golden/audio_post.npz, made by oracle/gen_golden.py); the Kaldi fbank restatement is
checked against every invariant the reference's tests pin (tests/test_audio.py:13-103); its
absolute values are pinned on third-party kaldi-compatible code (transformers.audio_utils, the numpy stand-in for
torchaudio.compliance.kaldi.fbank that HuggingFace's extractors use) since torchaudio itself is not installable."""
import numpy as np
import pytest

from oracle import fbank_oracle as FO
from helpers import load_golden, rel_err


def _wave(g):
    return g["wave_i16"].astype(np.float64) / 32768.0, int(g["sample_rate"])


def test_fbank_shape_and_determinism():
    g = load_golden("audio_post")
    x, sr = _wave(g)
    fb = FO.kaldi_fbank(x, sr, num_mel_bins=40)
    assert fb.shape == (392, 40)                       # tests/test_audio.py:13-24
    assert np.all(np.isfinite(fb))
    assert rel_err(fb, g["fbank"]) < 1e-6              # the committed fixture is this function's output
    assert FO.frame_geometry(len(x), sr) == (400, 160, 512, 392)


def test_fbank_matches_direct_dft_definition():
    """independent restatement: explicit DFT sums on a few frames"""
    rng = np.random.RandomState(0)
    x = rng.randn(16000 // 4)
    fb = FO.kaldi_fbank(x, 16000, num_mel_bins=23)
    win, shift, padded, m = FO.frame_geometry(len(x), 16000)
    W = FO.mel_banks(23, padded, 16000)
    for i in (0, m // 2, m - 1):
        fr = x[i * shift:i * shift + win].copy()
        fr -= fr.mean()
        fr = fr - 0.97 * np.concatenate([fr[:1], fr[:-1]])
        fr *= FO.povey_window(win)
        k = np.arange(padded // 2 + 1)[:, None]
        n = np.arange(win)[None, :]
        X = (fr[None, :] * np.exp(-2j * np.pi * k * n / padded)).sum(1)
        ref = np.log(np.maximum(W @ (np.abs(X) ** 2), FO.EPS))
        assert np.allclose(fb[i], ref, rtol=1e-9, atol=1e-9)


def test_mel_banks_properties():
    W = FO.mel_banks(40, 512, 16000)
    assert W.shape == (40, 257) and np.all(W >= 0) and np.all(W[:, -1] == 0)
    peaks = W.argmax(1)
    assert np.all(np.diff(peaks) > 0)                   # centre frequencies increase
    assert W[:, 0].sum() == 0                           # 0 Hz is below low_freq = 20 Hz


def test_cmvn_and_delta_invariants():
    g = load_golden("audio_post")
    x, sr = _wave(g)
    y0 = FO.audio_transform(x, sr, 40, delta_order=0)
    y1 = FO.audio_transform(x, sr, 40, delta_order=1)
    y2 = FO.audio_transform(x, sr, 40, delta_order=2)
    assert y0.shape == (392, 40) and y1.shape == (392, 80) and y2.shape == (392, 120)
    assert np.allclose(y0.mean(0), 0, atol=5e-5)        # tests/test_audio.py:41-55
    assert np.allclose(y0.std(0, ddof=1), 1, atol=1e-6)
    assert np.allclose(y1[:, :40], y0, rtol=1e-5, atol=1e-5)   # tests/test_audio.py:57-87
    assert np.allclose(y2[:, :80], y1, rtol=1e-5, atol=1e-5)


def test_delta_cmvn_postprocess_match_reference_classes():
    g = load_golden("audio_post")
    fb = g["fbank"].astype(np.float64)
    for order in (0, 1, 2):
        x = fb.T[None]
        if order >= 1:
            x = FO.delta(x[0], order, 2)
        y = FO.postprocess(FO.cmvn(x))
        assert rel_err(y, g["post_order%d" % order]) < 2e-5
    y = FO.postprocess(FO.delta(fb.T, 2, 2))
    assert rel_err(y, g["delta2_nocmvn"]) < 1e-6
    f = FO.delta_filters(2, 2)
    assert np.allclose(f[1], [0, 0, -.2, -.1, 0, .1, .2, 0, 0])
    assert np.allclose(f[2], [.04, .04, .01, -.04, -.1, -.04, .01, .04, .04])


def test_mfcc_oracle_is_dct_of_log_mel():
    """C0 = sum(log-mel)/sqrt(N) (lifter(0) = 1); with num_ceps == num_mel_bins and no lifter the
    orthonormal DCT preserves the energy of every frame"""
    rng = np.random.RandomState(4)
    x = 0.1 * rng.randn(16000)
    mel = FO.kaldi_fbank(x, 16000, num_mel_bins=13).astype(np.float64)
    c = FO.kaldi_mfcc(x, 16000, num_mel_bins=13, num_ceps=13)
    assert np.allclose(c[:, 0], mel.sum(1) / np.sqrt(13.0), rtol=1e-6, atol=1e-6)
    c_nolift = FO.kaldi_mfcc(x, 16000, num_mel_bins=13, num_ceps=13, cepstral_lifter=0.0)
    assert np.allclose((c_nolift ** 2).sum(1), (mel ** 2).sum(1), rtol=1e-6)
    assert FO.kaldi_mfcc(x, 16000, num_mel_bins=26, num_ceps=13).shape == (98, 13)


# ---------------------------------------------------------------------------------------------------
# Independent pins of the fbank restatement: nothing below shares code with oracle/fbank_oracle.py
# (tests/fbank_independent.py: scipy.signal / scipy.fft second implementation + closed-form answers).
import math

import fbank_independent as FI


def test_mel_scale_constants_by_hand():
    """literal constants from the published definitions (computed by hand, see fbank_independent.py)"""
    assert abs(float(FO.mel_scale(20.0)) - FI.MEL_LOW_20HZ) < 1e-8
    assert abs(float(FO.mel_scale(8000.0)) - FI.MEL_HIGH_8KHZ) < 1e-8
    for M, table in FI.CENTRES_HZ.items():
        W = FO.mel_banks(M, 512, 16000)
        for m, hz in table.items():
            k = hz / (16000 / 512)                              # fractional FFT bin of the centre
            assert abs(int(W[m].argmax()) - k) <= 1.0, (M, m)   # the peak bin brackets the centre
        # triangles are linear in mel and adjacent ones overlap: on every FFT bin between the first and
        # the last centre the weights of all filters sum to exactly one
        d = (FI.MEL_HIGH_8KHZ - FI.MEL_LOW_20HZ) / (M + 1)
        for k in range(256):
            mk = 1127.0 * math.log(1.0 + k * 31.25 / 700.0)
            if FI.MEL_LOW_20HZ + d <= mk <= FI.MEL_LOW_20HZ + M * d:
                assert abs(W[:, k].sum() - 1.0) < 1e-9, (M, k)


def test_fbank_known_answers_closed_form():
    sr = 16000
    # (1) constant input: DC removal leaves an all-zero frame -> every bin sits on the log floor
    fb = FO.kaldi_fbank(np.full(16000, 0.37), sr, num_mel_bins=40)
    assert fb.shape == (98, 40) and np.allclose(fb, FI.LOG_FLOOR, atol=1e-12)
    # (2) scaling the waveform by a shifts every log energy by 2 ln a
    rng = np.random.RandomState(1)
    x = 0.1 * rng.randn(4000)
    assert np.allclose(FO.kaldi_fbank(3.0 * x, sr, 40), FO.kaldi_fbank(x, sr, 40) + 2 * math.log(3.0), atol=1e-9)
    # (3) Parseval: total mel energy of a mid-band pure tone (1 kHz: 25 periods per frame, FFT bin 32)
    for f0 in (1000.0, 2000.0, 500.0):
        fb = FO.kaldi_fbank(FI.tone(f0, 400 + 160 * 3, sr), sr, num_mel_bins=40)
        want = FI.tone_frame_energy(f0, sr)
        got = np.exp(fb).sum(axis=1)
        assert np.allclose(got, want, rtol=2e-4), (f0, got, want)
        # and the energy sits in the filter whose band holds the tone
        hz = [700.0 * (math.exp((FI.MEL_LOW_20HZ + (m + 1) * (FI.MEL_HIGH_8KHZ - FI.MEL_LOW_20HZ) / 41) / 1127.0) - 1)
              for m in range(40)]
        nearest = int(np.argmin([abs(h - f0) for h in hz]))
        assert abs(int(fb[0].argmax()) - nearest) <= 1


@pytest.mark.parametrize("sr,nmel,n", [(16000, 40, 16000), (16000, 80, 5000), (16000, 23, 401), (8000, 23, 3000)])
def test_fbank_oracle_equals_scipy_implementation(sr, nmel, n):
    rng = np.random.RandomState(n)
    t = np.arange(n) / sr
    x = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 2750 * t + 1.0) + 0.05 * rng.randn(n) + 0.02
    a, b = FO.kaldi_fbank(x, sr, num_mel_bins=nmel), FI.scipy_fbank(x, sr, nmel)
    assert a.shape == b.shape and np.allclose(a, b, rtol=1e-9, atol=1e-9)


def test_fbank_sample_wav_equals_scipy_implementation():
    """the reference's own fixture utterance (tests/sample_data/3830-12529-0005.wav, 392 frames)"""
    g = load_golden("audio_post")
    x, sr = _wave(g)
    b = FI.scipy_fbank(x, sr, 40)
    assert b.shape == (392, 40)
    assert np.allclose(FO.kaldi_fbank(x, sr, num_mel_bins=40), b, rtol=1e-9, atol=1e-9)
    assert np.allclose(g["fbank"], b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("nmel", [23, 40, 80])
def test_fbank_oracle_equals_third_party_kaldi_compatible_pipeline(nmel):
    """PIN on third-party code: HuggingFace `transformers.audio_utils` ships the numpy pipeline its speech
    feature extractors (AST, SeamlessM4T, ...) run IN PLACE OF `torchaudio.compliance.kaldi.fbank` when
    torchaudio is not installed - the same call the reference makes at src/audio.py:104-108 - with exactly these
    arguments (povey window, 25 ms / 10 ms frames, 512-point FFT, power spectrum, no centring, pre-emphasis 0.97,
    per-frame DC removal, 'kaldi' mel scale triangularised in mel space from 20 Hz to Nyquist, floor
    FLT_EPSILON, natural log).  It was written against torchaudio by people who had it and shares no code with
    this repository; the oracle agrees with it to 2e-7 on the reference's fixture utterance and on synthetic
    signals.  (torchaudio itself is not installable here: SURVEY.md §8c.)"""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    g = load_golden("audio_post")
    sr = int(g["sample_rate"])
    rng = np.random.RandomState(nmel)
    t = np.arange(sr * 2 + 77) / sr
    signals = [g["wave_i16"].astype(np.float64) / 32768.0,
               0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.randn(len(t)) + 0.02,
               np.full(2000, 0.25)]                                      # constant: every bin hits the floor
    mf = audio_utils.mel_filter_bank(num_frequency_bins=257, num_mel_filters=nmel, min_frequency=20,
                                     max_frequency=sr // 2, sampling_rate=sr, norm=None, mel_scale="kaldi",
                                     triangularize_in_mel_space=True)
    win = audio_utils.window_function(400, "povey", periodic=False)
    for x in signals:
        ref = audio_utils.spectrogram(x, win, frame_length=400, hop_length=160, fft_length=512, power=2.0,
                                      center=False, preemphasis=0.97, mel_filters=mf, log_mel="log",
                                      mel_floor=1.192092955078125e-07, remove_dc_offset=True, dtype=np.float64).T
        got = FO.kaldi_fbank(x, sr, num_mel_bins=nmel)
        assert got.shape == ref.shape == (1 + (len(x) - 400) // 160, nmel)
        assert np.max(np.abs(got - ref)) < 2e-6
