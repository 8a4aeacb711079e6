"""CPU: audio oracle. Delta/CMVN/Postprocess are pinned to the reference's own classes
(tests/golden/audio_post.npz, made by oracle/gen_golden.py); the Kaldi fbank restatement is
checked against every invariant the reference's tests pin (tests/test_audio.py:13-103) — its
absolute values are PARITY-UNPINNED (no torchaudio in the image)."""
import numpy as np

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
