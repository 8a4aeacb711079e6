"""CPU ORACLE for the audio front end (test infrastructure only).

Restates in float64 numpy:
  * Kaldi-compliant log-mel filterbank as called at src/audio.py:104-108
    (`torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins, channel=-1, sample_frequency,
    frame_length, frame_shift, dither)` with the yaml kwargs of config/libri/asr_example.yaml:10-14).
    torchaudio is NOT in this image (and unpinned in requirements.txt:9), so this follows the
    published algorithm (Kaldi feature-window / mel-computations as mirrored by
    torchaudio.compliance.kaldi): snip_edges framing, per-frame DC removal, pre-emphasis 0.97 with
    replicate padding, povey window hann(N, periodic=False)^0.85, zero-pad to the next power of
    two, |rFFT|^2, triangular mel banks built in the mel domain (low 20 Hz, high = Nyquist,
    vtln_warp 1), log(max(E, FLT_EPSILON)).
    **No golden from torchaudio itself**: the reference's own tests pin only shapes and CMVN statistics
    (tests/test_audio.py:13-103) and no torchaudio build is available to produce a dump.  The absolute
    values are pinned (a) on THIRD-PARTY code: HuggingFace transformers.audio_utils (installed, v5.x) carries the
    numpy pipeline its speech feature extractors run in place of torchaudio.compliance.kaldi.fbank when torchaudio
    is missing - written and validated upstream against torchaudio; with the reference's arguments it agrees with
    this file to 2e-7 (tests/test_audio_cpu.py::test_fbank_oracle_equals_third_party_kaldi_compatible_pipeline);
    (b) on checks that share no code with this file (tests/fbank_independent.py,
    tests/test_audio_cpu.py): a second implementation on scipy.signal / scipy.fft primitives, closed-form
    known answers (log floor on constant input, 2 ln a scale shift, Parseval total of a pure tone, mel
    triangles summing to one, hand-computed mel constants) and the invariants the reference tests check.
  * Delta (src/audio.py:33-80), CMVN (src/audio.py:7-30), Postprocess (src/audio.py:83-89): these
    ARE pinned against the reference's own classes (imported with a torchaudio stub) by
    tests/golden/audio_post.npz.
"""
import math
import wave

import numpy as np

EPS = 1.1920928955078125e-07  # torch.finfo(torch.float32).eps, the log floor


def read_wav(path):
    """PCM wav -> (float64 [C, N] in [-1, 1), sample_rate); what torchaudio.load returns
    (src/audio.py:102) for 16-bit PCM."""
    with wave.open(path, "rb") as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw != 2:
        raise ValueError("only 16-bit PCM wav is supported")
    x = np.frombuffer(raw, dtype="<i2").astype(np.float64).reshape(-1, nch).T / 32768.0
    return x, sr


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, np.float64) / 700.0)


def mel_banks(num_bins, padded, sample_freq, low_freq=20.0, high_freq=0.0):
    """[num_bins, padded/2 + 1] triangular weights (last column zero, as torchaudio pads it)."""
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded
    mel_low, mel_high = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
    mel = mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.pad(w, ((0, 0), (0, 1)))


def povey_window(n):
    return np.hanning(n) ** 0.85 if n > 1 else np.ones(n)   # np.hanning == hann(periodic=False)


def frame_geometry(num_samples, sample_freq, frame_length=25.0, frame_shift=10.0):
    win = int(sample_freq * frame_length * 0.001)
    shift = int(sample_freq * frame_shift * 0.001)
    padded = 1 << (win - 1).bit_length()
    m = 0 if num_samples < win else 1 + (num_samples - win) // shift
    return win, shift, padded, m


def kaldi_fbank(wave_1d, sample_freq, num_mel_bins=40, frame_length=25.0, frame_shift=10.0,
                preemph=0.97, dither=0.0):
    """-> [m, num_mel_bins] log-mel energies."""
    assert dither == 0.0, "dither is random; the shipped configs set 0"
    x = np.asarray(wave_1d, np.float64)
    win, shift, padded, m = frame_geometry(len(x), sample_freq, frame_length, frame_shift)
    if m == 0:
        return np.zeros((0, num_mel_bins))
    idx = np.arange(win)[None, :] + shift * np.arange(m)[:, None]
    fr = x[idx]                                            # snip_edges framing
    fr = fr - fr.mean(axis=1, keepdims=True)               # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)  # replicate-pad the first sample
    fr = fr - preemph * prev                               # pre-emphasis
    fr = fr * povey_window(win)[None, :]
    spec = np.fft.rfft(fr, n=padded, axis=1)
    power = spec.real ** 2 + spec.imag ** 2                # use_power=True
    mel = power @ mel_banks(num_mel_bins, padded, sample_freq).T
    return np.log(np.maximum(mel, EPS))


def kaldi_mfcc(wave_1d, sample_freq, num_mel_bins=23, num_ceps=13, cepstral_lifter=22.0, **kw):
    """MFCC as torchaudio.compliance.kaldi.mfcc defines it on top of its fbank (feat_type 'mfcc',
    src/audio.py:96; parity unpinned like the fbank, see the header): log-mel energies -> Kaldi DCT-II
    (orthonormal, row 0 = sqrt(1/N)) -> first num_ceps -> lifter 1 + Q/2 sin(pi i / Q).  float64."""
    mel = kaldi_fbank(wave_1d, sample_freq, num_mel_bins=num_mel_bins, **kw).astype(np.float64)
    N = num_mel_bins
    out = np.zeros((mel.shape[0], num_ceps))
    for k in range(num_ceps):
        scale = math.sqrt(1.0 / N) if k == 0 else math.sqrt(2.0 / N)
        basis = np.array([math.cos(math.pi / N * (n + 0.5) * k) for n in range(N)])
        lift = 1.0 + 0.5 * cepstral_lifter * math.sin(math.pi * k / cepstral_lifter) if cepstral_lifter else 1.0
        out[:, k] = mel @ basis * scale * lift
    return out


def delta_filters(order, window_size=2):
    """src/audio.py:57-77: filter bank [order+1, L] (row i = i-th order delta, centred)."""
    scales = [[1.0]]
    for i in range(1, order + 1):
        prev_off = (len(scales[i - 1]) - 1) // 2
        cur_off = prev_off + window_size
        cur = [0.0] * (len(scales[i - 1]) + 2 * window_size)
        norm = 0.0
        for j in range(-window_size, window_size + 1):
            norm += j * j
            for k in range(-prev_off, prev_off + 1):
                cur[j + k + cur_off] += j * scales[i - 1][k + prev_off]
        scales.append([v / norm for v in cur])
    L = len(scales[-1])
    out = np.zeros((order + 1, L))
    for i, s in enumerate(scales):
        pad = (L - len(s)) // 2
        out[i, pad:pad + len(s)] = s
    return out


def delta(feat_dt, order, window_size=2):
    """feat [D, T] -> [order+1, D, T]; conv2d with ZERO padding along time (src/audio.py:48-54).
    conv2d is a cross-correlation: out[t] = sum_j f[j] * x[t + j - half]."""
    f = delta_filters(order, window_size)
    half = (f.shape[1] - 1) // 2
    D, T = feat_dt.shape
    xp = np.pad(np.asarray(feat_dt, np.float64), ((0, 0), (half, half)))
    out = np.zeros((order + 1, D, T))
    for c in range(order + 1):
        for j in range(f.shape[1]):
            out[c] += f[c, j] * xp[:, j:j + T]
    return out


def cmvn(x_cdt, eps=1e-10):
    """per (channel, feature) over time, unbiased std, eps added to std (src/audio.py:24-27)."""
    x = np.asarray(x_cdt, np.float64)
    mean = x.mean(axis=2, keepdims=True)
    std = x.std(axis=2, ddof=1, keepdims=True)
    return (x - mean) / (eps + std)


def postprocess(x_cdt):
    """[C, D, T] -> [T, C*D], feature index c*D + d (src/audio.py:85-89)."""
    C, D, T = x_cdt.shape
    return np.transpose(x_cdt, (2, 0, 1)).reshape(T, C * D)


def audio_transform(wave_1d, sample_freq, feat_dim=40, delta_order=0, delta_window_size=2,
                    apply_cmvn=True, frame_length=25.0, frame_shift=10.0, dither=0.0):
    """create_transform(...)(file) of src/audio.py:115-133 on an already loaded waveform."""
    fb = kaldi_fbank(wave_1d, sample_freq, feat_dim, frame_length, frame_shift, dither=dither)
    x = fb.T[None]                                         # [1, D, T] (src/audio.py:109)
    if delta_order >= 1:
        x = delta(x[0], delta_order, delta_window_size)
    if apply_cmvn:
        x = cmvn(x)
    return postprocess(x)
